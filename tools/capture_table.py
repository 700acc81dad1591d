"""Prints the measurement table of DESIGN.md section 5 from the bench lines of one capture (tools/round_capture.sh <tag>):
    python tools/capture_table.py gpurun_out/<tag>      (or: profiles round5_)"""
import json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "profiles"
pre = sys.argv[2] if len(sys.argv) > 2 else ("round5_" if d == "profiles" else "")
ROWS = ["default", "nc6", "nc4", "nc2", "nc1", "half", "half_nc6", "half_nc1", "804spokes", "cfg4_share", "32slices", "exact", "linear_nc1",
        "forward", "binned_kernel", "cfg4_strong_2ranks_shared_gpu"]
print("| line | short | sustained (ratio; clock MHz) | roofline.frac | traffic / algorithmic | parity | kernel |")
print("|---|---|---|---|---|---|---|")
for n in ROWS:
    p = os.path.join(d, f"{pre}bench_{n}.json")
    try:
        j = json.loads([l for l in open(p) if l.startswith("{")][-1])
    except Exception as e:
        print(f"| {n} | (missing: {e}) |")
        continue
    r, s = j["roofline"], j.get("sustained") or {}
    tr = f"{r['traffic'] / r['bytes_per_launch']:.2f}" if r.get("traffic") else "—"
    sus = f"{s['value'] / 1e3:.1f} k ({s['ratio_to_value']:.2f}; {s['shader_clock_mhz_start']:.0f} -> {s['shader_clock_mhz_end']:.0f})" if s else "—"
    par = f"{j['parity_rel_l2_vs_oracle']:.1e}" if j.get("parity_rel_l2_vs_oracle") is not None else "—"
    print(f"| {n} | {j['value'] / 1e3:.1f} k ({100 * j['algorithmic_frac_of_peak']:.1f} %) | {sus} | {r['frac']:.3f} | {tr} | {par} | {r['kernel'].split(' ')[0]} |")
