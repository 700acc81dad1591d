#!/usr/bin/env python3
"""Prints DESIGN.md 5's table of workloads from the bench lines of a capture (profiles/round6_bench_*.json), so that the document's numbers are
the evidence's.   usage: python tools/design_table.py [prefix = profiles/round6_bench_]"""
import json
import sys

prefix = sys.argv[1] if len(sys.argv) > 1 else "profiles/round6_bench_"
ROWS = [("default", "8 coils × 256 slices (the metric; the line with `cpu_baseline`, `one_coil`)"), ("nc6", "`--coils 6`"), ("nc4", "`--coils 4`"), ("nc2", "`--coils 2`"),
        ("nc1", "`--coils 1` (`grid_scatter_kernel`)"), ("half", "`--half` (complex-half k-space)"), ("half_nc6", "`--half --coils 6`"), ("half_nc1", "`--half --coils 1`"),
        ("804spokes", "`--spokes 804` (config 4's slice shape)"), ("cfg4_share", "`--spokes 804 --slices 32` (config 4's per-GPU share)"), ("32slices", "`--slices 32`"),
        ("exact", "`--kb exact` (bit-identical gather)"), ("linear_nc1", "`--linear --coils 1` (binned kernel, slice groups)"), ("forward", "`--forward` (images/s)"),
        ("binned_kernel", "`TRON_GRID_KERNEL=binned` (round 2's kernel on every tile)"),
        ("cfg4_strong_2ranks_shared_gpu", "2 ranks sharing the GPU, config 4 (`--gpus 2 --scaling strong --spokes 804`)")]


def k(v):
    return f"{v / 1000:.1f} k"


for name, label in ROWS:
    d = json.loads(open(f"{prefix}{name}.json").read().strip().splitlines()[-1])
    r, s = d["roofline"], d.get("sustained") or {}
    sus = f"{k(s['value'])} ({s['ratio_to_value']:.2f}; {s['shader_clock_mhz_start']:.0f} → {s['shader_clock_mhz_end']:.0f})" if s else "—"
    fr = d.get("fresh_trajectory_slices_per_s")
    fresh = f"{k(fr)} ({fr / d['value']:.2f})" if fr else "—"
    tr = r.get("traffic")
    ratio = f"{tr / r['bytes_per_launch']:.2f}" if tr else "—"
    alg = d.get("algorithmic_frac_of_peak")
    val = f"**{k(d['value'])}**" if name in ("default", "nc1") else k(d["value"])
    if name == "default":
        val += f" ({100 * alg:.1f} % of 8 TB/s by algorithmic bytes)"
    if name == "forward":
        val += f" ({100 * alg:.1f} %)"
    par = d.get("parity_rel_l2_vs_oracle")
    par = f"{par:.1e}" if par else "—"
    extra = ""
    if name == "default":
        oc = d["one_coil"]
        extra = f" — `one_coil` field: {k(oc['value'])} / {k(oc['sustained_slices_per_s'])}"
    print(f"| {label}{extra} | {val} | {sus} | {fresh} | {r['frac']:.3f} | {ratio} | {par} |")
