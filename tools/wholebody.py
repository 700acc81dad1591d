"""Whole-body-shaped end-to-end run of the `tron` CLI (config 3): synthetic [6,1,512,20271,1] complex64 stream
(498 180 184 B incl. header, the size the reference's LFS pointer declares), `tron -v -u 0.4 -d 21 -a -G`
(src/RUNME3_tron_grid_all.sh:10) -> 956 slices.  Prints file sizes and wall times (tooling)."""
import os, subprocess, sys, time
os_env_ = __import__("os").environ; os_env_.setdefault("TRON_TUNING", "1")   # the library reads TRON_* switches only under TRON_TUNING=1
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from tron_amd import ra
tmp = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
inp, out = os.path.join(tmp, "wb_in.ra"), os.path.join(tmp, "wb_out.ra")
t0 = time.perf_counter()
ra.write(inp, synth.kspace(6, 512, 20271, seed=synth.SEED_BASE + 3))
print(f"wrote {inp}: {os.path.getsize(inp)} B in {time.perf_counter()-t0:.1f} s")
extra = os.environ.get("WB_OPTIONS", "")           # e.g. WB_OPTIONS=pin=1: appended to TRON_OPTIONS
pause = float(os.environ.get("WB_PAUSE", "2"))     # seconds between runs: a process that starts while the previous one's GPU context is
                                                   # still being torn down pays 0.15 s more for its own (the HIP runtime line of -v)
for mode in ("fast", "exact"):
    env = dict(os.environ, TRON_OPTIONS=f"kb={mode}" + ("," + extra if extra else ""))
    for rep in range(3 if mode == "fast" else 1):
        time.sleep(pause)
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(ROOT, "tron_amd/bin/tron"), "-v", "-u", "0.4", "-d", "21", "-a", "-G", inp, out], capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        el = [l for l in r.stdout.splitlines() if "time" in l or "start-up" in l]
        print(f"KB {mode} run {rep}: rc={r.returncode} wall {dt:.2f} s  ({956/dt:.0f} slices/s end to end incl. file I/O); {el}")
h = ra.read_header(out)
print("output dims", h.dims, "bytes", os.path.getsize(out))
