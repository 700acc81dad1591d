"""tron_recon_radial2d_multi with eight workers on one GPU, again and again, against one plan's bytes (hunting the race behind
test_config4_eight_workers_on_one_gpu_give_the_single_plan_bytes failing on some runs)."""
import os, sys
os.environ.setdefault("TRON_TUNING", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth
from tron_amd import lib
nz = int(os.environ.get("NZ", "24")); npe = int(os.environ.get("NPE", "804")); nw = int(os.environ.get("NW", "8"))
data = synth.kspace(8, 512, npe * nz, seed=synth.SEED_BASE + 26)
fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / 512, prof_slide=npe)
one, dims = lib.recon(data, adjoint=True, **fl)
print("single plan done", dims.nz, flush=True)
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    multi, _ = lib.recon_multi(data, adjoint=True, devices=[0] * nw, **fl)
    if not np.array_equal(one, multi):
        d = np.abs(one - multi); w = np.argwhere(d > 0)
        bad += 1
        print(f"rep {rep}: DIFFERS at {len(w)} values, max {d.max():.3e} (ref max {np.abs(one).max():.3e}); slices {sorted(set(w[:, -1].tolist()))}; rows {w[:,2].min()}..{w[:,2].max()} cols {w[:,3].min()}..{w[:,3].max()}; nan {np.isnan(multi).sum()}", flush=True)
    again, _ = lib.recon(data, adjoint=True, **fl)
    if not np.array_equal(one, again):
        print(f"rep {rep}: single plan differs from itself!", flush=True); bad += 1
print("differing runs:", bad)
