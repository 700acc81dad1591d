import os, sys
os.environ.setdefault("TRON_TUNING", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, synth
from tron_amd import lib
from oracle import pyoracle
from conftest import rel_l2
nc, nro, npe = 1, 256, 180
fl = dict(golden_angle=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
full = synth.kspace(nc, nro, npe, seed=9500 + nc + nro + npe)
r = np.abs(np.arange(nro) - nro // 2)
for lo, hi in ((0, 14), (14, 20), (20, 40), (40, 80), (80, 120), (120, 129), (14, 129)):
    data = np.asfortranarray(full * ((r >= lo) & (r < hi))[None, None, :, None, None])
    got, _ = lib.recon(data, adjoint=True, **fl)
    want, _ = pyoracle.recon(data, adjoint=1, golden=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    print(os.environ.get("TRON_GRID_KERNEL", "default"), "radii", lo, hi, f"{rel_l2(got, want):.2e}", flush=True)
for step in (2, 8, 32):
    data = np.asfortranarray(full * (np.arange(npe) % step == 0)[None, None, None, :, None])
    got, _ = lib.recon(data, adjoint=True, **fl)
    want, _ = pyoracle.recon(data, adjoint=1, golden=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    print(os.environ.get("TRON_GRID_KERNEL", "default"), "every", step, "th spoke", f"{rel_l2(got, want):.2e}", flush=True)
