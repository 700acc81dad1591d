import os, sys
os.environ.setdefault("TRON_TUNING", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from tron_amd import lib
from oracle import pyoracle
from conftest import rel_l2
nc, nro, npe = 1, 256, 180
fl = dict(golden_angle=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
cfg = lib.default_config(adjoint=1, **fl)
rng = np.random.default_rng(3)
worst = []
with_plan = None
res = []
for trial in range(120):
    pe = int(rng.integers(0, npe)); ro = int(rng.integers(1, nro))
    if trial < 20: pe = 0; ro = 128 + 14 + trial * 5 if 128 + 14 + trial*5 < 256 else 200
    data = np.zeros((nc, 1, nro, npe, 1), dtype=np.complex64, order="F")
    data[0, 0, ro, pe, 0] = 1.0 + 0.5j
    got, _ = lib.recon(data, adjoint=True, **fl)
    want, _ = pyoracle.recon(data, adjoint=1, golden=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    e = rel_l2(got, want)
    res.append((e, pe, ro))
res.sort(reverse=True)
print(os.environ.get("TRON_GRID_KERNEL", "default"), "worst:", [(f"{e:.2e}", pe, ro) for e, pe, ro in res[:12]], "median", f"{res[len(res)//2][0]:.2e}")
