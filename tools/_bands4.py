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
ro = np.arange(nro)
sel = ((ro - 128 >= 80) & (ro - 128 < 120) & (ro % 2 == 1))[None, None, :, None, None]
def err(data):
    data = np.asfortranarray(data.astype(np.complex64))
    got, _ = lib.recon(data, adjoint=True, **fl)
    want, _ = pyoracle.recon(data, adjoint=1, golden=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    return rel_l2(got, want)
bad = []
for pe in range(91, 180, 2):
    e = err(full * sel * (np.arange(npe) == pe)[None, None, None, :, None])
    if e > 1e-6: bad.append((pe, e))
print("bad spokes", bad, flush=True)
for pe, _ in bad[:3]:
    for r in range(128 + 81, 128 + 120, 2):
        d = np.zeros_like(full); d[0, 0, r, pe, 0] = full[0, 0, r, pe, 0]
        e = err(d)
        if e > 1e-6: print("  pe", pe, "ro", r, "r", r - 128, f"{e:.2e}", flush=True)
