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
band = ((r >= 80) & (r < 120))[None, None, :, None, None]
tests = {"band": full * band, "band, real part only": (full.real * band).astype(np.complex64), "band x 0.37": full * band * np.float32(0.37),
         "band, every 2nd spoke": full * band * (np.arange(npe) % 2 == 0)[None, None, None, :, None],
         "band, every 2nd radius": full * band * (np.arange(nro) % 2 == 0)[None, None, :, None, None],
         "band, all ones": (np.ones_like(full) * band).astype(np.complex64)}
for name, data in tests.items():
    data = np.asfortranarray(data.astype(np.complex64))
    got, _ = lib.recon(data, adjoint=True, **fl)
    want, _ = pyoracle.recon(data, adjoint=1, golden=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    d = (got - want).ravel(); k = int(np.argmax(np.abs(d)))
    print(sys.argv[1] if len(sys.argv) > 1 else "", name, f"{rel_l2(got, want):.2e}", "max |diff| at", np.unravel_index(k, got.shape, order="C")[2:4], f"{abs(d[k]):.2e} of rms {np.sqrt(np.mean(np.abs(want)**2)):.2e}", flush=True)
