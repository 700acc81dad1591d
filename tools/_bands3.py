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
def run(name, data):
    data = np.asfortranarray(data.astype(np.complex64))
    got, _ = lib.recon(data, adjoint=True, **fl)
    want, _ = pyoracle.recon(data, adjoint=1, golden=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    print(name, f"{rel_l2(got, want):.2e}", flush=True)
for pp in (0, 1):
    for rp in (0, 1):
        run(f"pe%2=={pp} ro%2=={rp}", full * band * (np.arange(npe) % 2 == pp)[None, None, None, :, None] * (np.arange(nro) % 2 == rp)[None, None, :, None, None])
run("odd spokes", full * band * (np.arange(npe) % 2 == 1)[None, None, None, :, None])
run("odd radii", full * band * (np.arange(nro) % 2 == 1)[None, None, :, None, None])
run("first 90 spokes", full * band * (np.arange(npe) < 90)[None, None, None, :, None])
run("r>0 side", full * band * (np.arange(nro) > 128)[None, None, :, None, None])
run("r<0 side", full * band * (np.arange(nro) < 128)[None, None, :, None, None])
