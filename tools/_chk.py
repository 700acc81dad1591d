import os, sys
os.environ.setdefault("TRON_TUNING", "1")
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, synth
from tron_amd import lib
from oracle import pyoracle
from conftest import rel_l2
for nc, nro, npe, nz in ((1, 256, 180, 2), (2, 256, 150, 2), (1, 512, 402, 1)):
    data = synth.kspace(nc, nro, npe * nz, seed=9500 + nc + nro + npe)
    fl = dict(golden_angle=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    got, _ = lib.recon(data, adjoint=True, **fl)
    want, _ = pyoracle.recon(data, adjoint=1, golden=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    print(os.environ.get("TRON_GRID_KERNEL", "default"), nc, nro, npe, "rel l2 vs oracle", rel_l2(got, want), flush=True)
