"""PCIe-inclusive rate of the C-ABI host-buffer entry point (tron_recon_radial2d) on the metric shape (tooling):
upload of the spoke stream + gridding recon + download of the images, pageable host memory as the `tron` binary uses.
usage: python tools/hostbench.py [coils] [slices] [--half]      (--half: k-space stored as complex-half, the float16.cu rounding: half the upload)"""
import os, sys, time
os_env_ = __import__("os").environ; os_env_.setdefault("TRON_TUNING", "1")   # the library reads TRON_* switches only under TRON_TUNING=1
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
half = "--half" in sys.argv
argv = [a for a in sys.argv if a != "--half"]
nc = int(argv[1]) if len(argv) > 1 else 8
nz = int(argv[2]) if len(argv) > 2 else 256
NRO, NPE = 512, 402
cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=0.7852, prof_slide=NPE, kb_mode=lib.KB_FAST, input_half=1 if half else 0)
dims = lib.derive_dims(cfg, (nc, 1, NRO, NPE * nz, 1))
rng = np.random.default_rng(3)
flat = (rng.random(2 * nc * NRO * NPE * nz, dtype=np.float32) * 2 - 1)
flat = flat.astype(np.float16) if half else flat.view(np.complex64)
out = np.zeros(dims.out_bytes // 8, np.complex64)
with lib.Plan(cfg, dims) as plan:
    plan.recon(flat, out=out)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        plan.recon(flat, out=out)
    dt = (time.perf_counter() - t0) / reps
gb = (flat.nbytes + out.nbytes) / 1e9
print(f"host-buffer recon nc={nc} nz={nz}{' complex-half k-space' if half else ''}: {dt*1e3:.1f} ms per call = {nz/dt:.0f} slices/s; {gb:.2f} GB over PCIe = {gb/dt:.1f} GB/s effective")
