"""Stage timing of the forward (degridding) direction: nimg images of 256^2 -> 512 ro x 512 spokes (tooling)."""
import os, sys, time
os_env_ = __import__("os").environ; os_env_.setdefault("TRON_TUNING", "1")   # the library reads TRON_* switches only under TRON_TUNING=1
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nimg = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kb = lib.KB_FAST if (len(sys.argv) <= 3 or sys.argv[3] == "fast") else lib.KB_EXACT
cfg = lib.default_config(adjoint=0, golden_angle=1, kb_mode=kb)
dims = lib.derive_dims(cfg, (nc, 1, 256, 256, 1))
rng = np.random.default_rng(2)
img = (rng.random(2 * nc * 256 * 256 * nimg, dtype=np.float32) * 2 - 1)
with lib.Plan(cfg, dims) as plan:
    d_in = lib.DeviceBuffer.from_numpy(img)
    d_out = lib.DeviceBuffer(nimg * nc * dims.nro * dims.npe1work * 8)
    plan.forward_device(d_out.ptr, d_in.ptr, nimg); plan.sync()
    plan.timing(True); plan.timing_reset()
    t0 = time.perf_counter()
    for _ in range(3):
        plan.forward_device(d_out.ptr, d_in.ptr, nimg)
    plan.sync()
    dt = time.perf_counter() - t0
    ci = nc * nimg * 3
    line = f"forward nc={nc} nimg={nimg}: {dt/ci*1e6:.3f} us/coil-image ({nimg*3/dt:.0f} images/s)"
    for st, name in ((lib.STAGE_PRE, "pre"), (lib.STAGE_FFT, "fft"), (lib.STAGE_DEGRID, "degrid")):
        ms, n = plan.timing_get(st)
        line += f" | {name} {ms/ci*1e3:.3f}"
    print(line)
