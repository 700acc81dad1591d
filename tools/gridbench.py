"""Stage timing of the adjoint on the metric shape (tooling for kernel work).
usage: python tools/gridbench.py [coils] [slices] [kb fast|exact] [reps]"""
import ctypes, os, sys, time
os_env_ = __import__("os").environ; os_env_.setdefault("TRON_TUNING", "1")   # the library reads TRON_* switches only under TRON_TUNING=1
import numpy as np
if os.environ.get("WITH_TORCH") == "1":     # as bench.py: torch's HIP runtime first (the stage times of the two scripts differ by which runtime the process holds)
    import torch
    torch.zeros(1, device="cuda")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kb = lib.KB_FAST if (len(sys.argv) <= 3 or sys.argv[3] == "fast") else lib.KB_EXACT
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
NRO, NPE = 512, 402
cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=0.7852, prof_slide=NPE, kb_mode=kb)
dims = lib.derive_dims(cfg, (nc, 1, NRO, NPE * nz, 1))
rng = np.random.default_rng(1)
data = (rng.random(2 * nc * NRO * NPE * nz, dtype=np.float32) * 2 - 1)
if os.environ.get("DATA") == "zeros":      # same instruction stream, no toggling operands: a kernel that speeds up on zeros is held down by power (DVFS)
    data[:] = 0
with lib.Plan(cfg, dims) as plan:
    d_in = lib.DeviceBuffer.from_numpy(data)
    d_out = lib.DeviceBuffer(dims.out_bytes)
    # WARM untimed calls first (default 1): the chip's clock takes tens of milliseconds of load to settle, and a run of a few
    # launches from idle is timed on the ramp (gridbench's 5 reps read 15-19 % slower than bench.py's steady state for that reason)
    for _ in range(int(os.environ.get("WARM", "1"))):
        plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1)
    plan.sync()
    plan.timing(True); plan.timing_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1)
    try:
        plan.sync()
    except Exception as e:
        print('sync error (ignored for timing):', str(e)[:80])
    dt = time.perf_counter() - t0
    cs = nc * nz * reps
    line = f"nc={nc} nz={nz} kb={'fast' if kb else 'exact'}: total {dt/cs*1e6:.3f} us/coil-slice ({nz*reps/dt:.0f} slices/s)"
    for st, name in ((lib.STAGE_GRID, "grid"), (lib.STAGE_FFT, "fft"), (lib.STAGE_POST, "post")):
        ms, n = plan.timing_get(st)
        line += f" | {name} {ms/cs*1e3:.3f}"
    print(line)
