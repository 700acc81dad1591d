"""CGNR and Walsh-combination runs at the metric shape for profiling (tooling): device-resident, timed with host clocks.
usage: python tools/cgnrbench.py cgnr|walsh [coils] [slices] [iterations / patch] [reps]
Run under `rocprofv3 --kernel-trace --stats -- python3 tools/cgnrbench.py ...` for per-kernel times."""
import os, sys, time
os.environ.setdefault("TRON_TUNING", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
mode = sys.argv[1] if len(sys.argv) > 1 else "cgnr"
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nz = int(sys.argv[3]) if len(sys.argv) > 3 else 16
par = int(sys.argv[4]) if len(sys.argv) > 4 else (5 if mode == "cgnr" else 1)
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
NRO, NPE = 512, 402
flags = dict(adjoint=1, golden_angle=1, data_undersamp=0.7852, prof_slide=NPE, kb_mode=lib.KB_FAST)
if mode == "cgnr":
    flags["niter"] = par
else:
    flags.update(coil_combine=1, walsh_patch=par)
cfg = lib.default_config(**flags)
dims = lib.derive_dims(cfg, (nc, 1, NRO, NPE * nz, 1))
rng = np.random.default_rng(1)
data = (rng.random(2 * nc * NRO * NPE * nz, dtype=np.float32) * 2 - 1)
with lib.Plan(cfg, dims) as plan:
    d_in = lib.DeviceBuffer.from_numpy(data)
    d_out = lib.DeviceBuffer(dims.out_bytes)
    run = (lambda: plan.cgnr_device(d_out.ptr, d_in.ptr, 0, nz, 1)) if mode == "cgnr" else (lambda: plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1))
    run(); plan.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    plan.sync()
    dt = (time.perf_counter() - t0) / reps
print(f"{mode} nc={nc} nz={nz} par={par}: {dt * 1e3:.2f} ms per call = {dt / nz * 1e3:.3f} ms per slice")
