"""Where tron_plan_create spends its time (tooling): the metric-shape plan (or `coils slices spokes`) created three times with verbose = 1,
after the process's first plan (HIP runtime + code objects) -- prints the library's own breakdown of the table time.
    python tools/plantime.py [coils] [slices] [spokes]"""
import os, sys, time
os.environ.setdefault("TRON_TUNING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 256
npe = int(sys.argv[3]) if len(sys.argv) > 3 else 402
us = 0.7852 if npe == 402 else (npe + 0.5) / 512
for rep in range(4):
    cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=us, prof_slide=npe, verbose=1)
    dims = lib.derive_dims(cfg, (nc, 1, 512, npe * nz, 1))
    t0 = time.perf_counter()
    plan = lib.Plan(cfg, dims)
    dt = time.perf_counter() - t0
    t = plan.create_times()
    print(f"plan {rep}: {dt * 1e3:.1f} ms wall; tables {t['tables'] * 1e3:.1f} ms of which trajectory tables {t['run_tables'] * 1e3:.1f}, work buffers {t['work_buffers'] * 1e3:.1f}", flush=True)
    t0 = time.perf_counter()
    plan.retarget(12345)
    plan.sync()
    print(f"   retarget: {plan.retarget_times()}", flush=True)
    plan.close()
