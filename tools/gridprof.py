"""Where a gridding wave spends its cycles, phase by phase (kernel work tooling).
Needs a -DTRON_PHASE_CLOCK build of the gridding kernel copied over tron_amd/lib/libtronhip.so:
    tools/build_variants.sh prof:"-DTRON_PHASE_CLOCK"       (then, on the GPU box)
    cp tron_amd/lib/libtronhip_prof.so tron_amd/lib/libtronhip.so; python tools/gridprof.py [coils] [slices]
Prints shader-clock cycles per wave and phase (tron_grid_binned.hip: PROF_MARK slots); the
gridding kernel runs alone."""
import ctypes, os, sys
os_env_ = __import__("os").environ; os_env_.setdefault("TRON_TUNING", "1")   # the library reads TRON_* switches only under TRON_TUNING=1
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NRO, NPE = 512, 402
NAMES = ["setup+clip+segment scan", "first prefetch", "batch open (prefetch use, DMA issue, hist clear)", "barrier A",
         "stage", "next-batch bounds + prefetch", "barrier B", "scan", "barrier C", "place", "DMA wait", "barrier D",
         "apply", "barrier E", "(loop exit)", "output store"]
cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=0.7852, prof_slide=NPE, kb_mode=lib.KB_FAST)
dims = lib.derive_dims(cfg, (nc, 1, NRO, NPE * nz, 1))
rng = np.random.default_rng(1)
data = (rng.random(2 * nc * NRO * NPE * nz, dtype=np.float32) * 2 - 1)
L = lib.load()
fn = L.tron_debug_grid_profile
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
with lib.Plan(cfg, dims) as plan:
    d_in = lib.DeviceBuffer.from_numpy(data)
    d_out = lib.DeviceBuffer(dims.out_bytes)
    plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1); plan.sync()
    assert fn(buf, 16) == 0                                   # clears the warm-up launch
    plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1); plan.sync()
    assert fn(buf, 16) == 0
tot = float(sum(buf))
print(f"nc={nc} nz={nz}: {tot:.3e} wave-cycles in total")
for name, v in zip(NAMES, buf):
    print(f"  {name:50s} {100.0 * v / tot:6.2f} %")
