"""Where a wave of the scatter gridding kernel spends its cycles, phase by phase (kernel work tooling).
Needs a -DTRON_PHASE_CLOCK build of tron_grid_scatter.hip copied over tron_amd/lib/libtronhip.so:
    tools/build_variants.sh sprof:"-DTRON_PHASE_CLOCK":tron_grid_scatter.hip     (then, on the GPU box)
    cp tron_amd/lib/libtronhip_sprof.so tron_amd/lib/libtronhip.so; python tools/scatprof.py [coils] [slices]"""
import ctypes, os, sys
os.environ.setdefault("TRON_TUNING", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 128
NRO, NPE = 512, int(os.environ.get("NPE", "402"))
NAMES = ["set-up (first slice: window table)", "barrier", "first member of the quarter", "walk", "wait for the samples, maximum", "barrier", "scatter", "barrier", "store, zeroing, next table", "entries, loads issued"]
cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=(NPE + 0.5) / NRO, prof_slide=NPE, kb_mode=lib.KB_FAST)
dims = lib.derive_dims(cfg, (nc, 1, NRO, NPE * nz, 1))
rng = np.random.default_rng(1)
data = (rng.random(2 * nc * NRO * NPE * nz, dtype=np.float32) * 2 - 1)
L = lib.load()
fn = L.tron_debug_scat_profile
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
with lib.Plan(cfg, dims) as plan:
    assert "scatter" in plan.grid_kernel_name(), plan.grid_kernel_name()
    d_in = lib.DeviceBuffer.from_numpy(data)
    d_out = lib.DeviceBuffer(dims.out_bytes)
    for _ in range(int(os.environ.get("WARM", "20"))):
        plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1)
    plan.sync()
    assert fn(buf, 16) == 0
    plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1); plan.sync()
    assert fn(buf, 16) == 0
tot = float(sum(buf[:10]))
print(f"nc={nc} nz={nz}: {tot:.3e} wave-cycles in total ({tot / nz:.3e} per slice)")
for name, v in zip(NAMES, buf):
    print(f"  {name:44s} {100.0 * v / tot:6.2f} %")
if buf[11]:
    print(f"  in-kernel clock (s_memtime / s_memrealtime x 100 MHz): {buf[10] / buf[11] * 100:.0f} MHz")
print(f"  wave iterations (64 records) per slice {buf[12] / nz:.0f}; scatter cycles per iteration {buf[6] / max(buf[12], 1):.0f}")
