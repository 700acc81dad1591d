"""Where a wave of the arc gridding kernel spends its cycles, phase by phase, and how full its loops run (kernel work tooling).
Needs a -DTRON_PHASE_CLOCK build of tron_grid_arc.hip copied over tron_amd/lib/libtronhip.so:
    tools/build_variants.sh aprof:"-DTRON_PHASE_CLOCK":tron_grid_arc.hip     (then, on the GPU box)
    cp tron_amd/lib/libtronhip_aprof.so tron_amd/lib/libtronhip.so; python tools/arcprof.py [coils] [slices]"""
import ctypes, os, sys
os_env_ = __import__("os").environ; os_env_.setdefault("TRON_TUNING", "1")   # the library reads TRON_* switches only under TRON_TUNING=1
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NRO, NPE = 512, int(os.environ.get("NPE", "402"))
NAMES = ["tile setup + run table", "barrier", "DMA issue", "window search", "DMA wait", "barrier", "gather", "output store"]
cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=0.7852, prof_slide=NPE, kb_mode=lib.KB_FAST)
dims = lib.derive_dims(cfg, (nc, 1, NRO, NPE * nz, 1))
rng = np.random.default_rng(1)
data = (rng.random(2 * nc * NRO * NPE * nz, dtype=np.float32) * 2 - 1)
if os.environ.get("DATA") == "zeros":
    data[:] = 0
L = lib.load()
fn = L.tron_debug_arc_profile
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
with lib.Plan(cfg, dims) as plan:
    d_in = lib.DeviceBuffer.from_numpy(data)
    d_out = lib.DeviceBuffer(dims.out_bytes)
    for _ in range(int(os.environ.get("WARM", "1"))):         # the clock settles after tens of milliseconds of load (tools/README.md)
        plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1)
    plan.sync()
    assert fn(buf, 16) == 0                                   # clears the warm-up launches
    plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1); plan.sync()
    assert fn(buf, 16) == 0
tot = float(sum(buf[:8]))
print(f"nc={nc} nz={nz}: {tot:.3e} wave-cycles in total")
for name, v in zip(NAMES, buf):
    print(f"  {name:40s} {100.0 * v / tot:6.2f} %")
if buf[9]:
    print(f"  in-kernel clock (s_memtime / s_memrealtime x 100 MHz, summed over all waves): {buf[8] / buf[9] * 100:.0f} MHz")
print(f"  outer (spoke) iterations per slice {buf[12] / nz:.0f}, lanes active {buf[13] / max(buf[12], 1) / 64:.3f}")
print(f"  inner (radius) iterations per slice {buf[14] / nz:.0f}, lanes active {buf[15] / max(buf[14], 1) / 64:.3f}; "
      f"gather cycles per inner iteration {buf[6] / max(buf[14], 1):.0f}")
