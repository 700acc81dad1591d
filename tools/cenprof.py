"""Where a wave of the centre gridding kernel (tron_grid_centre.hip) spends its cycles (kernel work tooling).
Needs a -DTRON_PHASE_CLOCK build copied over tron_amd/lib/libtronhip.so:
    tools/build_variants.sh cprof:"-DTRON_PHASE_CLOCK":tron_grid_centre.hip     (then, on the GPU box)
    cp tron_amd/lib/libtronhip_cprof.so tron_amd/lib/libtronhip.so; python tools/cenprof.py [coils] [slices]"""
import ctypes, os, sys
os.environ.setdefault("TRON_TUNING", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 128
NRO, NPE = 512, int(os.environ.get("NPE", "402"))
cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=0.7852, prof_slide=NPE, kb_mode=lib.KB_FAST)
dims = lib.derive_dims(cfg, (nc, 1, NRO, NPE * nz, 1))
data = (np.random.default_rng(1).random(2 * nc * NRO * NPE * nz, dtype=np.float32) * 2 - 1)
L = lib.load()
fn = L.tron_debug_cen_profile
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)()
with lib.Plan(cfg, dims) as plan:
    d_in = lib.DeviceBuffer.from_numpy(data)
    d_out = lib.DeviceBuffer(dims.out_bytes)
    for _ in range(int(os.environ.get("WARM", "10"))):
        plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1)
    plan.sync()
    assert fn(buf) == 0
    plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1); plan.sync()
    assert fn(buf) == 0
waves, items, chunks, groups, visits = buf[12], buf[8], buf[6], buf[7], buf[9]
print(f"nc={nc} nz={nz}: {waves} waves, {items / nz:.0f} items per slice, {chunks / nz:.0f} chunks, {groups / nz:.0f} groups, {visits / nz:.0f} visits per slice")
print(f"  in-kernel clock {buf[5] / buf[10] * 100:.0f} MHz; cycles per wave {buf[5] / waves:.0f}, per item {buf[5] / items:.0f}")
print(f"  per chunk (clip) {buf[1] / max(chunks, 1):.0f}; per group: weights + requests {buf[2] / max(groups, 1):.0f}, sums {buf[3] / max(groups, 1):.0f}")
for name, i in (("table (once per wave)", 0), ("item: ticket, records", 11), ("clip", 1), ("weights + requests", 2), ("sums", 3), ("reduce + add to the grid", 4)):
    print(f"  {name:28s} {100.0 * buf[i] / buf[5]:6.2f} %   ({buf[i] / items:.0f} cycles per item)")
