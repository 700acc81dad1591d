"""Determinism / race screen for the gridding kernels (tooling): runs the adjoint several times on the
metric shape and compares output bits between runs and against the order-preserving gather kernel."""
import ctypes, hashlib, os, sys
os_env_ = __import__("os").environ; os_env_.setdefault("TRON_TUNING", "1")   # the library reads TRON_* switches only under TRON_TUNING=1
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tron_amd import lib
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
NRO, NPE = 512, 402
rng = np.random.default_rng(3)
data = (rng.random(2 * nc * NRO * NPE * nz, dtype=np.float32) * 2 - 1)
def run(kernel):
    os.environ["TRON_GRID_KERNEL"] = kernel
    cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=0.7852, prof_slide=NPE, kb_mode=lib.KB_FAST)
    dims = lib.derive_dims(cfg, (nc, 1, NRO, NPE * nz, 1))
    outs = []
    with lib.Plan(cfg, dims) as plan:
        d_in = lib.DeviceBuffer.from_numpy(data)
        d_out = lib.DeviceBuffer(dims.out_bytes)
        for _ in range(reps):
            plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1); plan.sync()
            outs.append(d_out.to_numpy(np.complex64, dims.out_bytes // 8))
    return outs
b = run("binned")
hs = [hashlib.md5(o.tobytes()).hexdigest()[:8] for o in b]
print("binned hashes", hs, "deterministic" if len(set(hs)) == 1 else "NONDETERMINISTIC")
g = run("gather")
print("gather hashes", set(hashlib.md5(o.tobytes()).hexdigest()[:8] for o in g))
for i, o in enumerate(b):
    err = np.linalg.norm(o - g[0]) / np.linalg.norm(g[0])
    bad = np.flatnonzero(np.abs(o - g[0]) > 1e-3 * np.abs(g[0]).max())
    print(i, "rel l2 vs gather %.3e" % err, "bad", bad.size, (bad[:5] // (256*256), (bad[:5] % (256*256)) // 256, bad[:5] % 256) if bad.size else "")
