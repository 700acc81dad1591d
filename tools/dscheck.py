"""Streaming degridding kernel against the tile kernel on the same inputs: bit-identical by construction (tooling)."""
import os, sys, subprocess
os.environ["TRON_TUNING"]="1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def run(tile_only, nc, nimg, kbm, W=2.0, golden=1):
    code = f'''
import os, sys
os.environ["TRON_TUNING"]="1"
{"os.environ['TRON_DEGRID_KERNEL']='tile'" if tile_only else ""}
sys.path.insert(0, "{ROOT}")
import numpy as np
from tron_amd import lib
cfg = lib.default_config(adjoint=0, golden_angle={golden}, kb_mode=lib.{kbm}, kernwidth={W})
dims = lib.derive_dims(cfg, ({nc}, 1, 256, 256, 1))
rng = np.random.default_rng(2)
img = (rng.random(2 * {nc} * 256 * 256 * {nimg}, dtype=np.float32) * 2 - 1)
with lib.Plan(cfg, dims) as plan:
    d_in = lib.DeviceBuffer.from_numpy(img)
    nb = {nimg} * {nc} * dims.nro * dims.npe1work * 8
    d_out = lib.DeviceBuffer(nb)
    plan.forward_device(d_out.ptr, d_in.ptr, {nimg}); plan.sync()
    out = d_out.to_numpy(np.float32, nb // 4)
    np.save("/tmp/ds_{int(tile_only)}.npy", out)
'''
    subprocess.run([sys.executable, "-c", code], check=True)
for nc, nimg, kbm, W, golden in [(8, 64, "KB_FAST", 2.0, 1), (4, 24, "KB_EXACT", 2.0, 1), (6, 19, "KB_FAST", 1.5, 0), (8, 16, "KB_EXACT", 3.0, 1)]:
    run(False, nc, nimg, kbm, W, golden); run(True, nc, nimg, kbm, W, golden)
    a = np.load("/tmp/ds_0.npy"); b = np.load("/tmp/ds_1.npy")
    print(nc, nimg, kbm, W, "identical" if np.array_equal(a.view(np.uint32), b.view(np.uint32)) else f"DIFF max {np.abs(a-b).max()} rel {np.linalg.norm(a-b)/np.linalg.norm(b)}", float(np.abs(b).max()))
