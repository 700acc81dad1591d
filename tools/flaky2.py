"""Hunt for run-to-run differences on a fresh box (kernel work tooling): the same reconstructions again and again, single plan and three
workers on one GPU, every result compared bit for bit with the first; prints where and by how much a run differs."""
import os, sys
os.environ.setdefault("TRON_TUNING", "1")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth
from tron_amd import lib

def report(tag, ref, got):
    if np.array_equal(ref, got):
        return 0
    d = np.abs(ref - got)
    idx = np.unravel_index(np.argmax(d), d.shape)
    bad = np.argwhere(d > 0)
    print(f"  {tag}: DIFFERS at {len(bad)} values, max {d.max():.3e} (|ref| max {np.abs(ref).max():.3e}) at {idx}; slices {sorted(set(bad[:, -1].tolist()))}; "
          f"rows {bad[:, 2].min()}..{bad[:, 2].max()} cols {bad[:, 3].min()}..{bad[:, 3].max()}", flush=True)
    return 1

cases = [
    ("2c 256 sliding", synth.kspace(2, 256, 90 + 30 * 8, seed=9300), dict(golden_angle=1, data_undersamp=0.3516, prof_slide=30), True),
    ("8c 1024", synth.kspace(8, 1024, 60, seed=10092), dict(golden_angle=1, data_undersamp=60.5 / 1024, prof_slide=60), False),
    ("8c 512x402 x6", synth.kspace(8, 512, 402 * 6, seed=77), dict(golden_angle=1, data_undersamp=0.7852, prof_slide=402), True),
]
nbad = 0
for name, data, fl, multi in cases:
    ref, _ = lib.recon(data, adjoint=True, **fl)
    print(name, "reference computed; finite:", bool(np.isfinite(ref).all()), flush=True)
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        got, _ = lib.recon(data, adjoint=True, **fl)
        nbad += report(f"{name} single #{rep}", ref, got)
        if multi:
            got, _ = lib.recon_multi(data, adjoint=True, devices=[0, 0, 0], **fl)
            nbad += report(f"{name} multi #{rep}", ref, got)
print("differing runs:", nbad)
