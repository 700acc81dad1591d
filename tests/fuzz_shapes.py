"""Random-shape cross-check (tooling): the fast paths (binned gridding, tiled degridding with packed polynomials,
fused FFTs) against the reference-order exact mode on the GPU, and both against the CPU oracle when the case is
small enough.  usage: python tests/fuzz_shapes.py [ncases] [seed] [ncases of the round-2 feature sweep]   (test infrastructure: the only users of oracle/ live under tests/)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from tron_amd import lib
from oracle import pyoracle

def rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))

def run(n, seed, verbose=True, scan=False):
    """Returns (worst error, list of failing case descriptions).  scan: k-space under a scanner's envelope (synth.scan_envelope) and smooth
    images instead of flat random fields -- the samples next to the origin carry the result, and an error in one of them shows."""
    rng = np.random.default_rng(seed)
    worst = 0.0
    failures = []
    for it in range(n):
        adjoint = bool(rng.integers(0, 2))
        half = False
        nc = int(rng.choice([1, 2, 4, 6, 8]))
        W = float(rng.choice([1.5, 2.0, 2.0, 2.0, 2.5, 3.0, 3.5, 4.0]))   # > 3: the fallback kernels
        gridos = float(rng.choice([1.25, 1.5, 2.0, 2.0, 2.0]))
        golden = int(rng.integers(0, 2))
        skip = int(rng.integers(0, 50))
        if adjoint:
            nro = int(rng.choice([16, 24, 32, 48, 64, 96, 128, 200, 512]))
            npe_w = int(rng.integers(1, 700 if nro <= 128 else 60))
            nz = int(rng.integers(1, 5))
            slide = int(rng.integers(1, npe_w + 1))
            npe1 = npe_w + (nz - 1) * slide
            us = (npe_w + 0.5) / nro
            flags = dict(golden_angle=golden, data_undersamp=us, prof_slide=slide, kernwidth=W, gridos=gridos, skip_angles=skip)
            data = (synth.kspace_scan if scan else synth.kspace)(nc, nro, npe1, seed=1000 + it)
            half = bool(rng.integers(0, 5) == 0)             # complex-half storage (config 5)
            if half:
                h16 = np.asfortranarray(data).reshape(-1, order="F").view(np.float32).astype(np.float16)
                data = h16.astype(np.float32).view(np.complex64).reshape(data.shape, order="F")   # what the oracle sees
                gpu_in, flags = h16.reshape((2,) + data.shape, order="F"), dict(flags, input_half=1)
            desc = ("scan " if scan else "") + ("half " if half else "") + f"adj nc={nc} nro={nro} npe={npe_w} nz={nz} slide={slide} W={W} os={gridos} G={golden} skip={skip}"
            small = nro <= 128 and npe_w * nz <= 1500
        else:
            nx = int(rng.choice([8, 12, 16, 24, 32, 50, 64, 256]))
            nro = int(gridos * nx)
            npe = int(rng.integers(1, 300 if nx <= 64 else 40))
            us = (npe + 0.5) / nro
            flags = dict(golden_angle=golden, data_undersamp=us, kernwidth=W, gridos=gridos, skip_angles=skip)
            data = synth.image(nc, nx, seed=2000 + it)
            if scan:
                g = np.exp(-0.5 * ((np.arange(nx) - nx // 2) / (0.2 * nx)) ** 2).astype(np.float32)
                data = np.asfortranarray((data * (g[:, None] * g[None, :])[None, None, :, :, None]).astype(np.complex64))
            desc = f"fwd nc={nc} nx={nx} nro={nro} npe={npe} W={W} os={gridos} G={golden} skip={skip}"
            small = nx <= 64
        try:
            src = gpu_in if (adjoint and half) else data
            ex, d = lib.recon(src, adjoint=adjoint, kb_mode=lib.KB_EXACT, **flags)
            fa, _ = lib.recon(src, adjoint=adjoint, kb_mode=lib.KB_FAST, **flags)
            flags.pop("input_half", None)
        except Exception as e:
            print("ERROR", desc, str(e)[:200]); worst = 1.0; failures.append(desc + ": " + str(e)[:200]); continue
        e1 = rel(fa, ex)
        e2 = e3 = float("nan")
        if small:
            oflags = {("golden" if k == "golden_angle" else k): v for k, v in flags.items()}
            want, p = pyoracle.recon(data, adjoint=int(adjoint), **oflags)
            e2, e3 = rel(ex, want), rel(fa, want)
        bad = (not np.isfinite(e1)) or e1 > 5e-6 or (small and (e2 > 1e-5 or e3 > 1e-5))
        worst = max(worst, e1, 0 if not small else max(e2, e3))
        if bad:
            failures.append(f"{desc}: {e1:.2e} {e2:.2e} {e3:.2e}")
        if verbose:
            print(("BAD " if bad else "ok  ") + f"{desc}: fast-vs-exact {e1:.2e} exact-vs-oracle {e2:.2e} fast-vs-oracle {e3:.2e}", flush=True)
    return worst, failures


def run2(n, seed, verbose=True):
    """Round-2 features on random shapes, HIP (fast and exact) against the oracle: CGNR, Walsh combination, nt > 1,
    chunked host pipeline, split centre tiles / linear-angle slice groups (whatever the shape triggers)."""
    rng = np.random.default_rng(seed)
    worst, failures = 0.0, []
    for it in range(n):
        mode = str(rng.choice(["cgnr", "walsh", "nt", "plain"]))
        nc = int(rng.choice([1, 2, 4, 8])) if mode != "walsh" else int(rng.choice([2, 4, 6, 8]))
        nt = int(rng.integers(2, 4)) if mode == "nt" else 1
        golden = int(rng.integers(0, 2))
        nro = int(rng.choice([16, 24, 32, 48, 64]))
        npe_w = int(rng.integers(4, 120))
        nz = int(rng.integers(1, 12)) if mode != "cgnr" else int(rng.integers(1, 4))
        slide = int(rng.integers(1, npe_w + 1))
        flags = dict(golden_angle=golden, data_undersamp=(npe_w + 0.5) / nro, prof_slide=slide, skip_angles=int(rng.integers(0, 30)),
                     chunk_slices=int(rng.choice([0, 1, 2, 5])), kb_mode=int(rng.integers(0, 2)), pin_host=int(rng.integers(0, 2)))
        oflags = dict(golden=golden, data_undersamp=flags["data_undersamp"], prof_slide=slide, skip_angles=flags["skip_angles"])
        data = synth.kspace(nc, nro, npe_w + (nz - 1) * slide, seed=3000 + it, nt=nt)
        desc = f"{mode} nc={nc} nt={nt} nro={nro} npe={npe_w} nz={nz} slide={slide} G={golden} chunk={flags['chunk_slices']} kb={flags['kb_mode']}"
        try:
            if mode == "cgnr":
                niter, cons = int(rng.integers(1, 4)), int(rng.integers(0, 2))
                desc += f" niter={niter} consistent={cons}"
                want, _ = pyoracle.recon_cgnr(data, niter, consistent=cons, **oflags)
                got, _ = lib.recon(data, adjoint=True, niter=niter, cgnr_consistent=cons, **flags)
            elif mode == "walsh":
                npatch = int(rng.integers(0, 3))
                desc += f" npatch={npatch}"
                want, _ = pyoracle.recon_combine(data, 1, npatch, **oflags)
                got, _ = lib.recon(data, adjoint=True, coil_combine=1, walsh_patch=npatch, **flags)
            else:
                want, _ = pyoracle.recon_combine(data, 0, **oflags)
                got, _ = lib.recon(data, adjoint=True, **flags)
            e = rel(got, want) if got.shape == want.shape else float("inf")
        except Exception as ex:
            e, desc = float("inf"), desc + ": " + str(ex)[:200]
        # CGNR on inconsistent random data amplifies fp32 rounding by its (small-problem) conditioning: 5e-5 there
        tol = 5e-5 if mode == "cgnr" else 1e-5
        bad = not np.isfinite(e) or e > tol
        worst = max(worst, e / tol)
        if bad:
            failures.append(f"{desc}: {e:.2e}")
        if verbose:
            print(("BAD " if bad else "ok  ") + f"{desc}: {e:.2e}", flush=True)
    return worst, failures


if __name__ == "__main__":
    w, f = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("worst", w, "failures", len(f))
    w, f = run2(int(sys.argv[3]) if len(sys.argv) > 3 else (int(sys.argv[1]) if len(sys.argv) > 1 else 40), int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("round-2 features: worst (in units of the tolerance)", w, "failures", f)
