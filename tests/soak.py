"""Soak (tooling; `python tests/soak.py [operations] [seed]`, or tests/test_gpu_stress.py for a short one): plans of random shapes created,
used, retargeted and destroyed in random order in ONE process, host arrays of every size on the heap and in mappings of their own --
whatever a plan returns must be, bit for bit, what a plan created fresh for that job returns.  No oracle: the parity tests own the
numbers; this looks for state that leaks from one plan, call or buffer into the next (round 6 found two such things by accident: a
registered heap buffer, tests/test_gpu_round2.py, and a table position one ulp outside its table, branch scatter-fold)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from tron_amd import lib


def flat(a):
    return np.asfortranarray(a).reshape(-1, order="F")


class Job:
    def __init__(self, rng, k):
        self.adjoint = bool(rng.integers(0, 5) != 0)
        self.nc = int(rng.choice([1, 2, 4, 6, 8]))
        self.golden = int(rng.integers(0, 4) != 0)
        self.skip = int(rng.integers(0, 4000))
        if self.adjoint:
            self.nro = int(rng.choice([64, 128, 256]))
            self.npe = int(rng.integers(8, 420 if self.nro >= 128 else 120))
            self.nz = int(rng.integers(1, 7))
            self.slide = int(rng.integers(1, self.npe + 1))
            maker = synth.kspace_scan if rng.integers(0, 2) else synth.kspace
            self.data = maker(self.nc, self.nro, self.npe + (self.nz - 1) * self.slide, seed=7000 + k)
            self.half = bool(rng.integers(0, 4) == 0)
            if rng.integers(0, 3) == 0 and not self.half:
                self.data = np.asfortranarray(self.data * np.float32(2.0 ** int(rng.integers(-40, 41))))
            self.flags = dict(golden_angle=self.golden, data_undersamp=(self.npe + 0.5) / self.nro, prof_slide=self.slide,
                              chunk_slices=int(rng.choice([0, 0, 1, 2])), pin_host=int(rng.integers(0, 2)))
            # now and then: CGNR, the Walsh combination, the exact kernels, another oversampling ratio, another window width
            pick = int(rng.integers(0, 12))
            if pick == 0 and self.nc > 1:
                self.flags.update(coil_combine=1, walsh_patch=int(rng.integers(0, 3)))
            elif pick == 1 and not self.half:                       # (CGNR takes complex64 k-space)
                self.flags.update(niter=int(rng.integers(1, 4)), cgnr_consistent=int(rng.integers(0, 2)))
            elif pick == 2:
                self.flags.update(kb_mode=lib.KB_EXACT)
            elif pick == 3:
                self.flags.update(gridos=float(rng.choice([1.25, 1.5, 3.0])))
            elif pick == 4:
                self.flags.update(kernwidth=float(rng.choice([1.5, 2.5, 3.0, 4.0])))
            if self.half:
                self.src = np.stack([self.data.real, self.data.imag]).astype(np.float16)
                self.flags["input_half"] = 1
            else:
                self.src = self.data
        else:
            self.nx = int(rng.choice([32, 64, 128, 256]))
            self.half = False
            self.data = synth.image(self.nc, self.nx, seed=8000 + k)
            self.src = self.data
            self.flags = dict(golden_angle=self.golden)
            if rng.integers(0, 4) == 0:
                self.flags.update(kernwidth=float(rng.choice([1.5, 2.5, 3.0])))
        self.extra = {k: v for k, v in self.flags.items() if k in ("coil_combine", "walsh_patch", "niter", "cgnr_consistent", "kb_mode", "gridos", "kernwidth")}
        self.desc = (f"{'adj' if self.adjoint else 'fwd'} nc={self.nc} G={self.golden} " +
                     (f"nro={self.nro} npe={self.npe} nz={self.nz} slide={self.slide} half={int(self.half)} chunk={self.flags['chunk_slices']} pin={self.flags['pin_host']}"
                      if self.adjoint else f"nx={self.nx}") + (f" {self.extra}" if self.extra else ""))

    def config(self, skip):
        return lib.default_config(adjoint=int(self.adjoint), skip_angles=skip, **self.flags)

    def fresh(self, skip):
        return lib.recon(self.src, adjoint=self.adjoint, skip_angles=skip, **self.flags)[0]


def run(nops, seed, verbose=False):
    rng = np.random.default_rng(seed)
    live = []                                            # (job, plan, skip it is at)
    failures = []
    for k in range(nops):
        op = int(rng.integers(0, 10))
        if (op <= 2 and len(live) < 4) or not live:
            job = Job(rng, k)
            cfg = job.config(job.skip)
            dims = lib.derive_dims(cfg, job.src.shape[1:] if job.half else job.src.shape)
            live.append([job, lib.Plan(cfg, dims), job.skip, dims])
            what = "create " + job.desc
        elif op <= 4 and any(e[0].adjoint and e[0].golden for e in live):
            cand = [e for e in live if e[0].adjoint and e[0].golden]
            e = cand[int(rng.integers(0, len(cand)))]
            e[2] = int(rng.integers(0, 6000))
            e[1].retarget(e[2])
            what = f"retarget -> {e[2]} " + e[0].desc
        elif op == 5 and len(live) > 1:
            e = live.pop(int(rng.integers(0, len(live))))
            e[1].close()
            what = "destroy " + e[0].desc
        else:
            e = live[int(rng.integers(0, len(live)))]
            job, plan, skip, dims = e
            z0, zc = 0, dims.nz
            if job.adjoint and dims.nz > 1 and rng.integers(0, 3) == 0:       # a slice sub-range of the full buffers (tron_recon_radial2d_range)
                z0 = int(rng.integers(0, dims.nz)); zc = int(rng.integers(1, dims.nz - z0 + 1))
            got = plan.recon(_flat_in(job), zfirst=z0, zcount=zc, out=np.zeros(dims.out_bytes // 8, np.complex64))
            want = flat(job.fresh(skip))
            if (z0, zc) != (0, dims.nz):
                img = dims.out_bytes // 8 // dims.nz
                got, want = got[z0 * img:(z0 + zc) * img], want[z0 * img:(z0 + zc) * img]
            what = f"run slices [{z0}, {z0 + zc}) at skip {skip} " + job.desc
            # (linear angles, at most four channels: several slices share a pass of the binned kernel, DESIGN.md 4.2, and a slice's sums are
            #  taken in the order its GROUP's batches come in -- its bits follow the grouping at the level of fp32 rounding; every other
            #  plan returns the same bits however the slices are asked for)
            grouped = job.adjoint and not job.golden and job.nc <= 4 and dims.nz > 1
            same = np.array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
            if grouped and not same:
                same = float(np.linalg.norm(got - want)) <= 2e-6 * float(np.linalg.norm(want))
            if not same:
                d = np.abs(got - want)
                failures.append(f"op {k}: {what}: max |diff| {np.nanmax(d):.3e} of {np.nanmax(np.abs(want)):.3e}, nan {int(np.isnan(got).sum())}/{int(np.isnan(want).sum())}")
        if verbose:
            print(k, what, flush=True)
    for e in live:
        e[1].close()
    return failures


def _flat_in(job):
    return np.asfortranarray(job.src).reshape(-1, order="F")          # (complex-half: (re, im) is the first axis, as lib.recon takes it)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    f = run(n, seed, verbose="-v" in sys.argv)
    print("soak:", n, "operations, seed", seed, "failures", len(f))
    for x in f:
        print("  ", x)
    sys.exit(1 if f else 0)
