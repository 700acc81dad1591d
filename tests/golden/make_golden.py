"""Regenerates the fixtures in this directory (run from the repo root, in the build container).

  ref_written_c64.ra, ref_written_f32.ra   written by the REFERENCE's own ra_write (src/ra.cu:131-162,
                                           compiled unmodified into oracle/_ref) from seeded arrays
  half_vectors.npz                         float32/float64 bit patterns and the half bits the REFERENCE's
                                           src/float16.cu returns for them; all 65536 half->float results
  cgnr_*.ra / walsh_*.ra / nt_*.ra         (round 2) the same for CGNR (src/tron.cu:665-720 as restated in oracle/), the Walsh
                                           coil combination (:222-302) and nt > 1 -- written by `make_golden.py round2`
  adj_*.ra / fwd_*.ra                      seeded inputs and the outputs of the CPU oracle (oracle/), i.e.
                                           regression vectors for the grid/degrid path.  The reference holds
                                           no golden vectors for this path and src/tron.cu cannot be built here
                                           (DESIGN.md section 2), so these pin the ORACLE, not the reference.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import synth                       # noqa: E402
from oracle import pyoracle        # noqa: E402
from tron_amd import ra            # noqa: E402


def ref_write(path, arr, eltype, elbyte):
    R = pyoracle.ref()
    dims = (ctypes.c_uint64 * arr.ndim)(*arr.shape)
    payload = np.asfortranarray(arr).tobytes(order="F")
    buf = (ctypes.c_uint8 * len(payload)).from_buffer_copy(payload)
    a = pyoracle.RaT(0, eltype, elbyte, len(payload), arr.ndim, dims, ctypes.cast(buf, ctypes.POINTER(ctypes.c_uint8)))
    assert R.ra_write(ctypes.byref(a), path.encode()) == 0


def round2():
    """Fixtures of the round-2 features; the round-1 files are left untouched."""
    img = synth.image(2, 16, seed=911)
    kdata, p = pyoracle.recon(img, 0, golden=1)                      # consistent k-space: CGNR has something to converge to
    kdata = np.asfortranarray(kdata.reshape((2, 1, p.nro, p.npe1work, 1), order="F"))
    cases = {
        "cgnr3_ga_nc2": (kdata, lambda d: pyoracle.recon_cgnr(d, 3, golden=1)[0]),
        "walsh_ga_nc4": (synth.kspace(4, 32, 30, seed=912), lambda d: pyoracle.recon_combine(d, 1, 1, golden=1)[0]),
        "nt2_ga_nc2": (synth.kspace(2, 32, 44, seed=913, nt=2), lambda d: pyoracle.recon_combine(d, 0, golden=1, data_undersamp=0.5, prof_slide=14)[0]),
    }
    for name, (data, fn) in cases.items():
        out = fn(data)
        ra.write(os.path.join(HERE, name + "_in.ra"), data)
        ra.write(os.path.join(HERE, name + "_out.ra"), out)
        print(name, data.shape, "->", out.shape)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        return round2()
    assert pyoracle.have_ref(), "build oracle/_ref first (make -C oracle)"
    R = pyoracle.ref()
    ref_write(os.path.join(HERE, "ref_written_c64.ra"), synth.kspace(2, 8, 5, seed=901), 4, 8)
    ref_write(os.path.join(HERE, "ref_written_f32.ra"), np.arange(60, dtype=np.float32).reshape(3, 4, 5) / 7, 3, 4)

    rng = np.random.default_rng(902)
    f32 = np.concatenate([
        np.array([0x00000000, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0x7f800001, 0xffffffff, 0x477fe000, 0x477fefff,
                  0x477ff000, 0x47800000, 0x38800000, 0x387fffff, 0x33000000, 0x33000001, 0x32ffffff, 0x3f801000, 0x3f803000], np.uint32),
        rng.integers(0, 2 ** 32, 4000, dtype=np.uint64).astype(np.uint32),
        (0x33000000 + rng.integers(0, 0x05800000, 4000)).astype(np.uint32)])
    f64 = np.concatenate([rng.integers(0, 2 ** 64, 2000, dtype=np.uint64),
                          (np.float64(2.0) ** rng.uniform(-26, 17, 2000) * rng.choice([-1, 1], 2000)).view(np.uint64)])
    np.savez_compressed(os.path.join(HERE, "half_vectors.npz"),
                        f32_bits=f32, f32_to_half=np.array([R.f2h(int(x)) for x in f32], np.uint16),
                        f64_bits=f64, f64_to_half=np.array([R.d2h(int(x)) for x in f64], np.uint16),
                        half_to_f32=np.array([R.h2f(h) for h in range(65536)], np.uint32))

    cases = {
        "adj_ga_nc2_slide": (synth.kspace(2, 32, 60, seed=903), 1, dict(golden=1, data_undersamp=0.5, prof_slide=11, skip_angles=2)),
        "adj_lin_nc1": (synth.kspace(1, 32, 24, seed=904), 1, dict(golden=0, data_undersamp=2.0)),
        "fwd_lin_nc1": (synth.image(1, 16, seed=905), 0, dict()),
        "fwd_ga_nc2": (synth.image(2, 16, seed=906), 0, dict(golden=1, data_undersamp=0.5)),
    }
    for name, (data, adjoint, flags) in cases.items():
        out, _ = pyoracle.recon(data, adjoint, **flags)
        ra.write(os.path.join(HERE, name + "_in.ra"), data)
        ra.write(os.path.join(HERE, name + "_out.ra"), out)
        print(name, data.shape, "->", out.shape)


if __name__ == "__main__":
    main()
