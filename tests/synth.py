"""Seeded synthetic inputs: complex64, re/im i.i.d. uniform [-1,1) from a
splitmix64 counter generator, so every test, fixture script and the bench agree."""
import os

import numpy as np

SEED_BASE = 0x54524F4E  # "TRON"


def splitmix64(idx, seed):
    z = (idx.astype(np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def uniform_c64(n, seed=SEED_BASE):
    """n complex64 values with re, im uniform in [-1, 1) (24-bit mantissas)."""
    with np.errstate(over="ignore"):
        bits = splitmix64(np.arange(2 * n, dtype=np.uint64), seed)
    u = (bits >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)
    return u.view(np.complex64)


def kspace(nc, nro, npe1, seed=SEED_BASE, nt=1, npe2=1):
    """Radial k-space shaped like the .ra file: (nc, nt, nro, npe1, npe2), first dim fastest."""
    flat = uniform_c64(nc * nt * nro * npe1 * npe2, seed)
    d = flat.reshape((nc, nt, nro, npe1, npe2), order="F")
    if os.environ.get("TRON_TEST_SCAN") == "1":      # exploratory: every test's k-space under scan_envelope() (thresholds are tuned for flat fields: read the failures, do not gate on them)
        d = np.asfortranarray((d * scan_envelope(nro)[None, None, :, None, None]).astype(np.complex64))
    return d


def scan_envelope(nro):
    """Magnitude of k-space as it comes off a scanner, along the readout: 1 at the centre sample, falling as 1 / (1 + (r / 2)^2) to a
    noise floor of 1e-4 -- what the flat random field above hides: the samples next to the origin carry nearly all of the energy, so an
    error in any one of them shows in the image (round 6: the centre kernel's window edge, tests/test_gpu_arc.py)."""
    r = np.abs(np.arange(nro) - nro // 2).astype(np.float32)
    return (1.0 / (1.0 + (r / 2.0) ** 2) + 1e-4).astype(np.float32)


def kspace_scan(nc, nro, npe1, seed=SEED_BASE, nt=1, npe2=1):
    """kspace() under scan_envelope()."""
    d = kspace(nc, nro, npe1, seed=seed, nt=nt, npe2=npe2)
    return np.asfortranarray((d * scan_envelope(nro)[None, None, :, None, None]).astype(np.complex64))


def image(nc, nx, seed=SEED_BASE + 1, nz=1):
    flat = uniform_c64(nc * nx * nx * nz, seed)
    return flat.reshape((nc, 1, nx, nx, nz), order="F")
