"""CGNR (`tron -a -i N`, src/tron.cu:665-720) on the GPU against the oracle's restatement of the same iteration
(oracle/tron_oracle.c: Knopp et al. 2007 Alg. 1 with the reference's operators and the five repairs F1-F5), plus the
properties the algorithm must have: N = 0 is the plain adjoint, the weighted residual falls with every iteration."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import rel_l2
import synth
from tron_amd import lib, ra

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRON = os.path.join(ROOT, "tron_amd", "bin", "tron")
TOL = 1e-5


def _consistent_data(oracle, nc, nx, golden, seed, **fwd_flags):
    """k-space of a random image through the oracle's forward operator, so that CGNR has something to converge to."""
    img = synth.image(nc, nx, seed=seed)
    data, p = oracle.recon(img, adjoint=0, golden=golden, **fwd_flags)
    return img, np.asfortranarray(data.reshape((nc, 1, p.nro, p.npe1work, 1), order="F"))


@pytest.mark.parametrize("nc,niter,kb", [(1, 1, lib.KB_EXACT), (1, 4, lib.KB_EXACT), (2, 3, lib.KB_FAST), (8, 5, lib.KB_FAST)])
def test_cgnr_golden_angle_vs_oracle(oracle, nc, niter, kb):
    _, data = _consistent_data(oracle, nc, 32, 1, 1401 + nc)
    want, p = oracle.recon_cgnr(data, niter, golden=1)
    got, dims = lib.recon(data, adjoint=True, golden_angle=1, niter=niter, kb_mode=kb)
    assert got.shape == want.shape and dims.nx == 32
    assert rel_l2(got, want) <= TOL


def test_cgnr_sliding_windows_multi_slice_vs_oracle(oracle):
    """Several slices advance through the iteration together, each with its own step sizes and its own golden angles
    (skip_angles + z*prof_slide in BOTH operators, F4)."""
    data = synth.kspace(2, 64, 110, seed=1410)
    flags = dict(data_undersamp=0.5, prof_slide=13, skip_angles=4)
    want, p = oracle.recon_cgnr(data, 3, golden=1, **flags)
    got, dims = lib.recon(data, adjoint=True, golden_angle=1, niter=3, **flags)
    assert dims.nz == p.nz == 7
    for z in range(dims.nz):
        assert rel_l2(got[..., z], want[..., z]) <= TOL, z
    got2, _ = lib.recon(data, adjoint=True, golden_angle=1, niter=3, chunk_slices=2, **flags)      # chunked: same bytes
    assert np.array_equal(got, got2)


@pytest.mark.parametrize("consistent", [0, 1])
def test_cgnr_linear_angles_both_conventions_vs_oracle(oracle, consistent):
    """Q5: the reference's gridding (src/tron.cu:509) and degridding (:555) kernels disagree on the linear-angle
    convention.  cgnr_consistent = 0 keeps both as they are (what the reference's CGNR would run), 1 gives the
    iteration a matched pair; the oracle restates both."""
    _, data = _consistent_data(oracle, 1, 32, 0, 1420)
    want, _ = oracle.recon_cgnr(data, 3, consistent=consistent, golden=0)
    got, _ = lib.recon(data, adjoint=True, golden_angle=0, niter=3, cgnr_consistent=consistent)
    assert rel_l2(got, want) <= TOL


def test_cgnr_zero_iterations_is_the_adjoint_and_residual_decreases(oracle):
    img, data = _consistent_data(oracle, 2, 32, 1, 1430)
    adj, _ = lib.recon(data, adjoint=True, golden_angle=1)
    it0, _ = lib.recon(data, adjoint=True, golden_angle=1, niter=0)
    assert np.array_equal(adj, it0)
    # weighted residual || W^(1/2) (y - A x_k) ||: non-increasing in k for CGNR; x_k from the device-resident entry point
    cfg = lib.default_config(adjoint=1, golden_angle=1)
    nro, npe = data.shape[2], data.shape[3]
    w = (2.0 - 2.0 / npe) / nro * np.abs(np.arange(nro) - nro // 2) + 1.0 / npe
    res = []
    for k in (1, 2, 4, 8):
        cfg = lib.default_config(adjoint=1, golden_angle=1, niter=k)
        dims = lib.derive_dims(cfg, data.shape)
        with lib.Plan(cfg, dims) as plan:
            d_in = lib.DeviceBuffer.from_numpy(np.asfortranarray(data).reshape(-1, order="F"))
            d_out = lib.DeviceBuffer(2 * 32 * 32 * 8)
            plan.cgnr_device(d_out.ptr, d_in.ptr, 0, 1, combine=0)
            plan.sync()
            x = d_out.to_numpy(np.complex64, 2 * 32 * 32).reshape(32, 32, 2)           # [row][col][coil]
        xi = np.transpose(x, (2, 1, 0)).reshape((2, 1, 32, 32, 1), order="F")            # file order (nc, nt, nx, ny, nz): col fastest
        y, _ = oracle.recon(xi, adjoint=0, golden=1)
        r = (y.reshape(data.shape, order="F") - data)[:, 0, :, :, 0]
        res.append(float(np.sqrt(np.sum(w[None, :, None] * np.abs(r) ** 2))))
    assert all(b < a for a, b in zip(res, res[1:])), res
    assert res[-1] < 0.35 * res[0], res


def test_cgnr_cli(oracle, tmp_path):
    _, data = _consistent_data(oracle, 2, 32, 1, 1440)
    want, _ = oracle.recon_cgnr(data, 2, golden=1)
    src, dst = str(tmp_path / "in.ra"), str(tmp_path / "out.ra")
    ra.write(src, data)
    r = subprocess.run([TRON, "-a", "-G", "-i", "2", src, dst], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert rel_l2(ra.read(dst), want) <= TOL
    r = subprocess.run([TRON, "-G", "-i", "2", src, dst], capture_output=True, text=True, timeout=120)   # forward + -i: ignored as in the reference
    assert r.returncode in (0, 1, 2)
