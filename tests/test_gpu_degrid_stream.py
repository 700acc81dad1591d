"""degrid_stream_kernel (tron_degrid_stream.hip): launches of 16 images and more on grids of whole 32x32 tiles.

One workgroup walks a run of images of its tile, the next tile arriving by LDS-DMA while the current one is sampled.  The
sample loop is the tile kernel's (tron_degrid_sample.h), so the two kernels must agree BIT FOR BIT on any input; the
oracle (degridradial2d, src/tron.cu:540-577 restated) is the checker for both."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import synth
from conftest import rel_l2
from tron_amd import lib

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOL_PIPELINE = 1e-5


def _forward(imgs, nimg, **flags):
    """nimg images through one plan; returns (samples per image as complex64 [nimg, per], kernel name)."""
    cfg = lib.default_config(adjoint=0, **flags)
    dims = lib.derive_dims(cfg, imgs[0].shape)
    per = imgs[0].shape[0] * dims.nro * dims.npe1work
    flat = np.concatenate([np.asfortranarray(im).reshape(-1, order="F") for im in imgs])
    with lib.Plan(cfg, dims) as plan:
        d_in = lib.DeviceBuffer.from_numpy(flat)
        d_out = lib.DeviceBuffer(nimg * per * 8)
        plan.forward_device(d_out.ptr, d_in.ptr, nimg)
        plan.sync()
        name = plan.degrid_kernel_name()
        got = d_out.to_numpy(np.complex64, nimg * per)
    return got.reshape(nimg, per), name


CASES = [
    # nc, nx, nimg, kb, W, golden, undersamp (None: default), what it covers
    (8, 256, 32, lib.KB_FAST, 2.0, 1, 64 / 512 + 1e-6, "metric shape: fused FFT (transposed grid), runs of 8"),
    (4, 256, 17, lib.KB_EXACT, 2.0, 1, 48 / 512 + 1e-6, "exact weights; a last run of one image"),
    (6, 256, 19, lib.KB_FAST, 1.5, 0, 48 / 512 + 1e-6, "a tail chunk of two coils; linear angles; fractional W"),
    (4, 128, 16, lib.KB_EXACT, 3.0, 1, None, "ceil(W) = 3: the 40-row tile at pitch 44"),
    (4, 128, 16, lib.KB_EXACT, 4.0, 1, None, "ceil(W) = 4"),
    (4, 192, 16, lib.KB_FAST, 2.0, 1, 64 / 384 + 1e-6, "384^2 grid: rocFFT, grid not transposed"),
    (4, 256, 16, lib.KB_FAST, 2.0, 1, 804.5 / 512, "804 spokes: two clipping rounds per image"),
    # round 5: the kept records are dealt by the bank class of their footprint's first point (rows of 32 classes, padded while wide)
    (8, 256, 16, lib.KB_FAST, 2.0, 1, None, "512 spokes: padded rows; lists longer than the kept passes round the centre"),
    (4, 256, 16, lib.KB_FAST, 2.0, 0, 300.5 / 512, "300 linear spokes: rows with holes next to rows without"),
    (4, 256, 16, lib.KB_FAST, 2.0, 1, 8.5 / 512, "8 spokes: a few records per tile, every row narrow"),
    (4, 256, 16, lib.KB_FAST, 3.0, 1, 200.5 / 512, "ceil(W) = 3 with the fast weights: the 40-row tile's classes"),
]


@pytest.mark.parametrize("nc,nx,nimg,kb,W,golden,us,what", CASES, ids=[c[-1].split(":")[0].split(";")[0] for c in CASES])
def test_stream_kernel_equals_tile_kernel_bit_for_bit(monkeypatch, nc, nx, nimg, kb, W, golden, us, what):
    imgs = [synth.image(nc, nx, seed=9100 + k) for k in range(nimg)]
    flags = dict(golden_angle=golden, kernwidth=W, kb_mode=kb)
    if us is not None:
        flags["data_undersamp"] = us
    got, name = _forward(imgs, nimg, **flags)
    assert name == "degrid_stream_kernel", what
    monkeypatch.setenv("TRON_DEGRID_KERNEL", "tile")
    ref, rname = _forward(imgs, nimg, **flags)
    assert rname == "degrid_tile_kernel"
    assert np.isfinite(got.view(np.float32)).all()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), what


@pytest.mark.parametrize("kb", [lib.KB_EXACT, lib.KB_FAST])
def test_stream_kernel_vs_oracle(oracle, kb):
    """Every image of a 16-image launch against the oracle's forward transform of that image (first, middle, last)."""
    nc, nimg = 4, 16
    imgs = [synth.image(nc, 256, seed=9300 + k) for k in range(nimg)]
    us = 40 / 512 + 1e-6
    got, name = _forward(imgs, nimg, golden_angle=1, data_undersamp=us, kb_mode=kb)
    assert name == "degrid_stream_kernel"
    for k in (0, 7, nimg - 1):
        want, _ = oracle.recon(imgs[k], adjoint=0, golden=1, data_undersamp=us)
        assert rel_l2(got[k], want.reshape(-1, order="F")) <= TOL_PIPELINE, k


@pytest.mark.parametrize("nc", [2, 6, 10, 12, 16])
def test_fused_forward_fft_coil_counts_vs_oracle(oracle, nc):
    """The fused 256 -> 512 forward FFT passes (tron_fft512.hip: next block by LDS-DMA beside the transform) take `rows` image
    rows x all coils per step, rows = 16 // nc (8, 2, 1, 1, 1 here): coil counts that do and do not divide 16 (odd counts other than 1: src/tron.cu:963 rejects them)."""
    img = synth.image(nc, 256, seed=9600 + nc)
    us = 24 / 512 + 1e-6
    want, p = oracle.recon(img, adjoint=0, golden=1, data_undersamp=us)
    assert (p.nxos, p.nx) == (512, 256)
    got, _ = lib.recon(img, adjoint=False, golden_angle=1, data_undersamp=us)
    assert rel_l2(got, want) <= TOL_PIPELINE


def _forward_in_child(imgs, nimg, env, **flags):
    """_forward in a child process: the switches below are read once per process."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import test_gpu_degrid_stream as t\n"
        "d = np.load(sys.argv[1]); imgs = [np.asfortranarray(d[..., k:k+1]) for k in range(d.shape[-1])]\n"
        "out, name = t._forward(imgs, len(imgs), **eval(sys.argv[3])); np.save(sys.argv[2], out.view(np.float32))\n"
        % (ROOT, os.path.join(ROOT, "tests")))
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "in.npy"), np.concatenate(imgs, axis=-1))
        r = subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "in.npy"), os.path.join(tmp, "out.npy"), repr(flags)],
                           env=dict(os.environ, TRON_TUNING="1", **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return np.load(os.path.join(tmp, "out.npy"))


def test_forward_switches_change_no_bit():
    """The forward grid's line rotation (DegridParams::in_rot) only moves points in memory, and the tile kernel
    samples what the streaming kernel samples: every switch must leave the samples bit-identical."""
    nc, nimg = 8, 16
    imgs = [synth.image(nc, 256, seed=9700 + k) for k in range(nimg)]
    flags = dict(golden_angle=1, data_undersamp=32 / 512 + 1e-6)
    ref = _forward_in_child(imgs, nimg, {}, **flags)
    assert np.isfinite(ref).all()
    for env in ({"TRON_GRID_ROT": "0"}, {"TRON_GRID_ROT": "4"}, {"TRON_DEGRID_KERNEL": "tile", "TRON_GRID_ROT": "0"}, {"TRON_DEGRID_KERNEL": "simple"}):
        got = _forward_in_child(imgs, nimg, env, **flags)
        if env.get("TRON_DEGRID_KERNEL") == "simple":         # the thread-per-sample kernel: the exact weights' order, fast weights differ in the last bits
            assert rel_l2(got, ref) <= 2e-6, env
        else:
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), env


def test_small_launches_stay_on_the_tile_kernel():
    """Fewer than 16 images leave too few workgroups for runs of images: the tile kernel takes them."""
    imgs = [synth.image(4, 256, seed=9400 + k) for k in range(8)]
    _, name = _forward(imgs, 8, golden_angle=1, data_undersamp=24 / 512 + 1e-6)
    assert name == "degrid_tile_kernel"


def test_degrid_stage_linearity_at_metric_size():
    """Size-independent property at BASELINE's forward shape (64 images x 8 coils, 512 x 512 spokes): A(a x + y) = a A x + A y
    to rounding, through the streaming kernel."""
    nc, nimg = 8, 64
    rng = np.random.default_rng(9500)
    x = (rng.standard_normal((nimg, 2 * nc * 256 * 256), dtype=np.float32))
    y = (rng.standard_normal((nimg, 2 * nc * 256 * 256), dtype=np.float32))
    a = np.float32(0.75)
    cfg = lib.default_config(adjoint=0, golden_angle=1)
    dims = lib.derive_dims(cfg, (nc, 1, 256, 256, 1))
    per = nc * dims.nro * dims.npe1work
    outs = []
    with lib.Plan(cfg, dims) as plan:
        d_out = lib.DeviceBuffer(nimg * per * 8)
        for v in (x, y, a * x + y):
            d_in = lib.DeviceBuffer.from_numpy(v.reshape(-1))
            plan.forward_device(d_out.ptr, d_in.ptr, nimg)
            plan.sync()
            outs.append(d_out.to_numpy(np.float32, nimg * per * 2).astype(np.float64))
            d_in.free()
        assert plan.degrid_kernel_name() == "degrid_stream_kernel"
    ax, ay, axy = outs
    err = np.linalg.norm(axy - (a * ax + ay)) / np.linalg.norm(axy)
    assert err <= 2e-6
