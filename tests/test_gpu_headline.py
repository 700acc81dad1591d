"""The headline shapes of BASELINE.json at their stated sizes, HIP path (through the C ABI, host-buffer entry
point) against the CPU oracle on more than one slice of a launch-sized batch:

  (i)   metric shape: 512 readout x 402 golden-angle spokes x 8 coils, 64 slices in one launch (the real record
        batch size, centre-tile clip rounds and disc skipping are in play), slices first / middle / last;
  (ii)  BASELINE config 4's slice shape: 512 x 804 spokes x 8 coils (two clip rounds per tile);
  (iii) config 5 at its stated size: complex-half k-space, 512^2 grid, 8 coils;
  (iv)  config 2 / RUNME1+RUNME3 shape: 256^2 image -> 512 x 512 LINEAR spokes, 1 coil, through the fused forward
        FFT, and the adjoint of that data back to 256^2.

Tolerance: north_star's 1e-5 relative L2 per slice (src/tron.cu:465-536, 540-577 via the oracle).
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import rel_l2
import synth
from tron_amd import lib

pytestmark = pytest.mark.gpu

TOL = 1e-5
NRO = 512
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_slices(oracle, data, got, slices, **oflags):
    worst = 0.0
    for z in slices:
        want, _ = oracle.recon(data, adjoint=1, zfirst=z, zcount=1, **oflags)
        err = rel_l2(got[..., z], want[..., z])
        assert err <= TOL, (z, err)
        worst = max(worst, err)
    return worst


@pytest.fixture(scope="module")
def metric_stream():
    """64 slices x 8 coils x 512 x 402: one full default-size launch of the gridding kernel (843 MB)."""
    return synth.kspace(8, NRO, 402 * 64, seed=synth.SEED_BASE + 21)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kb", [lib.KB_FAST, lib.KB_EXACT])
def test_metric_shape_402_spokes_8_coils_64_slices(oracle, metric_stream, kb):
    flags = dict(golden_angle=1, data_undersamp=0.7852, prof_slide=402)        # tron -a -G -u 0.7852 -d 402
    got, dims = lib.recon(metric_stream, adjoint=True, kb_mode=kb, **flags)
    assert (dims.nz, dims.npe1work, dims.nxos, dims.nx) == (64, 402, 512, 256)
    assert np.isfinite(got).all()
    _check_slices(oracle, metric_stream, got, (0, 31, 63), golden=1, data_undersamp=0.7852, prof_slide=402)


def test_metric_shape_on_k_space_whose_energy_sits_at_the_centre(oracle):
    """The metric's slice shape (512 x 402 golden-angle spokes x 8 coils) on k-space with a scanner's envelope (synth.scan_envelope): the
    first slice holds spoke 305, which runs 5e-4 rad off the ky axis -- the case the centre kernel had wrong until round 6
    (tests/test_gpu_arc.py) -- and the samples |r| < 14 carry the image."""
    data = synth.kspace_scan(8, NRO, 402 * 2, seed=synth.SEED_BASE + 27)
    flags = dict(golden_angle=1, data_undersamp=0.7852, prof_slide=402)
    got, dims = lib.recon(data, adjoint=True, **flags)
    assert (dims.nz, dims.npe1work) == (2, 402)
    worst = _check_slices(oracle, data, got, (0, 1), golden=1, data_undersamp=0.7852, prof_slide=402)
    assert worst <= 3e-6, worst


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kb", [lib.KB_FAST, lib.KB_EXACT])
def test_config4_shape_804_spokes_8_coils(oracle, kb):
    nz = 32                                                                     # config 4 over 8 GPUs = 32 slices per GPU
    data = synth.kspace(8, NRO, 804 * nz, seed=synth.SEED_BASE + 22)
    flags = dict(golden_angle=1, data_undersamp=1.5704, prof_slide=804)        # 512 * 1.5704f -> 804
    got, dims = lib.recon(data, adjoint=True, kb_mode=kb, **flags)
    assert (dims.nz, dims.npe1work) == (nz, 804)
    _check_slices(oracle, data, got, (0, nz - 1), golden=1, data_undersamp=1.5704, prof_slide=804)


@pytest.mark.timeout(900)
def test_config4_eight_workers_on_one_gpu_give_the_single_plan_bytes(tmp_path):
    """BASELINE config 4's decomposition without an 8-GPU node: 804 spokes x 8 coils, the slices sharded over EIGHT workers
    (threads, one plan each, contiguous slice blocks, global angle index: src/tron.cu:582-597, 735-736) that all sit on GPU 0,
    through `tron -g 0,0,0,0,0,0,0,0` and through tron_recon_radial2d_multi: the bytes of one plan."""
    from tron_amd import ra
    nz = 24
    data = synth.kspace(8, NRO, 804 * nz, seed=synth.SEED_BASE + 26)
    flags = dict(golden_angle=1, data_undersamp=1.5704, prof_slide=804)
    one, dims = lib.recon(data, adjoint=True, **flags)
    assert (dims.nz, dims.npe1work) == (nz, 804)
    multi, _ = lib.recon_multi(data, adjoint=True, devices=[0] * 8, **flags)
    assert np.array_equal(one, multi)
    src, a, b = (str(tmp_path / n) for n in ("in.ra", "a.ra", "b.ra"))
    ra.write(src, data)
    tron = os.path.join(ROOT, "tron_amd", "bin", "tron")
    args = ["-a", "-G", "-u", "1.5704", "-d", "804", src]
    assert subprocess.run([tron] + args + [a], timeout=300).returncode == 0
    assert subprocess.run([tron, "-g", "0,0,0,0,0,0,0,0"] + args + [b], timeout=300).returncode == 0
    assert open(a, "rb").read() == open(b, "rb").read()
    assert np.array_equal(ra.read(a), one)


@pytest.mark.timeout(900)
def test_config5_complex_half_at_512_grid_8_coils(oracle):
    nz = 16
    data = synth.kspace(8, NRO, 402 * nz, seed=synth.SEED_BASE + 23)
    halves = np.asfortranarray(data).reshape(-1, order="F").view(np.float32).astype(np.float16)   # RNE, as src/float16.cu
    rounded = halves.astype(np.float32).view(np.complex64).reshape(data.shape, order="F")
    flags = dict(golden_angle=1, data_undersamp=0.7852, prof_slide=402)
    h = halves.reshape((2,) + data.shape, order="F")
    for kb in (lib.KB_FAST, lib.KB_EXACT):
        got, dims = lib.recon(h, adjoint=True, input_half=1, kb_mode=kb, **flags)
        assert dims.nz == nz
        _check_slices(oracle, rounded, got, (0, nz - 1), golden=1, data_undersamp=0.7852, prof_slide=402)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kb", [lib.KB_FAST, lib.KB_EXACT])
def test_config2_linear_256_image_to_512x512_spokes_and_back(oracle, kb):
    """`tron sl.ra data.ra` (src/RUNME1_tron_degrid_phantom.sh:5) then `tron -a data.ra img.ra`
    (src/RUNME3_tron_grid_all.sh:6): default flags = linear angles, 512 readout x 512 spokes, 1 coil."""
    img = synth.image(1, 256, seed=synth.SEED_BASE + 24)
    want, p = oracle.recon(img, adjoint=0)
    assert (p.nxos, p.nro, p.npe1work) == (512, 512, 512)
    got, dims = lib.recon(img, adjoint=False, kb_mode=kb)
    assert (dims.nro, dims.npe1work) == (512, 512)
    assert rel_l2(got, want) <= TOL
    # adjoint of the oracle's data (so both sides grid identical inputs); Q5: the two linear conventions differ, parity is per direction
    back_want, q = oracle.recon(want, adjoint=1)
    assert (q.nx, q.npe1work) == (256, 512)
    back, _ = lib.recon(want, adjoint=True, kb_mode=kb)
    assert rel_l2(back, back_want) <= TOL


@pytest.mark.parametrize("kb", [lib.KB_FAST, lib.KB_EXACT])
def test_config2_at_its_literal_wording_256_readout_402_linear_spokes_1_coil(oracle, kb):
    """BASELINE config 2 as BASELINE.json words it: "256^2 grid, 256-pt readout x 402 linear-radial spokes, 1 coil" =
    `tron -a -u 2 in.ra` on [1, 1, 256, 402, 1]: nro 256 -> nx 128, nxos 256, all 402 spokes in one window (402 <= 256 * 2,
    src/tron.cu:916), linear angles (src/tron.cu:509).  (The RUNME1 / RUNME3 shape of the same config is the test above.)"""
    data = synth.kspace(1, 256, 402, seed=synth.SEED_BASE + 26)
    want, p = oracle.recon(data, adjoint=1, data_undersamp=2.0)
    assert (p.nxos, p.nx, p.npe1work, p.nz) == (256, 128, 402, 1)
    got, dims = lib.recon(data, adjoint=True, kb_mode=kb, data_undersamp=2.0)
    assert (dims.nxos, dims.nx, dims.npe1work, dims.nz) == (256, 128, 402, 1)
    assert got.shape == want.shape and np.isfinite(got).all()
    assert rel_l2(got, want) <= TOL


@pytest.mark.timeout(1800)
def test_device_resident_256_slices_as_the_bench_times_them(oracle):
    """What bench.py times, under pytest: `tron_nufft_adj_radial2d` on 256 slices x 8 coils resident in HBM = two 128-slice
    batches (gridding, then the fused FFT passes, on the plan's one stream), called twice in a row without a synchronisation
    between the calls; slices 0 / 127 / 128 / 255 of the second call against the oracle.  The stream is assembled from
    eight 32-slice blocks (one seeded generator call each); the oracle grids a block with the global angle index of its first
    spoke (-s, src/tron.cu:509)."""
    nc, nz, npe, blk = 8, 256, 402, 32
    flags = dict(golden=1, data_undersamp=0.7852, prof_slide=npe)
    cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=0.7852, prof_slide=npe)
    dims = lib.derive_dims(cfg, (nc, 1, NRO, npe * nz, 1))
    assert (dims.nz, dims.npe1work, dims.nxos, dims.nx) == (nz, npe, 512, 256)
    block_bytes = nc * NRO * npe * blk * 8
    kept = {}
    with lib.Plan(cfg, dims) as plan:
        assert "grid_arc_kernel" in plan.grid_kernel_name()
        d_in = lib.DeviceBuffer(block_bytes * (nz // blk))
        for k in range(nz // blk):
            data = synth.kspace(nc, NRO, npe * blk, seed=synth.SEED_BASE + 40 + k)
            d_in.write(np.asfortranarray(data).reshape(-1, order="F"), k * block_bytes)
            if k in (0, 3, 4, 7):
                kept[k] = data
        d_out = lib.DeviceBuffer(dims.out_bytes)
        plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1)
        plan.adjoint_device(d_out.ptr, d_in.ptr, 0, nz, 1)
        plan.sync()
        got = d_out.to_numpy(np.complex64, 256 * 256 * nz).reshape((256, 256, nz), order="F")
    assert np.isfinite(got).all()
    for z in (0, 127, 128, 255):
        k = z // blk
        want, _ = oracle.recon(kept[k], adjoint=1, zfirst=z - k * blk, zcount=1, skip_angles=k * blk * npe, **flags)
        err = rel_l2(got[..., z], want[0, 0, :, :, z - k * blk])
        assert err <= TOL, (z, err)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("golden", [1, 0])
def test_forward_at_the_bench_shape_64_images_8_coils_512x512_spokes(oracle, golden):
    """`bench.py --forward`'s launch under pytest: 8 coils x 64 images of 256^2 -> 512 readout x 512 spokes in ONE
    tron_nufft_radial2d call (degrid_stream_kernel at full sampling density: the kept-record and sorted-deal paths of the centre
    tiles are in play), images 0 / 31 / 63 against the oracle's degridradial2d pipeline (src/tron.cu:540-577, 639-649), golden and
    linear angles."""
    from test_gpu_degrid_stream import _forward
    nc, nimg = 8, 64
    imgs = [synth.image(nc, 256, seed=synth.SEED_BASE + 60 + k) for k in range(nimg)]
    got, name = _forward(imgs, nimg, golden_angle=golden)
    assert name == "degrid_stream_kernel"
    assert got.shape == (nimg, nc * 512 * 512) and np.isfinite(got.view(np.float32)).all()
    for k in (0, 31, 63):
        want, p = oracle.recon(imgs[k], adjoint=0, golden=golden)
        assert (p.nro, p.npe1work, p.nxos) == (512, 512, 512)
        assert rel_l2(got[k], want.reshape(-1, order="F")) <= TOL, k


@pytest.mark.timeout(1200)
def test_cgnr_at_the_metric_shape(oracle):
    """CGNR (src/tron.cu:665-720) at the headline size: 512 x 402 golden-angle spokes, 2 coils, 2 non-overlapping slices,
    2 iterations -- the fused 256 -> 512 forward FFT, tiled degridding with a per-slice angle stride, binned gridding and
    the batched reductions, against the oracle's restatement."""
    data = synth.kspace(2, NRO, 402 * 2, seed=synth.SEED_BASE + 25)
    flags = dict(data_undersamp=0.7852, prof_slide=402)
    want, p = oracle.recon_cgnr(data, 2, golden=1, **flags)
    got, dims = lib.recon(data, adjoint=True, golden_angle=1, niter=2, **flags)
    assert (dims.nz, dims.npe1work, dims.nxos) == (2, 402, 512) and p.nz == 2
    for z in range(2):
        assert rel_l2(got[..., z], want[..., z]) <= TOL, z


@pytest.mark.timeout(900)
def test_config3_whole_body_at_full_size_through_the_cli(oracle, tmp_path):
    """BASELINE config 3 at its real size: a synthetic [6, 1, 512, 20271, 1] complex64 stream (498 180 184 bytes with the
    header, what the reference's LFS pointer for ex_whole_body.ra declares) through `tron -v -u 0.4 -d 21 -a -G`
    (src/RUNME3_tron_grid_all.sh:10): 956 sliding windows of 204 spokes -> [1, 1, 256, 256, 956]; the first, the middle and
    the last slice against the oracle."""
    from tron_amd import ra
    nc, nro, npe1 = 6, 512, 20271
    data = synth.kspace(nc, nro, npe1, seed=synth.SEED_BASE + 3)
    inp, outp = str(tmp_path / "wb_in.ra"), str(tmp_path / "wb_out.ra")
    ra.write(inp, data)
    assert os.path.getsize(inp) == 498180184
    r = subprocess.run([os.path.join(ROOT, "tron_amd", "bin", "tron"), "-v", "-u", "0.4", "-d", "21", "-a", "-G", inp, outp],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = ra.read(outp)
    assert got.shape == (1, 1, 256, 256, 956)
    for z in (0, 478, 955):
        want, _ = oracle.recon(data, adjoint=1, zfirst=z, zcount=1, golden=1, data_undersamp=0.4, prof_slide=21)
        assert rel_l2(got[..., z], want[..., z]) <= 1e-5


@pytest.mark.timeout(600)
def test_whole_body_shape_with_complex_half_input_through_the_cli(oracle, tmp_path):
    """The whole-body shape (6 coils, 204-spoke windows sliding by 21, `tron -u 0.4 -d 21 -a -G`, src/RUNME3_tron_grid_all.sh:10)
    with the k-space file stored as complex-half (eltype 4, elbyte 4): 24-byte records on the arc kernel (round 5; the binned
    kernel until then).  86 windows of a [6, 1, 512, 2000, 1] stream; first, middle and last slice against the oracle run on
    the half-rounded input."""
    from tron_amd import ra
    nc, nro, npe1 = 6, 512, 2000
    data = synth.kspace(nc, nro, npe1, seed=synth.SEED_BASE + 31)
    h = np.stack([data.real, data.imag]).astype(np.float16)
    inp, outp = str(tmp_path / "wbh_in.ra"), str(tmp_path / "wbh_out.ra")
    ra.write(inp, h, complex_half=True)
    r = subprocess.run([os.path.join(ROOT, "tron_amd", "bin", "tron"), "-v", "-u", "0.4", "-d", "21", "-a", "-G", inp, outp],
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "grid_arc_kernel" in r.stdout, r.stdout[-1500:]
    got = ra.read(outp)
    assert got.shape == (1, 1, 256, 256, 86)
    rounded = np.asfortranarray((h[0].astype(np.float32) + 1j * h[1].astype(np.float32)).astype(np.complex64))
    for z in (0, 43, 85):
        want, _ = oracle.recon(rounded, adjoint=1, zfirst=z, zcount=1, golden=1, data_undersamp=0.4, prof_slide=21)
        assert rel_l2(got[..., z], want[..., z]) <= 1e-5
