"""Parity of the HIP path (through the C ABI) with the CPU oracle on identical seeded inputs.

Tolerances: north_star asks for fp32 relative L2 <= 1e-5 on the reconstruction.  The
gridding kernel keeps the reference's summation order, so with TRON_KB_EXACT the interpolation
stages are required to match the oracle to rounding of a single fused operation at most
(asserted at 1e-7, typically bit-exact); full pipelines add rocFFT-vs-double-DFT rounding.
"""
import ctypes

import numpy as np
import pytest

from conftest import rel_l2
import synth
from tron_amd import lib

pytestmark = pytest.mark.gpu

TOL_PIPELINE = 1e-5      # north_star tolerance, relative L2
TOL_STAGE_EXACT = 1e-7   # interpolation stages in exact mode
TOL_STAGE_FAST = 2e-6    # polynomial Kaiser-Bessel


def _plan(in_dims, adjoint, **flags):
    cfg = lib.default_config(adjoint=int(adjoint), **flags)
    dims = lib.derive_dims(cfg, in_dims)
    return lib.Plan(cfg, dims)


@pytest.mark.parametrize("golden,nchan,nxos,nro,npe,W,kb", [
    (1, 1, 64, 64, 48, 2.0, lib.KB_EXACT),
    (0, 1, 64, 64, 48, 2.0, lib.KB_EXACT),
    (1, 2, 64, 64, 33, 2.0, lib.KB_EXACT),
    (1, 6, 32, 32, 20, 2.0, lib.KB_EXACT),
    (1, 8, 48, 48, 20, 2.0, lib.KB_EXACT),
    (1, 1, 48, 64, 30, 2.0, lib.KB_EXACT),     # nro != nxos: readout resampling
    (1, 1, 40, 40, 25, 1.5, lib.KB_EXACT),
    (1, 2, 40, 40, 25, 2.5, lib.KB_EXACT),
    (1, 1, 72, 72, 300, 2.0, lib.KB_EXACT),    # more than one 256-spoke clip chunk
    (1, 1, 64, 64, 48, 2.0, lib.KB_FAST),
    (0, 2, 64, 64, 48, 2.0, lib.KB_FAST),
])
def test_grid_stage_vs_oracle(oracle, golden, nchan, nxos, nro, npe, W, kb):
    """tron_gridradial2d == gridradial2d kernel (src/tron.cu:465-536), reference layouts."""
    raw = synth.uniform_c64(npe * nro * nchan, 101).reshape(npe, nro, nchan)
    nu = oracle.precompensate(raw)
    skip = 7
    want = oracle.gridradial2d(nu, nxos, W=W, gridos=nxos / (nro / 2), skip_angles=skip, golden=golden)
    # a plan whose derived dims are (nxos, nro, npe): adjoint, -o nxos/(nro/2), -u big
    with _plan((nchan, 1, nro, npe, 1), 1, golden_angle=golden, gridos=nxos / (nro / 2), kernwidth=W,
               data_undersamp=1e6, kb_mode=kb) as plan:
        assert plan.dims.nxos == nxos and plan.dims.npe1work == npe
        d_in = lib.DeviceBuffer.from_numpy(raw)
        plan.precompensate_device(d_in.ptr)            # = the precompensate kernel, src/tron.cu:405-416, in place
        plan.sync()
        assert np.array_equal(d_in.to_numpy(np.complex64, npe * nro * nchan).view(np.uint32), np.ascontiguousarray(nu).reshape(-1).view(np.uint32))
        d_out = lib.DeviceBuffer(nxos * nxos * nchan * 8)
        plan.grid_device(d_out.ptr, d_in.ptr, skip)
        plan.sync()
        got = d_out.to_numpy(np.complex64, nxos * nxos * nchan).reshape(nxos, nxos, nchan)
    err = rel_l2(got, want)
    if kb == lib.KB_EXACT:
        assert err <= TOL_STAGE_EXACT, err
        # same terms, same order, unfused arithmetic: expect identical bits
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"not bit-exact (rel {err:.3e})"
    else:
        assert err <= TOL_STAGE_FAST, err


@pytest.mark.parametrize("golden,nrep,n,nro,npe,kb", [
    (1, 1, 64, 64, 40, lib.KB_EXACT),
    (0, 1, 64, 64, 40, lib.KB_EXACT),
    (1, 2, 32, 32, 17, lib.KB_EXACT),
    (1, 6, 32, 32, 17, lib.KB_EXACT),
    (1, 1, 64, 64, 40, lib.KB_FAST),
])
def test_degrid_stage_vs_oracle(oracle, golden, nrep, n, nro, npe, kb):
    """tron_degridradial2d == degridradial2d kernel (src/tron.cu:540-577)."""
    u = synth.uniform_c64(n * n * nrep, 202).reshape(n, n, nrep)
    nx = n // 2
    want = oracle.degridradial2d(u, nro, npe, skip_angles=3, golden=golden)
    with _plan((nrep, 1, nx, nx, 1), 0, golden_angle=golden, data_undersamp=npe / nro + 1e-6, skip_angles=3, kb_mode=kb) as plan:
        assert (plan.dims.nxos, plan.dims.nro, plan.dims.npe1work) == (n, nro, npe)
        d_in = lib.DeviceBuffer.from_numpy(u)
        d_out = lib.DeviceBuffer(npe * nro * nrep * 8)
        plan.degrid_device(d_out.ptr, d_in.ptr)
        plan.sync()
        got = d_out.to_numpy(np.complex64, npe * nro * nrep).reshape(npe, nro, nrep)
    err = rel_l2(got, want)
    if kb == lib.KB_EXACT:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"not bit-exact (rel {err:.3e})"
    else:
        assert err <= TOL_STAGE_FAST, err


ADJ_CASES = [
    # nc, nro, npe1, flags
    (1, 64, 50, dict(golden_angle=1, data_undersamp=2.0)),
    (1, 64, 50, dict(golden_angle=0, data_undersamp=2.0)),
    (2, 64, 120, dict(golden_angle=1, data_undersamp=0.5, prof_slide=11)),          # sliding window, 9 slices
    (6, 32, 90, dict(golden_angle=1, data_undersamp=0.5, prof_slide=7, skip_angles=5)),
    (8, 32, 40, dict(golden_angle=1, data_undersamp=1.0)),                           # > MAXCHAN of the reference
    (1, 64, 40, dict(golden_angle=1, data_undersamp=2.0, gridos=1.5)),
    (1, 48, 30, dict(golden_angle=1, data_undersamp=2.0, kernwidth=2.5)),
    (2, 24, 81, dict(golden_angle=1, data_undersamp=1.0, prof_slide=8, gridos=1.5, kernwidth=1.5)),   # nxos 18: nxos/2 odd
    (4, 16, 50, dict(golden_angle=0, data_undersamp=4.0, gridos=1.25, kernwidth=3.0)),                # nxos 10
    (2, 64, 64, dict(golden_angle=0, data_undersamp=0.25, prof_slide=16)),           # linear angle windows
]


@pytest.mark.parametrize("kb", [lib.KB_EXACT, lib.KB_FAST])
@pytest.mark.parametrize("nc,nro,npe1,flags", ADJ_CASES)
def test_adjoint_recon_vs_oracle(oracle, nc, nro, npe1, flags, kb):
    """tron -a ... == recon_radial2d adjoint branch (src/tron.cu:726-786)."""
    data = synth.kspace(nc, nro, npe1, seed=303)
    want, p = oracle.recon(data, adjoint=1, **{("golden" if k == "golden_angle" else k): v for k, v in flags.items()})
    got, dims = lib.recon(data, adjoint=True, kb_mode=kb, **flags)
    assert got.shape == want.shape == tuple(p.out_dims)
    assert (dims.nz, dims.npe1work) == (p.nz, p.npe1work)
    err = rel_l2(got, want)
    assert err <= TOL_PIPELINE, err


@pytest.mark.parametrize("kb", [lib.KB_EXACT, lib.KB_FAST])
@pytest.mark.parametrize("nc,nx,flags", [
    (1, 32, dict()),                                     # linear angle, like RUNME1
    (1, 32, dict(golden_angle=1)),
    (2, 32, dict(golden_angle=1, data_undersamp=0.5)),
    (6, 16, dict(golden_angle=1, skip_angles=4)),
])
def test_forward_recon_vs_oracle(oracle, nc, nx, flags, kb):
    """tron (no -a) == recon_radial2d forward branch."""
    img = synth.image(nc, nx, seed=404)
    want, p = oracle.recon(img, adjoint=0, **{("golden" if k == "golden_angle" else k): v for k, v in flags.items()})
    got, dims = lib.recon(img, adjoint=False, kb_mode=kb, **flags)
    assert got.shape == want.shape
    err = rel_l2(got, want)
    assert err <= TOL_PIPELINE, err


def test_sharded_ranges_assemble(oracle):
    """Disjoint slice ranges computed by separate plans fill the same output."""
    data = synth.kspace(2, 32, 100, seed=505)
    flags = dict(golden_angle=1, data_undersamp=0.5, prof_slide=6)
    full, dims = lib.recon(data, adjoint=True, **flags)
    cfg = lib.default_config(adjoint=1, **flags)
    d = lib.derive_dims(cfg, data.shape)
    flat = np.asfortranarray(data).reshape(-1, order="F")
    out = np.zeros(d.out_bytes // 8, np.complex64)
    cuts = [0, d.nz // 3, d.nz // 3 + 1, d.nz]
    for a, b in zip(cuts[:-1], cuts[1:]):
        with lib.Plan(cfg, d) as plan:
            plan.recon(flat, zfirst=a, zcount=b - a, out=out)
    assert np.array_equal(out.reshape(full.shape, order="F"), full)   # deterministic kernels: identical bits


def test_device_resident_adjoint_uncombined(oracle):
    """combine=0 returns the deapodised coil images of tron_nufft_adj_radial2d (src/tron.cu:623-637)."""
    nc, nro, npe = 2, 32, 24
    data = synth.kspace(nc, nro, npe, seed=606)
    p = oracle.make_params(data.shape, adjoint=1, golden=1, data_undersamp=2.0)
    d_u = np.ascontiguousarray(np.asfortranarray(data).reshape(-1, order="F"))
    nbuf = nc * max(nro * npe, p.nxos * p.nxos)
    u = np.zeros(nbuf, np.complex64); u[:d_u.size] = d_u
    v = np.zeros(nbuf, np.complex64)
    oracle.lib().oracle_nufft_adj_radial2d(ctypes.byref(p), v.ctypes.data_as(ctypes.c_void_p), u.ctypes.data_as(ctypes.c_void_p), 0)
    want = v[:p.nx * p.nx * nc]
    with _plan(data.shape, 1, golden_angle=1, data_undersamp=2.0) as plan:
        d_in = lib.DeviceBuffer.from_numpy(d_u)
        d_out = lib.DeviceBuffer(p.nx * p.nx * nc * 8)
        plan.adjoint_device(d_out.ptr, d_in.ptr, 0, 1, combine=0)
        plan.sync()
        got = d_out.to_numpy(np.complex64, p.nx * p.nx * nc)
    assert rel_l2(got, want) <= TOL_PIPELINE


def test_errors_are_reported_not_fatal():
    cfg = lib.default_config(adjoint=1, kernwidth=5.0)            # kernel widths beyond 4 are not implemented
    d = lib.derive_dims(cfg, (1, 1, 32, 10, 1))
    with pytest.raises(lib.TronError) as e:
        lib.Plan(cfg, d)
    assert e.value.code == lib.TRON_ERR_UNSUPPORTED
    cfg = lib.default_config(adjoint=1, niter=2, input_half=1)    # CGNR keeps its residual in fp32
    with pytest.raises(lib.TronError) as e:
        lib.Plan(cfg, d)
    assert e.value.code == lib.TRON_ERR_UNSUPPORTED
    with pytest.raises(lib.TronError):
        lib.derive_dims(lib.default_config(adjoint=1), (3, 1, 32, 10, 1))      # odd coil count (tron.cu:963)
    cfg = lib.default_config(adjoint=1, device=99)
    with pytest.raises(lib.TronError):
        lib.Plan(cfg, lib.derive_dims(cfg, (1, 1, 32, 10, 1)))


@pytest.mark.parametrize("nc,kb", [(1, lib.KB_EXACT), (2, lib.KB_FAST), (8, lib.KB_FAST)])
def test_metric_size_pipeline_vs_oracle(oracle, monkeypatch, nc, kb):
    """512-point readout -> 512^2 grid -> 256^2 image: the shape the fused pruned FFT (tron_fft512.hip) and the
    32x32 binned gridding kernel are specialised for.  Checked against the oracle and against the
    rocFFT + post_kernel path."""
    npe = 48
    data = synth.kspace(nc, 512, 2 * npe, seed=707)
    flags = dict(golden_angle=1, data_undersamp=npe / 512 + 1e-6, prof_slide=npe)
    want, p = oracle.recon(data, adjoint=1, golden=1, data_undersamp=npe / 512 + 1e-6, prof_slide=npe)
    assert (p.nz, p.nxos, p.nx, p.npe1work) == (2, 512, 256, npe)
    # the fused path neither stores nor loads grid points beyond the sampled disc: NaN-fill the work grid to prove it
    monkeypatch.setenv("TRON_DEBUG", "poison")
    got, _ = lib.recon(data, adjoint=True, kb_mode=kb, **flags)
    assert np.isfinite(got).all()
    assert rel_l2(got, want) <= TOL_PIPELINE
    monkeypatch.setenv("TRON_FFT", "rocfft")
    ref, _ = lib.recon(data, adjoint=True, kb_mode=kb, **flags)
    assert rel_l2(ref, want) <= TOL_PIPELINE
    assert rel_l2(got, ref) <= 2e-6


@pytest.mark.parametrize("nc,kb", [(1, lib.KB_EXACT), (2, lib.KB_FAST), (8, lib.KB_FAST)])
def test_metric_size_forward_vs_oracle(oracle, monkeypatch, nc, kb):
    """256^2 image -> 512^2 grid -> 512-point spokes: the shape the fused pruned forward FFT (pad + deapodise +
    shift + FFT, tron_fft512.hip) is specialised for.  Checked against the oracle and against the
    pre_kernel + rocFFT path."""
    img = synth.image(nc, 256, seed=717)
    flags = dict(golden_angle=1, data_undersamp=40 / 512 + 1e-6)
    want, p = oracle.recon(img, adjoint=0, golden=1, data_undersamp=40 / 512 + 1e-6)
    assert (p.nxos, p.nx, p.nro, p.npe1work) == (512, 256, 512, 40)
    got, _ = lib.recon(img, adjoint=False, kb_mode=kb, **flags)
    assert rel_l2(got, want) <= TOL_PIPELINE
    monkeypatch.setenv("TRON_FFT", "rocfft")
    ref, _ = lib.recon(img, adjoint=False, kb_mode=kb, **flags)
    assert rel_l2(ref, want) <= TOL_PIPELINE
    assert rel_l2(got, ref) <= 2e-6


@pytest.mark.parametrize("nc", [2, 8])
def test_half_precision_kspace_input(oracle, nc):
    """Config 5: k-space stored as complex-half (.ra eltype 4 / elbyte 4), converted with the reference's
    round-to-nearest-even (src/float16.cu); gridded from half storage with fp32 accumulation.  Parity is
    against the fp32 oracle run on the SAME half-rounded values."""
    nro, npe = 64, 90                       # 8 coils: the 16-byte (four-coil) load path of the binned kernel
    data = synth.kspace(nc, nro, npe, seed=808)
    halves = np.asfortranarray(data).reshape(-1, order="F").view(np.float32).astype(np.float16)
    L = lib.load()
    mine = np.array([L.ra_float_to_half_bits(int(b)) for b in np.asfortranarray(data).reshape(-1, order="F").view(np.uint32)[:512]], np.uint16)
    assert np.array_equal(mine, halves.view(np.uint16)[:512])                 # same rounding as the C converter
    rounded = halves.astype(np.float32).view(np.complex64).reshape(data.shape, order="F")
    flags = dict(golden_angle=1, data_undersamp=0.5, prof_slide=13)
    want, _ = oracle.recon(rounded, adjoint=1, golden=1, data_undersamp=0.5, prof_slide=13)
    h = halves.reshape((2,) + data.shape, order="F")
    for kb in (lib.KB_EXACT, lib.KB_FAST):
        got, _ = lib.recon(h, adjoint=True, input_half=1, kb_mode=kb, **flags)
        assert rel_l2(got, want) <= TOL_PIPELINE


def test_whole_body_shaped_stream(oracle):
    """Config 3 in miniature: `tron -u 0.4 -d 21 -a -G` (src/RUNME3_tron_grid_all.sh:10) on a 6-coil,
    512-readout golden-angle stream: sliding windows of 204 spokes, hop 21.  Every slice is computed on the
    GPU; the first, a middle and the last slice are checked against the oracle."""
    nc, nro, npe1 = 6, 512, 204 + 21 * 7
    data = synth.kspace(nc, nro, npe1, seed=909)
    flags = dict(golden_angle=1, data_undersamp=0.4, prof_slide=21)
    got, dims = lib.recon(data, adjoint=True, kb_mode=lib.KB_FAST, **flags)
    assert (dims.nz, dims.npe1work, dims.nx) == (8, 204, 256)
    for z in (0, 4, 7):
        want, _ = oracle.recon(data, adjoint=1, zfirst=z, zcount=1, golden=1, data_undersamp=0.4, prof_slide=21)
        assert rel_l2(got[..., z], want[..., z]) <= TOL_PIPELINE


@pytest.mark.parametrize("kb", [lib.KB_EXACT, lib.KB_FAST])
def test_odd_and_tiny_grids(oracle, kb):
    """Odd oversampled sizes (forward shift n/2, inverse shift n - n/2 differ, src/tron.cu:164) go through
    rocFFT + post_kernel / pre_kernel; grids smaller than one tile exercise the partial-tile paths."""
    # adjoint: nro 40, -o 1.25 -> nx 20, nxos 25 (odd)
    data = synth.kspace(2, 40, 30, seed=1001)
    flags = dict(golden_angle=1, data_undersamp=2.0, gridos=1.25)
    want, p = oracle.recon(data, adjoint=1, golden=1, data_undersamp=2.0, gridos=1.25)
    assert p.nxos == 25
    got, _ = lib.recon(data, adjoint=True, kb_mode=kb, **flags)
    assert rel_l2(got, want) <= TOL_PIPELINE
    # forward: nx 10, -o 1.5 -> nxos 15 (odd), nro 15
    img = synth.image(1, 10, seed=1002)
    want, p = oracle.recon(img, adjoint=0, golden=1, gridos=1.5)
    assert p.nxos == 15
    got, _ = lib.recon(img, adjoint=False, kb_mode=kb, golden_angle=1, gridos=1.5)
    assert rel_l2(got, want) <= TOL_PIPELINE
    # tiny: 8x8 image -> 16x16 grid (smaller than a 32x32 tile), one spoke per image
    img = synth.image(2, 8, seed=1003)
    want, _ = oracle.recon(img, adjoint=0, golden=1, data_undersamp=0.0626)
    got, d = lib.recon(img, adjoint=False, kb_mode=kb, golden_angle=1, data_undersamp=0.0626)
    assert d.npe1work == 1 and rel_l2(got, want) <= TOL_PIPELINE
    data = synth.kspace(1, 16, 1, seed=1004)
    want, _ = oracle.recon(data, adjoint=1, golden=1, data_undersamp=2.0)
    got, d = lib.recon(data, adjoint=True, kb_mode=kb, golden_angle=1, data_undersamp=2.0)
    assert d.npe1work == 1 and rel_l2(got, want) <= TOL_PIPELINE


def test_sharded_cli_single_rank(tmp_path):
    """python -m tron_amd.shard with one rank == the tron binary (the N>1 gather is covered on CPU with gloo)."""
    import os, subprocess, sys
    from tron_amd import ra
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = synth.kspace(2, 32, 60, seed=1005)
    inp, o1, o2 = (str(tmp_path / n) for n in ("in.ra", "a.ra", "b.ra"))
    ra.write(inp, data)
    argv = ["-a", "-G", "-u", "0.5", "-d", "9"]
    assert subprocess.run([os.path.join(root, "tron_amd", "bin", "tron")] + argv + [inp, o1]).returncode == 0
    r = subprocess.run([sys.executable, "-m", "tron_amd.shard"] + argv + [inp, o2], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(ra.read(o1), ra.read(o2))


@pytest.mark.parametrize("kb", [lib.KB_EXACT, lib.KB_FAST])
def test_many_spokes_clip_rounds(oracle, kb):
    """More spokes than one clip round holds (512 in the binned gridding kernel, 256 in the gather and in the tiled
    degridding kernel): BASELINE config 4 has 804 spokes per slice."""
    data = synth.kspace(2, 64, 804, seed=1101)
    want, p = oracle.recon(data, adjoint=1, golden=1, data_undersamp=13.0)
    assert p.npe1work == 804 and p.nz == 1
    got, _ = lib.recon(data, adjoint=True, kb_mode=kb, golden_angle=1, data_undersamp=13.0)
    assert rel_l2(got, want) <= TOL_PIPELINE
    img = synth.image(2, 32, seed=1102)
    want, p = oracle.recon(img, adjoint=0, golden=1, data_undersamp=9.4)
    assert p.npe1work == 601
    got, _ = lib.recon(img, adjoint=False, kb_mode=kb, golden_angle=1, data_undersamp=9.4)
    assert rel_l2(got, want) <= TOL_PIPELINE


def test_cli_complex_half_input_and_flag_variants(oracle, tmp_path):
    """The `tron` binary end to end: a complex-half .ra input (eltype 4 / elbyte 4, BASELINE config 5) is gridded from
    half storage; non-default -k / -o / -s / -d travel through getopt to the kernels."""
    import os, subprocess
    from tron_amd import ra
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tron = os.path.join(root, "tron_amd", "bin", "tron")
    data = synth.kspace(4, 48, 70, seed=1201)
    h16 = np.asfortranarray(data).reshape(-1, order="F").view(np.float32).astype(np.float16)
    rounded = h16.astype(np.float32).view(np.complex64).reshape(data.shape, order="F")
    inp, out = str(tmp_path / "half.ra"), str(tmp_path / "img.ra")
    ra.write(inp, h16.reshape((2,) + data.shape, order="F"), complex_half=True)
    assert subprocess.run([tron, "-a", "-G", "-u", "0.5", "-d", "9", inp, out]).returncode == 0
    want, _ = oracle.recon(rounded, adjoint=1, golden=1, data_undersamp=0.5, prof_slide=9)
    got = ra.read(out)
    assert got.shape == want.shape and rel_l2(got, want) <= TOL_PIPELINE
    # fp32 input, wider kernel, lower oversampling, angle offset
    inp2 = str(tmp_path / "f32.ra")
    ra.write(inp2, data)
    assert subprocess.run([tron, "-a", "-G", "-k", "2.5", "-o", "1.5", "-s", "17", "-u", "0.6", "-d", "20", inp2, out]).returncode == 0
    want, _ = oracle.recon(data, adjoint=1, golden=1, kernwidth=2.5, gridos=1.5, skip_angles=17, data_undersamp=0.6, prof_slide=20)
    got = ra.read(out)
    assert got.shape == want.shape and rel_l2(got, want) <= TOL_PIPELINE
    # forward, linear angles, two coils
    img = synth.image(2, 24, seed=1202)
    ra.write(inp2, img)
    assert subprocess.run([tron, "-k", "1.5", inp2, out]).returncode == 0
    want, _ = oracle.recon(img, adjoint=0, golden=0, kernwidth=1.5)
    # the reference's forward header keeps dims[0] = 1 while the payload holds all coils: compare the payload
    h = ra.read_header(out)
    assert h.dims[0] == 1 and h.size == want.size * 8
    with open(out, "rb") as f:
        f.seek(h.nbytes_header)
        got = np.frombuffer(f.read(h.size), np.complex64)
    assert rel_l2(got, want.reshape(-1, order="F")) <= TOL_PIPELINE


def test_two_plans_alive_and_threads(oracle):
    """Plans are independent objects (no globals, unlike src/tron.cu:54-87): an adjoint and a forward plan alive at once,
    used alternately, and two plans driven from two host threads, give the results of isolated runs."""
    import threading
    data = synth.kspace(2, 64, 60, seed=1301)
    img = synth.image(2, 32, seed=1302)
    fa = dict(golden_angle=1, data_undersamp=0.5, prof_slide=10)
    ff = dict(golden_angle=1)
    ref_a, _ = lib.recon(data, adjoint=True, **fa)
    ref_f, _ = lib.recon(img, adjoint=False, **ff)
    ca, cf = lib.default_config(adjoint=1, **fa), lib.default_config(adjoint=0, **ff)
    da, df = lib.derive_dims(ca, data.shape), lib.derive_dims(cf, img.shape)
    flat_a = np.asfortranarray(data).reshape(-1, order="F")
    flat_f = np.asfortranarray(img).reshape(-1, order="F")
    with lib.Plan(ca, da) as pa, lib.Plan(cf, df) as pf:
        for _ in range(3):
            oa = np.zeros(da.out_bytes // 8, np.complex64)
            of = np.zeros(df.out_bytes // 8, np.complex64)
            pa.recon(flat_a, out=oa)
            pf.recon(flat_f, out=of)
            assert np.array_equal(oa, ref_a.reshape(-1, order="F")) and np.array_equal(of, ref_f.reshape(-1, order="F"))
        res = {}

        def work(name, plan, flat, nbytes):
            o = np.zeros(nbytes // 8, np.complex64)
            for _ in range(4):
                plan.recon(flat, out=o)
            res[name] = o
        ta = threading.Thread(target=work, args=("a", pa, flat_a, da.out_bytes))
        tf = threading.Thread(target=work, args=("f", pf, flat_f, df.out_bytes))
        ta.start(); tf.start(); ta.join(); tf.join()
        assert np.array_equal(res["a"], ref_a.reshape(-1, order="F")) and np.array_equal(res["f"], ref_f.reshape(-1, order="F"))


def test_c_client_matches_the_tron_binary(tmp_path):
    """examples/recon_c_abi.c (a C99 program against the C ABI) writes the same bytes as the tron binary."""
    import os, subprocess
    from test_host import _build_c_example
    from tron_amd import ra
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = _build_c_example(tmp_path)
    data = synth.kspace(2, 32, 40, seed=1401)
    inp, o1, o2 = (str(tmp_path / n) for n in ("in.ra", "c.ra", "t.ra"))
    ra.write(inp, data)
    assert subprocess.run([exe, "-a", inp, o1]).returncode == 0
    tron = os.path.join(root, "tron_amd", "bin", "tron")
    assert subprocess.run([tron, "-a", "-G", inp, o2]).returncode == 0
    assert open(o1, "rb").read() == open(o2, "rb").read()
    # a later batch of a continuing acquisition on ONE plan (tron_plan_retarget in the client) = the binary's -s (src/tron.cu:629-630)
    assert subprocess.run([exe, "-a", "-s", "4020", inp, o1]).returncode == 0
    assert subprocess.run([tron, "-a", "-G", "-s", "4020", inp, o2]).returncode == 0
    assert open(o1, "rb").read() == open(o2, "rb").read()
    assert subprocess.run([tron, "-a", "-G", inp, o2]).returncode == 0
    # the multi-GPU entry point (every visible device) and CGNR through the same client
    assert subprocess.run([exe, "-a", "-m", inp, o1]).returncode == 0
    assert open(o1, "rb").read() == open(o2, "rb").read()
    assert subprocess.run([exe, "-a", "-i", "2", inp, o1]).returncode == 0
    assert subprocess.run([tron, "-a", "-G", "-i", "2", inp, o2]).returncode == 0
    assert open(o1, "rb").read() == open(o2, "rb").read()


def _adjoint_device_resident(data, flags, twice=False):
    cfg = lib.default_config(adjoint=1, **flags)
    dims = lib.derive_dims(cfg, data.shape)
    with lib.Plan(cfg, dims) as plan:
        d_in = lib.DeviceBuffer.from_numpy(np.asfortranarray(data).reshape(-1, order="F"))
        d_out = lib.DeviceBuffer(dims.out_bytes)
        for _ in range(2 if twice else 1):          # back to back, no synchronisation between the calls
            plan.adjoint_device(d_out.ptr, d_in.ptr, 0, dims.nz, combine=1)
        plan.sync()
        return d_out.to_numpy(np.complex64, dims.out_bytes // 8), dims


def test_device_resident_batches_do_not_change_bytes():
    """Device-resident runs cut a slice range into equal batches of about chunk_slices (one work grid, one stream since round 6:
    the second lane of rounds 2-5 returned nothing measurable and is gone).  The batch size changes no byte, two calls queued back
    to back reuse the work grid safely, and the host-buffer entry point on the same plan settings agrees."""
    data = synth.kspace(2, 512, 9 * 20, seed=1501)
    flags = dict(golden_angle=1, data_undersamp=20 / 512 + 1e-6, prof_slide=20)
    ref, d = _adjoint_device_resident(data, dict(flags, chunk_slices=9))
    assert d.nz == 9 and d.nxos == 512
    for chunk in (1, 2, 4):
        got, _ = _adjoint_device_resident(data, dict(flags, chunk_slices=chunk), twice=chunk == 2)
        assert np.array_equal(got, ref), chunk
    host, _ = lib.recon(data, adjoint=True, chunk_slices=2, **flags)
    assert np.array_equal(host.reshape(-1, order="F"), ref)


def test_forward_batches_split_into_chunks(oracle, monkeypatch):
    """tron_nufft_radial2d on more images than one internal batch holds (fused 256 -> 512 path and the rocFFT path):
    every image equals the oracle's forward of that image."""
    nc, nimg = 2, 7
    imgs = [synth.image(nc, 256, seed=1600 + k) for k in range(nimg)]
    flags = dict(golden_angle=1, data_undersamp=24 / 512 + 1e-6)
    wants = [oracle.recon(im, adjoint=0, golden=1, data_undersamp=24 / 512 + 1e-6)[0].reshape(-1, order="F") for im in imgs[:2] + imgs[-1:]]
    flat = np.concatenate([np.asfortranarray(im).reshape(-1, order="F") for im in imgs])
    for fft in ("fused", "rocfft"):
        if fft == "rocfft":
            monkeypatch.setenv("TRON_FFT", "rocfft")
        cfg = lib.default_config(adjoint=0, chunk_slices=3, **flags)
        dims = lib.derive_dims(cfg, imgs[0].shape)
        per = nc * dims.nro * dims.npe1work
        with lib.Plan(cfg, dims) as plan:
            d_in = lib.DeviceBuffer.from_numpy(flat)
            d_out = lib.DeviceBuffer(nimg * per * 8)
            plan.forward_device(d_out.ptr, d_in.ptr, nimg)
            plan.sync()
            got = d_out.to_numpy(np.complex64, nimg * per)
        for k, want in zip((0, 1, nimg - 1), wants):
            assert rel_l2(got[k * per:(k + 1) * per], want) <= TOL_PIPELINE, (fft, k)
