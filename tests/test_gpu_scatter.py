"""The sample-driven gridding kernel for one and two channels (tron_grid_scatter.hip, round 5): lane = sample, 64-bit fixed-point
sums in LDS.  Checked against the oracle (the reference's gridradial2d + pipeline, src/tron.cu:465-536, 623-655) at the
north_star's 1e-5 relative L2, against the arc kernel it replaces for these channel counts (TRON_GRID_KERNEL=arc) at 2e-6,
bit for bit run to run, and on data whose magnitude spans four decades (the fixed-point scale is per tile and slice)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import rel_l2
import synth
from tron_amd import lib

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# Plans take this kernel for one channel (fp32 or complex-half k-space) and for two channels of complex-half; two fp32 channels stay with
# the arc kernel (faster there) unless TRON_GRID_KERNEL=scatter asks for it: set for this whole module (the switch is read at plan
# creation), so that the two-channel instantiation is tested all the same.
@pytest.fixture(autouse=True, scope="module")
def _scatter_for_two_channels_too():
    # (TRON_SLICES_PER_PASS=0: linear-angle plans with few channels otherwise grid several slices per pass of the BINNED kernel)
    old = {k: os.environ.get(k) for k in ("TRON_GRID_KERNEL", "TRON_SLICES_PER_PASS")}
    os.environ["TRON_GRID_KERNEL"] = "scatter"
    os.environ["TRON_SLICES_PER_PASS"] = "0"
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _kernel_name(shape, **flags):
    cfg = lib.default_config(adjoint=1, **flags)
    dims = lib.derive_dims(cfg, shape)
    with lib.Plan(cfg, dims) as plan:
        return plan.grid_kernel_name()


def _child(data, env_extra, **flags):
    """The same reconstruction in a child process with tuning switches set (they are read once per process)."""
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); from tron_amd import lib\n"
        "d = np.load(sys.argv[1]); out, _ = lib.recon(d, adjoint=True, **eval(sys.argv[3])); np.save(sys.argv[2], out)\n" % ROOT)
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "in.npy"), data)
        env = dict(os.environ, TRON_TUNING="1", **env_extra)
        r = subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "in.npy"), os.path.join(tmp, "out.npy"), repr(flags)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return np.load(os.path.join(tmp, "out.npy"))


CASES = [
    # nc, nro, spokes per slice, slices, flags
    (1, 256, 180, 3, dict(golden_angle=1)),
    (2, 256, 150, 4, dict(golden_angle=1, prof_slide=37)),          # sliding windows
    (1, 128, 64, 5, dict(golden_angle=1, skip_angles=7)),            # smallest grid (4 x 4 tiles)
    (2, 256, 100, 2, dict(golden_angle=0)),                         # linear angles: samples at exactly |x| = W on the axis spokes
    (1, 256, 120, 2, dict(golden_angle=0)),
    (2, 512, 402, 2, dict(golden_angle=1)),                         # the metric's shape
    (1, 1024, 60, 1, dict(golden_angle=1)),                         # 1024^2 grid, few spokes
    (2, 256, 100, 2, dict(golden_angle=1, gridos=1.5)),             # nro != nxos: radius r reads sample (r nro) / nxos (src/tron.cu:517)
    (1, 256, 100, 2, dict(golden_angle=1, gridos=3.0)),
    (2, 256, 1300, 2, dict(golden_angle=1)),                        # more than 1 024 spokes per window: two passes, the second adds
    (1, 256, 12, 64, dict(golden_angle=1)),                         # few spokes, 4 slices per workgroup: empty runs between them
    (1, 256, 1300, 2, dict(golden_angle=1)),                        # one channel, two passes of 650 spokes (64-tiles, the second pass adds)
    (1, 256, 150, 7, dict(golden_angle=1, prof_slide=37)),          # one channel, sliding windows, a slice count that is no multiple of the slices per workgroup
    (1, 512, 804, 2, dict(golden_angle=1)),                         # config 4's slice shape with one coil: the largest runs a 64-tile sees here
]


@pytest.mark.parametrize("nc,nro,npe,nz,flags", CASES)
def test_scatter_kernel_vs_oracle_and_arc(oracle, nc, nro, npe, nz, flags):
    slide = flags.get("prof_slide", npe)
    data = synth.kspace(nc, nro, npe + slide * (nz - 1), seed=9500 + nc + nro + npe)
    fl = dict(flags)
    fl.setdefault("prof_slide", npe)
    fl["data_undersamp"] = (npe + 0.5) / nro          # npe1work = int(data_undersamp * nro), src/tron.cu:916
    assert "grid_scatter_kernel" in _kernel_name(data.shape, **fl)
    got, dims = lib.recon(data, adjoint=True, **fl)
    assert dims.nz == nz and dims.npe1work == npe
    again, _ = lib.recon(data, adjoint=True, **fl)
    assert np.array_equal(got, again)                 # integer sums: the same bits whatever order the additions ran in
    oflags = {("golden" if k == "golden_angle" else k): v for k, v in fl.items()}
    for z in sorted({0, nz // 2, nz - 1}):
        want, _ = oracle.recon(data, adjoint=1, zfirst=z, zcount=1, **oflags)
        assert rel_l2(got[..., z], want[..., z]) <= 1e-5, z
    other = _child(data, dict(TRON_GRID_KERNEL="arc"), **fl)
    assert rel_l2(got, other) <= 2e-6


@pytest.mark.parametrize("nc", [1, 2])
def test_scatter_kernel_complex_half_input(oracle, nc):
    """complex-half k-space with one and two channels: read by plain 4- / 8-byte loads (the arc kernel copies 16 bytes = four
    channels at a time and left these counts to the binned kernel)."""
    npe = 140
    data = synth.kspace(nc, 256, npe * 2, seed=9600 + nc)
    h = np.stack([data.real, data.imag]).astype(np.float16)
    fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / 256, prof_slide=npe)
    assert "grid_scatter_kernel" in _kernel_name(data.shape, input_half=1, **fl)
    a, dims = lib.recon(h, adjoint=True, input_half=1, **fl)
    b, _ = lib.recon(h, adjoint=True, input_half=1, **fl)
    assert np.array_equal(a, b)
    rounded = (h[0].astype(np.float32) + 1j * h[1].astype(np.float32)).astype(np.complex64)
    want, _ = oracle.recon(rounded, adjoint=1, golden=1, data_undersamp=(npe + 0.5) / 256, prof_slide=npe)
    assert rel_l2(a, want) <= 1e-5


@pytest.mark.parametrize("nc", [1, 2])
def test_scatter_kernel_dynamic_range(oracle, nc):
    """k-space as it comes off a scanner: magnitude falling by four decades from the centre to the rim, a noise floor, and one
    spike 300 times its neighbourhood.  The fixed-point scale is chosen per (tile, slice) from the tile's largest sample, so the
    outer tiles keep their own precision; the tile with the spike loses bits below 2^-24 of the SPIKE, which the 1e-5 bound on the
    whole image tolerates (it is what fp32 itself does to the points the spike reaches)."""
    nro, npe = 256, 160
    rng = np.random.default_rng(9700 + nc)
    data = synth.kspace(nc, nro, npe * 2, seed=9700 + nc)
    r = np.abs(np.arange(nro) - nro // 2).astype(np.float32)
    env = (1.0 / (1.0 + (r / 2.0) ** 2) + 1e-4).astype(np.float32)            # 1 .. 1e-4
    data = (data * env[None, None, :, None, None]).astype(np.complex64)
    data[0, 0, nro // 2 + 77, 5, 0] *= 300.0
    data = np.asfortranarray(data)
    fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    assert "grid_scatter_kernel" in _kernel_name(data.shape, **fl)
    got, _ = lib.recon(data, adjoint=True, **fl)
    want, _ = oracle.recon(data, adjoint=1, golden=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    assert rel_l2(got, want) <= 1e-5
    other = _child(data, dict(TRON_GRID_KERNEL="arc"), **fl)
    assert rel_l2(got, other) <= 2e-6
    # the same data scaled by 2^12 and by 2^-12: the scale is a power of two, so the bits of the result only shift
    # (no further: two coils are combined by a root of a sum of squares, whose smallest terms would leave fp32's range)
    for k in (12, -12):
        s, _ = lib.recon(np.asfortranarray(data * np.float32(2.0 ** k)), adjoint=True, **fl)
        assert np.array_equal(s * np.float32(2.0 ** -k), got)


def _band_error(got, want, lo, hi):
    """Relative L2 error of `got` inside the spatial-frequency annulus lo <= |k| < hi of the image (the gridded k-space at those radii,
    up to deapodisation): where a per-tile fixed-point scale would show, long before the whole-image figure does."""
    e, w = np.fft.fftshift(np.fft.fft2(got - want)), np.fft.fftshift(np.fft.fft2(want))
    n = got.shape[0]
    k = np.hypot(*np.meshgrid(np.arange(n) - n // 2, np.arange(n) - n // 2))
    m = (k >= lo) & (k < hi)
    return float(np.linalg.norm(e[m]) / np.linalg.norm(w[m]))


def test_scatter_kernel_many_spokes_and_dynamic_range_on_64_tiles(oracle):
    """the fixed-point step of a tile is max(|d| dcf) * M * wsum / 2^31 with M the most spokes through one of the tile's
    blocks -- 0.76 of a window at the centre tiles, so 486 of this test's 640 spokes -- and a 64-tile next to the centre spans radii 5 .. 90,
    over which scanner-like data (magnitude ~ 1 / r^2, density compensation ~ r) falls by a factor of 16: the rim of such a tile is summed
    in steps ~ 16 * 486 * wsum / 2^31 of its own values.  Whole image against the oracle at the north_star's 1e-5, and the outer band of
    spatial frequencies (|k| >= 64 of 128: the gridded samples at radii >= 128 of 256) against the arc kernel's fp32 sums of the same band."""
    nro, npe = 512, 640
    data = synth.kspace(1, nro, npe, seed=9750)
    r = np.abs(np.arange(nro) - nro // 2).astype(np.float32)
    env = (1.0 / (1.0 + (r / 2.0) ** 2) + 1e-4).astype(np.float32)
    data = (data * env[None, None, :, None, None]).astype(np.complex64)
    data[0, 0, nro // 2 + 41, 7, 0] *= 300.0                       # a spike inside a centre 64-tile
    data = np.asfortranarray(data)
    fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    assert "grid_scatter_kernel" in _kernel_name(data.shape, **fl)
    got, dims = lib.recon(data, adjoint=True, **fl)
    assert (dims.nxos, dims.npe1work) == (512, 640)
    want, _ = oracle.recon(data, adjoint=1, golden=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    arc = _child(data, dict(TRON_GRID_KERNEL="arc"), **fl)
    g, w, a = got[0, 0, :, :, 0], want[0, 0, :, :, 0], arc[0, 0, :, :, 0]
    assert rel_l2(g, w) <= 1e-5
    for lo, hi in ((0, 32), (32, 64), (64, 128)):
        eg, ea = _band_error(g, w, lo, hi), _band_error(a, w, lo, hi)
        print(f"band {lo}-{hi}: scatter {eg:.3e}, arc {ea:.3e}")
        # every band on its own meets the whole-image bound, with room: 2.2 / 5.0 / 1.2 e-6 measured.  (The arc path's fp32 sums read 5.1 / 2.9 / 2.4 e-6
        # when this test was written -- most of that was the centre kernel's window edge on spoke 305, which this window holds (tests/test_gpu_arc.py); with
        # it mended fp32 reads 6-7e-7 in every band, and the fixed-point step of the tile that holds the spike is what is left of the difference.)
        assert eg <= 7e-6, (lo, hi, eg, ea)


@pytest.mark.parametrize("nc", [1, 2])
def test_scatter_kernel_rescales_its_sums_when_a_later_round_brings_larger_samples(oracle, nc):
    """A tile of more than 2 048 records is gridded in rounds of the angle-sorted spoke list, the fixed-point scale following the
    largest sample seen so far: spokes whose amplitude grows with their line angle by 2^0 .. 2^24 make every later round of the
    inner tiles shrink the scale and divide the sums gathered so far (grid_scatter_kernel, `ksh`)."""
    nro, npe = 256, 300                                            # 300 / (pi r) samples per unit area: > 2 048 per tile + halo inside r ~ 60
    data = synth.kspace(nc, nro, npe, seed=9800 + nc)
    PHI = np.float32(1.9416089796736116)
    ang = np.mod((PHI * np.arange(npe, dtype=np.float32)).astype(np.float64), np.pi)       # line angle of spoke pe (src/tron.cu:509, mod pi)
    amp = np.exp2(np.floor(ang / np.pi * 25.0)).astype(np.float32)                         # 2^0 .. 2^24 with the line angle
    data = np.asfortranarray((data * amp[None, None, None, :, None]).astype(np.complex64))
    fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    assert "grid_scatter_kernel" in _kernel_name(data.shape, **fl)
    got, _ = lib.recon(data, adjoint=True, **fl)
    again, _ = lib.recon(data, adjoint=True, **fl)
    assert np.array_equal(got, again)
    want, _ = oracle.recon(data, adjoint=1, golden=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    assert rel_l2(got, want) <= 1e-5


def test_scatter_kernel_first_row_of_a_footprint_when_k_minus_W_rounds(oracle):
    """Spoke 155 of the golden-angle series has (cos, sin) = (0.8, -0.6) in fp32 (a 3-4-5 triangle): its sample at r = 105 lies at
    ky = -63.0000038, and row -65 is inside its footprint by the reference's test (|ky - Y| = 1.9999962 < 2, src/tron.cu:341,
    516) although fl(ky - W) = -65 exactly.  A first build of the kernel took floor(k - W) + 1 for the first row and lost that
    row's weight: 4e-4 of this sample, 4e-6 of a whole 180-spoke slice."""
    nro, npe = 256, 180
    fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    data = np.zeros((1, 1, nro, npe, 1), dtype=np.complex64, order="F")
    data[0, 0, 128 + 105, 155, 0] = 1.0 - 0.5j
    got, _ = lib.recon(data, adjoint=True, **fl)
    want, _ = oracle.recon(data, adjoint=1, golden=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    assert rel_l2(got, want) <= 2e-6


@pytest.mark.parametrize("nt,half", [(3, 0), (5, 0), (3, 1)])
def test_scatter_kernel_odd_channel_counts(oracle, nt, half):
    """One coil x nt repetitions = an odd number of channels (the reference asserts one or an even number of COILS, src/tron.cu:963;
    nt > 1 is defined as nt separate runs): one channel per pass of the scatter kernel (blockIdx.y), 8- or 4-byte loads at the
    records' 24- / 40- / 12-byte stride.  The binned kernel took these plans until round 5."""
    nro, npe = 256, 110
    data = synth.kspace(1, nro, npe * 2, seed=9900 + nt, nt=nt)
    fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    src = data
    if half:
        h = np.stack([data.real, data.imag]).astype(np.float16)
        data = np.asfortranarray((h[0].astype(np.float32) + 1j * h[1].astype(np.float32)).astype(np.complex64))
        src = h
        fl["input_half"] = 1
    assert "grid_scatter_kernel" in _kernel_name(data.shape, **fl)
    got, _ = lib.recon(src, adjoint=True, **fl)
    again, _ = lib.recon(src, adjoint=True, **fl)
    assert np.array_equal(got, again)
    ofl = dict(golden=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    for t in range(nt):
        want, _ = oracle.recon(np.asfortranarray(data[:, t:t + 1]), adjoint=1, **ofl)
        assert rel_l2(got[:, t:t + 1], want) <= 1e-5, t


@pytest.mark.parametrize("nc,nro,npe,nz", [(1, 512, 402, 4), (1, 256, 300, 2), (3, 256, 120, 2)])
def test_scatter_kernel_tile_sizes_agree(oracle, nc, nro, npe, nz):
    """One channel per pass runs on 64 x 64 tiles of eight waves where the grid's centre is a corner of four of them (fewer samples
    handled twice, the per-slice overheads shared by four times the samples), else on 32 x 32 tiles of four; TRON_SCAT_TILE picks
    one.  The two differ in their fixed-point scales (per tile), not in what they compute: each within 1e-5 of the oracle, 2e-6 of
    each other."""
    nt = nc if nc > 1 else 1
    data = synth.kspace(1, nro, npe * nz, seed=9950 + nro + npe, nt=nt) if nt > 1 else synth.kspace(1, nro, npe * nz, seed=9950 + nro + npe)
    fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    out = {}
    for tile in ("32", "64"):
        out[tile] = _child(data, dict(TRON_GRID_KERNEL="scatter", TRON_SCAT_TILE=tile), **fl)
    assert rel_l2(out["32"], out["64"]) <= 2e-6
    got, _ = lib.recon(data, adjoint=True, **fl)
    assert np.array_equal(got, out["64"])                      # 64 is the plan's own choice on these grids
    ofl = dict(golden=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    for t in range(nt):
        want, _ = oracle.recon(np.asfortranarray(data[:, t:t + 1]), adjoint=1, **ofl)
        assert rel_l2(out["32"][:, t:t + 1], want) <= 1e-5 and rel_l2(out["64"][:, t:t + 1], want) <= 1e-5


_EMPTY_RUN_ORACLE = {}


@pytest.mark.parametrize("tile", ["32", "64"])
@pytest.mark.parametrize("nro,npe,nz", [(256, 12, 64), (512, 30, 32), (256, 6, 128)])
def test_scatter_kernel_empty_runs_between_the_slices_of_one_workgroup(oracle, tile, nro, npe, nz):
    """Few spokes per window: tiles that no spoke of window z crosses but some spoke of window z + 1 does.  A workgroup grids up to
    four consecutive slices of its tile and issues the next slice's requests (run table, member table) from inside the current
    slice's rounds -- an empty run has none (the arc kernel's round-4 bug in another coat, found by running the suite on 32 x 32
    tiles: a stale member table sent the sample loads out of bounds).  Both tile sizes, against the oracle."""
    data = synth.kspace(1, nro, npe * nz, seed=9990 + npe)
    fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    got = _child(data, dict(TRON_GRID_KERNEL="scatter", TRON_SCAT_TILE=tile), **fl)
    key = (nro, npe, nz)
    if key not in _EMPTY_RUN_ORACLE:                      # (the same data for both tile sizes: one reference run)
        # every slice from the bit-exact gather kernel (bit-identical gridding to the oracle, tests/test_gpu_parity.py: the whole volume
        # through the CPU oracle took 40 s for 128 slices), three slices from the oracle itself
        exact, _ = lib.recon(data, adjoint=True, kb_mode=lib.KB_EXACT, **fl)
        ora = {z: oracle.recon(data, adjoint=1, zfirst=z, zcount=1, golden=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)[0][..., z]
               for z in (0, nz // 2, nz - 1)}
        _EMPTY_RUN_ORACLE[key] = (exact, ora)
    exact, ora = _EMPTY_RUN_ORACLE[key]
    for z in range(nz):
        assert rel_l2(got[..., z], exact[..., z]) <= 1e-5, z
    for z, want in ora.items():
        assert rel_l2(got[..., z], want) <= 1e-5, z


def test_shapes_the_scatter_kernel_leaves_to_the_arc_kernel():
    os.environ.pop("TRON_GRID_KERNEL", None)              # the plan's own choice
    try:
        assert "grid_scatter_kernel" in _kernel_name((1, 1, 256, 100, 1), golden_angle=1, data_undersamp=0.39)
        assert "grid_arc_kernel" in _kernel_name((2, 1, 256, 100, 1), golden_angle=1, data_undersamp=0.39)                  # two fp32 channels
        assert "grid_scatter_kernel" in _kernel_name((2, 1, 256, 100, 1), golden_angle=1, data_undersamp=0.39, input_half=1)
    finally:
        os.environ["TRON_GRID_KERNEL"] = "scatter"
    assert "grid_arc_kernel" in _kernel_name((4, 1, 256, 100, 1), golden_angle=1, data_undersamp=0.39)
    assert "grid_arc_kernel" in _kernel_name((2, 1, 256, 100, 1), golden_angle=1, data_undersamp=0.39, kernwidth=2.5)     # six points per axis
    assert "grid_arc_kernel" in _kernel_name((2, 1, 256, 100, 1), golden_angle=1, data_undersamp=0.39, kernwidth=1.5)     # the inner points are not always inside their band
    assert "grid_scatter_kernel" in _kernel_name((2, 1, 256, 100, 1), golden_angle=1, data_undersamp=0.39)


def test_scatter_kernel_random_shapes_vs_the_bit_exact_gather():
    """Random one- and three-channel plans on grids that take the scatter kernel (W = 2; 64- and 32-tiles, resampled readouts, sliding
    windows, complex-half input, few to many spokes, linear and golden angles) against TRON_KB_EXACT on the GPU -- the order-preserving
    gather that is bit-identical to the reference's sums.  Fixed seed."""
    rng = np.random.default_rng(20261005)
    worst = 0.0
    for it in range(36):
        nt = int(rng.choice([1, 1, 1, 3]))
        nro = int(rng.choice([128, 192, 256, 256, 320, 512]))
        gridos = float(rng.choice([1.5, 2.0, 2.0, 2.0, 2.5, 3.0]))
        nxos = int((nro // 2) * gridos)
        if nxos % 64 != 0 or nxos < 128:
            gridos = 2.0
        npe = int(rng.integers(3, 40)) if rng.integers(0, 3) == 0 else int(rng.integers(40, 520))
        nz = int(rng.integers(1, 10))
        slide = int(rng.integers(1, npe + 1))
        golden = int(rng.integers(0, 4) != 0)
        half = bool(rng.integers(0, 4) == 0)
        fl = dict(golden_angle=golden, data_undersamp=(npe + 0.5) / nro, prof_slide=slide, gridos=gridos, skip_angles=int(rng.integers(0, 40)))
        data = synth.kspace(1, nro, npe + slide * (nz - 1), seed=7000 + it, nt=nt)
        src = data
        if half:
            h = np.stack([data.real, data.imag]).astype(np.float16)
            src = h
            fl["input_half"] = 1
        desc = f"nt={nt} nro={nro} os={gridos} npe={npe} nz={nz} slide={slide} G={golden} half={half}"
        name = _kernel_name(data.shape, **fl)
        assert "grid_scatter_kernel" in name, (desc, name)
        fast, _ = lib.recon(src, adjoint=True, kb_mode=lib.KB_FAST, **fl)
        exact, _ = lib.recon(src, adjoint=True, kb_mode=lib.KB_EXACT, **fl)
        e = rel_l2(fast, exact)
        worst = max(worst, e)
        assert e <= 1e-5, (desc, e)
    assert worst <= 1e-5
