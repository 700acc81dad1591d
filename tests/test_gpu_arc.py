"""The arc gridding kernel (tron_grid_arc.hip, round 3): every shape it takes, against the oracle AND against the binned kernel
it replaced (TRON_GRID_KERNEL=binned), and the plans it must NOT take.

What is checked is the reference's gridradial2d + pipeline (src/tron.cu:465-536, 623-655) through the oracle at the
north_star's 1e-5 relative L2, and agreement of the two fast kernels at 2e-6 (they differ in summation order and in how the
Kaiser-Bessel window is evaluated: table vs polynomial)."""
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


# One-channel plans take grid_scatter_kernel since round 5 (tests/test_gpu_scatter.py); this module keeps testing the arc kernel's
# one-coil instantiation, which TRON_GRID_KERNEL=arc (read at plan creation) still selects.
@pytest.fixture(autouse=True, scope="module")
def _arc_kernel_for_one_channel_too():
    # (TRON_SLICES_PER_PASS=0: linear-angle plans with few channels otherwise grid several slices per pass of the BINNED kernel)
    old = {k: os.environ.get(k) for k in ("TRON_GRID_KERNEL", "TRON_SLICES_PER_PASS")}
    os.environ["TRON_GRID_KERNEL"] = "arc"
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


def _binned(data, **flags):
    """The same reconstruction with the binned kernel on every tile: a child process, because the switch is read once."""
    return _child(data, dict(TRON_GRID_KERNEL="binned"), **flags)


def _child(data, env_extra, **flags):
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
    (8, 256, 201, 3, dict(golden_angle=1)),                       # 256^2 grid, centre = tile corner
    (2, 256, 150, 4, dict(golden_angle=1, prof_slide=37)),         # sliding windows, 2 coils
    (4, 128, 64, 5, dict(golden_angle=1, skip_angles=7)),            # smallest grid the kernel takes (4 x 4 tiles)
    (6, 256, 100, 2, dict(golden_angle=0)),                         # linear angles (6 coils: no slice groups)
    (1, 256, 180, 3, dict(golden_angle=1)),                          # one coil: 4-byte LDS-DMA planes
    (8, 256, 120, 2, dict(golden_angle=1, kernwidth=1.5)),          # W = 1.5 (the narrowest family of widths with a pair table: W > 1)
    (2, 256, 120, 2, dict(golden_angle=0, kernwidth=1.25)),         # W = 1.25, linear angles: samples at exactly |x| = W on the axis spokes
    (2, 256, 120, 2, dict(golden_angle=1, kernwidth=3.0)),          # W = 3
    (4, 256, 90, 2, dict(golden_angle=1, kernwidth=2.5)),           # fractional W
    (8, 1024, 60, 1, dict(golden_angle=1)),                       # 1024^2 grid, few spokes
    (12, 256, 100, 2, dict(golden_angle=1)),                      # two coil chunks (6 + 6; centre kernel 8 + 4)
    (10, 256, 80, 2, dict(golden_angle=1)),                       # a partial last chunk (6 + 4; centre kernel 8 + 2)
    (16, 256, 60, 1, dict(golden_angle=0)),                       # 8 + 8, linear angles
    # nro != nxos (-o other than 2): radius r reads sample (r nro) / nxos, truncating towards zero (src/tron.cu:517)
    (2, 256, 100, 2, dict(golden_angle=1, gridos=1.5)),           # 192^2 grid: four samples per three radii (every fourth sample unused)
    (8, 256, 120, 2, dict(golden_angle=1, gridos=3.0)),           # 384^2 grid: two samples per three radii (a sample gridded twice)
    (4, 256, 90, 2, dict(golden_angle=0, gridos=2.5)),            # 320^2 grid, linear angles
    (1, 256, 100, 2, dict(golden_angle=1, gridos=1.5)),           # one coil
    # more than 1 024 spokes per window: the arc kernel runs in passes over <= 1 024 spokes, the later ones adding to the grid
    (8, 256, 900, 2, dict(golden_angle=1)),                       # two passes of 450 (813 .. 1 024 spokes: the binned kernel until round 6)
    (2, 256, 1300, 2, dict(golden_angle=1)),                      # two passes of 650
    (8, 128, 2100, 2, dict(golden_angle=1, prof_slide=700)),      # three passes of 700, sliding windows
    (4, 256, 1030, 1, dict(golden_angle=0)),                      # two passes of 515, linear angles
    (2, 256, 1100, 1, dict(golden_angle=1, gridos=1.5)),          # passes AND a resampled readout
]


@pytest.mark.parametrize("nc,nro,npe,nz,flags", CASES)
def test_arc_kernel_vs_oracle_and_binned(oracle, nc, nro, npe, nz, flags):
    slide = flags.get("prof_slide", npe)
    data = synth.kspace(nc, nro, npe + slide * (nz - 1), seed=9000 + nc + nro + npe)
    fl = dict(flags)
    fl.setdefault("prof_slide", npe)
    fl["data_undersamp"] = (npe + 0.5) / nro          # npe1work = int(data_undersamp * nro), src/tron.cu:916
    assert "grid_arc_kernel" in _kernel_name(data.shape, **fl)
    got, dims = lib.recon(data, adjoint=True, **fl)
    assert dims.nz == nz and dims.npe1work == npe
    oflags = {("golden" if k == "golden_angle" else k): v for k, v in fl.items()}
    for z in sorted({0, nz // 2, nz - 1}):
        want, _ = oracle.recon(data, adjoint=1, zfirst=z, zcount=1, **oflags)
        assert rel_l2(got[..., z], want[..., z]) <= 1e-5, z
    other = _binned(data, **fl)
    assert rel_l2(got, other) <= 2e-6


def test_arc_kernel_half_input_and_determinism(oracle):
    """complex-half k-space, 4 and 8 coils (the halves are converted in LDS, in place); identical bits run to run.  The third case
    has nro != nxos (src/tron.cu:517), the fourth more than 1 024 spokes per window."""
    # 6 and 10 coils (round 5): 24- and 40-byte records, the last 16-byte piece of a record read 8 bytes early (coils 2..5 / 6..9);
    # 6 coils is the whole-body shape (src/RUNME3_tron_grid_all.sh:10)
    for nc, npe, extra in ((4, 140, {}), (8, 140, {}), (4, 140, dict(gridos=3.0)), (4, 1040, {}),     # (4, 1040): two passes over the spokes
                           (6, 140, {}), (10, 140, {}), (6, 140, dict(gridos=3.0)), (14, 100, {})):
        data = synth.kspace(nc, 256, npe * 2, seed=9100 + nc)
        h = np.stack([data.real, data.imag]).astype(np.float16)
        fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / 256, prof_slide=npe, **extra)
        assert "grid_arc_kernel" in _kernel_name(data.shape, input_half=1, **fl)
        a, dims = lib.recon(h, adjoint=True, input_half=1, **fl)
        b, _ = lib.recon(h, adjoint=True, input_half=1, **fl)
        assert np.array_equal(a, b)
        rounded = (h[0].astype(np.float32) + 1j * h[1].astype(np.float32)).astype(np.complex64)
        want, _ = oracle.recon(rounded, adjoint=1, golden=1, data_undersamp=(npe + 0.5) / 256, prof_slide=npe, **extra)
        assert rel_l2(a, want) <= 1e-5


@pytest.mark.parametrize("shape,flags,why", [
    ((1, 3, 256, 100, 1), dict(golden_angle=1, data_undersamp=0.39), "odd channel count (1 coil x 3 repetitions; the reference takes 1 or an even number of coils)"),
    ((2, 1, 160, 100, 1), dict(golden_angle=1, data_undersamp=0.63), "grid centre inside a tile (nxos 160)"),
    ((2, 1, 64, 40, 1), dict(golden_angle=1, data_undersamp=0.625), "grid smaller than 4 x 4 tiles"),
    ((8, 1, 256, 120, 1), dict(golden_angle=1, data_undersamp=0.47, kernwidth=1.0), "W = 1: window B's support starts at the table's origin (build_kb_pair_lut)"),
    ((2, 1, 256, 120, 1), dict(golden_angle=1, data_undersamp=0.47, kernwidth=0.5), "W < 1: no Kaiser-Bessel pair table (the round-3 table was read out of bounds here)"),
    ((2, 1, 256, 120, 1), dict(golden_angle=1, data_undersamp=0.47, kernwidth=2.3), "W 2^k is no integer: the window's support would not end on a table piece"),
])
def test_shapes_the_arc_kernel_leaves_to_the_binned_kernel(oracle, shape, flags, why):
    assert "grid_arc_kernel" not in _kernel_name(shape, **flags), why
    data = synth.kspace(shape[0], shape[2], shape[3], seed=9200 + shape[2], nt=shape[1])
    got, _ = lib.recon(data, adjoint=True, **flags)
    oflags = {("golden" if k == "golden_angle" else k): v for k, v in flags.items()}
    if shape[1] == 1:
        want, _ = oracle.recon(data, adjoint=1, **oflags)
        assert rel_l2(got, want) <= 1e-5, why
    else:                                       # nt > 1 is defined as nt separate runs (test_repetitions_nt_gt_1)
        for t in range(shape[1]):
            want, _ = oracle.recon(np.asfortranarray(data[:, t:t + 1]), adjoint=1, **oflags)
            assert rel_l2(got[:, t:t + 1], want) <= 1e-5, (why, t)


def test_a_workers_plan_holds_only_its_own_run_tables():
    """tron_recon_radial2d_multi: three workers on one GPU, each with a plan (and arc run tables) for its own slice block,
    give the single-plan bytes."""
    data = synth.kspace(2, 256, 90 + 30 * 8, seed=9300)
    fl = dict(golden_angle=1, data_undersamp=0.3516, prof_slide=30)
    one, dims = lib.recon(data, adjoint=True, **fl)
    assert dims.nz == 9
    multi, _ = lib.recon_multi(data, adjoint=True, devices=[0, 0, 0], **fl)
    assert np.array_equal(one, multi)


@pytest.mark.parametrize("nc,nro,npe,nz", [(8, 256, 12, 64), (2, 512, 30, 32), (1, 256, 16, 64)])
def test_empty_runs_between_the_slices_of_one_workgroup(oracle, nc, nro, npe, nz):
    """Few spokes per window: rim tiles that no spoke of window z crosses but some spoke of window z + 1 does.  A workgroup grids
    2 (launches of >= 32 slices) or 4 (>= 64) consecutive slices of its tile and asks for the next slice's run table one slice
    ahead; round 4 issued that request inside the batch loop, which an empty run never enters, and slice z + 1 of such a tile
    was gridded with slice z's empty table.  One launch against one slice per workgroup (TRON_ARC_ZPER=1), bit
    for bit, and against the reference's sums (src/tron.cu:465-536)."""
    data = synth.kspace(nc, nro, npe * nz, seed=9400 + npe)
    fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
    assert "grid_arc_kernel" in _kernel_name(data.shape, **fl)
    got, dims = lib.recon(data, adjoint=True, **fl)
    assert dims.nz == nz
    one = _child(data, dict(TRON_ARC_ZPER="1"), **fl)
    assert np.array_equal(got, one)
    # every slice against the bit-exact gather kernel (bit-identical gridding to the oracle, tests/test_gpu_parity.py; the whole
    # volume through the CPU oracle took 20 s of this test's 20), three slices against the oracle itself
    exact, _ = lib.recon(data, adjoint=True, kb_mode=lib.KB_EXACT, **fl)
    for z in range(nz):
        assert rel_l2(got[..., z], exact[..., z]) <= 1e-5, z
    for z in (0, nz // 2, nz - 1):
        want, _ = oracle.recon(data, adjoint=1, zfirst=z, zcount=1, golden=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe)
        assert rel_l2(got[..., z], want[..., z]) <= 1e-5, z


@pytest.mark.parametrize("nc", [1, 2, 8])
def test_k_space_whose_energy_sits_at_the_centre_and_a_spoke_next_to_an_axis(oracle, nc):
    """Round 6: the window jumps by 6e-4 of its peak at |k - X| = W, and which side a sample is on is the reference's own fp32
    subtraction, k - X (src/tron.cu:516).  The centre kernel took the distance to a block's second column from the first, (k - X0) - 1:
    the r = 1 sample of a spoke 5e-4 rad off an axis (k = 0.99999988: spokes 305, 2817, 3122, 3427 ... of a golden-angle run, one slice
    of 402 spokes in four has one) then lay outside where the reference has it inside, and on k-space whose energy sits at the centre --
    every scan -- a slice came out at 1.1e-5 to 1.9e-5 of the oracle (flat random data hides it: 4e-7).  Windows that hold those spokes,
    every slice against the north_star's 1e-5 with room to spare."""
    nro, npe = 256, 160
    data = synth.kspace_scan(nc, nro, npe * 2, seed=9700 + nc)
    for skip in (160, 2700, 3000):                                             # (spokes 305 | 2817 | 3122 in the first window)
        fl = dict(golden_angle=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe, skip_angles=skip)
        got, dims = lib.recon(data, adjoint=True, **fl)
        want, _ = oracle.recon(data, adjoint=1, golden=1, data_undersamp=(npe + 0.5) / nro, prof_slide=npe, skip_angles=skip)
        assert dims.nz == 2
        for z in range(2):
            assert rel_l2(got[..., z], want[..., z]) <= 3e-6, (skip, z)
