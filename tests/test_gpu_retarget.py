"""tron_plan_retarget (round 6): a continuing golden-angle acquisition on ONE plan.

The reference takes its angle index per kernel call -- `skip_angles + peoffset` into gridradial2d / degridradial2d
(src/tron.cu:509-511, 555-559, 629-630, the `-s` flag) -- so a later batch of spokes is just another call.  Here the angles live in
tables; a plan holds two sets, and tron_plan_retarget rebuilds the idle one on the device (sort by line angle, centre windows, run
tables: tron_traj_dev.hip, arc_prep_kernel) beside the work already queued.  What is checked:

  * the bytes are those of a plan CREATED with the new skip_angles -- for every gridding kernel family (arc + centre, scatter on 64-
    and 32-tiles, multi-pass windows, the binned and the bit-exact gather kernels, which only read the (cos, sin) table), for
    complex-half input, for the forward direction and for per-GPU workers' slice shares;
  * the double buffer: adjoint / retarget / adjoint / retarget / adjoint queued back to back with no synchronisation in between;
  * the oracle at the north_star's 1e-5 on retargeted angles, also where fp32 has long stopped resolving the golden angle
    (skip_angles = 5 000 000: PHI * float(index) has an ulp of 1 rad there, SURVEY Q6 -- replicated, not fixed);
  * linear angles: a no-op.
"""
import ctypes

import numpy as np
import pytest

from conftest import rel_l2
import synth
from tron_amd import lib

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _flat(a):
    return np.asfortranarray(a).reshape(-1, order="F")


def _fresh(data, skip, **flags):
    """A plan created with skip_angles = skip (what a retargeted plan must reproduce bit for bit)."""
    out, dims = lib.recon(data, adjoint=True, skip_angles=skip, **flags)
    return _flat(out), dims


def _plan(data, **flags):
    half = flags.get("input_half", 0)
    shape = data.shape[1:] if half else data.shape
    cfg = lib.default_config(adjoint=1, **flags)
    dims = lib.derive_dims(cfg, shape)
    return lib.Plan(cfg, dims), dims


CASES = [
    # nc, nro, spokes per slice, slices, flags, kernel the plan must run
    (8, 256, 201, 3, dict(golden_angle=1), "grid_arc_kernel"),
    (4, 128, 64, 5, dict(golden_angle=1, prof_slide=17), "grid_arc_kernel"),              # sliding windows
    (1, 256, 180, 3, dict(golden_angle=1), "grid_scatter_kernel"),                        # 64-tiles
    (1, 128, 70, 4, dict(golden_angle=1), "grid_scatter_kernel"),                         # 32-tiles (128 / 2 is no multiple of 64)
    (2, 256, 1300, 2, dict(golden_angle=1), "grid_arc_kernel"),                           # two passes of 650 spokes
    (2, 256, 100, 2, dict(golden_angle=1, gridos=1.5), "grid_arc_kernel"),                # resampled readout
    (2, 256, 90, 2, dict(golden_angle=1, kernwidth=3.5), "grid_tile_kernel"),             # W > 3: the gather kernel (reads the (cos, sin) table only)
    (2, 64, 40, 3, dict(golden_angle=1), "grid_binned_kernel"),                           # grid too small for centre relief: binned kernel
    (8, 256, 120, 2, dict(golden_angle=1, kb_mode=lib.KB_EXACT), "grid_tile_kernel"),     # bit-exact mode
]


@pytest.mark.parametrize("nc,nro,npe,nz,flags,kernel", CASES)
def test_retargeted_plan_equals_a_fresh_plan_bit_for_bit(oracle, nc, nro, npe, nz, flags, kernel):
    slide = flags.get("prof_slide", npe)
    data = synth.kspace(nc, nro, npe + slide * (nz - 1), seed=9600 + nc + nro + npe)
    fl = dict(flags)
    fl.setdefault("prof_slide", npe)
    fl["data_undersamp"] = (npe + 0.5) / nro
    skips = (0, 977, 12 * npe * nz + 5, 0)
    plan, dims = _plan(data, **fl)
    with plan:
        assert kernel in plan.grid_kernel_name()
        assert dims.nz == nz and dims.npe1work == npe
        for k, skip in enumerate(skips):
            if k:
                plan.retarget(skip)
            got = plan.recon(_flat(data))
            want, _ = _fresh(data, skip, **fl)
            assert np.array_equal(got, want), (k, skip)
            assert kernel in plan.grid_kernel_name()
    # ... and the oracle on the last non-zero angle index (-s, src/tron.cu:509)
    oflags = {("golden" if k == "golden_angle" else k): v for k, v in fl.items() if k != "kb_mode"}
    want, _ = oracle.recon(data, adjoint=1, zfirst=nz - 1, zcount=1, skip_angles=skips[2], **oflags)
    got, _ = _fresh(data, skips[2], **fl)
    img = dims.nx * dims.ny
    assert rel_l2(got[(nz - 1) * img: nz * img], _flat(want[..., nz - 1])) <= TOL


@pytest.mark.parametrize("nc", [8, 1])
def test_retargets_queued_back_to_back_without_a_synchronisation(nc):
    """adjoint / retarget / adjoint / retarget / adjoint / retarget / adjoint on device-resident data, one tron_plan_sync at the end: the
    build of set B runs beside the gridding that still reads set A, the next build of A must wait for A's last reader (ev_released), and
    every output holds its own angles' bytes."""
    nro, npe, nz = 256, 150, 6
    data = synth.kspace(nc, nro, npe * nz, seed=9700 + nc)
    fl = dict(golden_angle=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    skips = (0, 4001, 90210, 4001)
    plan, dims = _plan(data, **fl)
    with plan:
        d_in = lib.DeviceBuffer.from_numpy(_flat(data))
        outs = [lib.DeviceBuffer(dims.out_bytes) for _ in skips]
        for k, skip in enumerate(skips):
            if k:
                plan.retarget(skip)
            for _ in range(3):                                   # several launches per set: work queued well ahead of the build
                plan.adjoint_device(outs[k].ptr, d_in.ptr, 0, nz, combine=1)
        plan.sync()
        got = [o.to_numpy(np.complex64, dims.out_bytes // 8) for o in outs]
    for k, skip in enumerate(skips):
        want, _ = _fresh(data, skip, **fl)
        assert np.array_equal(got[k], want), (k, skip)
    assert np.array_equal(got[1], got[3])


def test_two_retargets_in_a_row_and_a_retarget_nobody_uses():
    nro, npe, nz = 256, 100, 3
    data = synth.kspace(4, nro, npe * nz, seed=9710)
    fl = dict(golden_angle=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro)
    plan, dims = _plan(data, **fl)
    with plan:
        plan.retarget(111)
        plan.retarget(222)                   # the first build is finished and dropped: the sets alternate
        plan.retarget(333)
        got = plan.recon(_flat(data))
        want, _ = _fresh(data, 333, **fl)
        assert np.array_equal(got, want)
        plan.retarget(444)                   # never used: the plan is destroyed with a build in flight
    t = None
    plan, _ = _plan(data, **fl)
    with plan:
        plan.retarget(5)
        t = plan.retarget_times()
    assert t["call"] >= t["trig"] >= 0.0


def test_retarget_where_fp32_no_longer_resolves_the_golden_angle(oracle):
    """skip_angles = 5 000 000: PHI * float(index) is ~ 9.7e6 with an ulp of 1.0 rad -- the 402 spokes of a window fall onto a few dozen
    distinct directions (the reference's arithmetic, to be replicated).  Runs pile up on few angles (ties keep acquisition
    order; a tile's run holds many spokes of ONE direction); whatever kernel the tables allow, the bytes are a fresh plan's and the
    image is the oracle's."""
    nro, npe, nz, skip = 512, 402, 2, 5_000_000
    for nc in (8, 1):
        data = synth.kspace(nc, nro, npe * nz, seed=9720 + nc)
        fl = dict(golden_angle=1, prof_slide=npe, data_undersamp=0.7852)
        plan, dims = _plan(data, **fl)
        with plan:
            assert dims.npe1work == npe
            plan.retarget(skip)
            got = plan.recon(_flat(data))
            name = plan.grid_kernel_name()
        fplan, _ = _plan(data, skip_angles=skip, **fl)
        with fplan:
            want = fplan.recon(_flat(data))
            fname = fplan.grid_kernel_name()
        # run tables that overflow send a retargeted plan straight to the binned kernel, a fresh one down its formulations one at a time:
        # the same kernel means the same bytes, different kernels the fast kernels' mutual 2e-6
        if name == fname:
            assert np.array_equal(got, want), (nc, name)
        else:
            assert rel_l2(got, want) <= 2e-6, (nc, name, fname)
        ref, _ = oracle.recon(data, adjoint=1, zfirst=1, zcount=1, golden=1, prof_slide=npe, data_undersamp=0.7852, skip_angles=skip)
        img = dims.nx * dims.ny
        assert rel_l2(got[img: 2 * img], _flat(ref[..., 1])) <= TOL, nc


def test_retarget_complex_half_input_and_forward_direction(oracle):
    # complex-half k-space on the arc kernel (6 coils: the whole-body shape's coil count)
    nro, npe, nz = 256, 120, 2
    data = synth.kspace(6, nro, npe * nz, seed=9730)
    halves = _flat(data).view(np.float32).astype(np.float16)
    h = halves.reshape((2,) + data.shape, order="F")
    fl = dict(golden_angle=1, prof_slide=npe, data_undersamp=(npe + 0.5) / nro, input_half=1)
    plan, dims = _plan(h, **fl)
    with plan:
        plan.retarget(3210)
        got = plan.recon(_flat(h))
    want, _ = lib.recon(h, adjoint=True, skip_angles=3210, **fl)
    assert np.array_equal(got, _flat(want))
    # forward: degridradial2d's angles, PHI * (pe + skip) (src/tron.cu:555)
    img = synth.image(2, 128, seed=9731)
    ffl = dict(golden_angle=1, data_undersamp=0.25)
    cfg = lib.default_config(adjoint=0, **ffl)
    fdims = lib.derive_dims(cfg, img.shape)
    with lib.Plan(cfg, fdims) as fplan:
        base = fplan.recon(_flat(img)).copy()
        fplan.retarget(777)
        got = fplan.recon(_flat(img)).copy()
        fplan.retarget(0)
        again = fplan.recon(_flat(img))
    want, _ = lib.recon(img, adjoint=False, skip_angles=777, **ffl)
    assert np.array_equal(got, _flat(want)) and np.array_equal(again, base) and not np.array_equal(got, base)
    ref, _ = oracle.recon(img, adjoint=0, golden=1, data_undersamp=0.25, skip_angles=777)
    assert rel_l2(got, _flat(ref)) <= TOL


def test_retarget_linear_angles_is_a_no_op():
    data = synth.kspace(2, 256, 100, seed=9740)
    fl = dict(golden_angle=0, data_undersamp=(100 + 0.5) / 256)
    plan, dims = _plan(data, **fl)
    with plan:
        base = plan.recon(_flat(data)).copy()
        plan.retarget(123456)
        assert np.array_equal(plan.recon(_flat(data)), base)


def test_retarget_rejects_a_null_plan():
    assert lib.load().tron_plan_retarget(None, 1) == lib.TRON_ERR_INVALID
    assert lib.load().tron_plan_retarget_times(None, (ctypes.c_double * 2)()) == lib.TRON_ERR_INVALID


def test_retarget_under_cgnr_walsh_and_repetitions(oracle):
    """The pipelines that sit on top of the adjoint / forward pair read the plan's current angle tables too: CGNR (the forward operator
    inside the iteration uses the slice's own golden angles, F4), Walsh combination and nt > 1 (the uncombined fused tail).  A retargeted
    plan against a fresh one bit for bit, CGNR also against the oracle's restatement at the new angle index."""
    nro, npe, nz, skip = 256, 100, 2, 2468
    us = (npe + 0.5) / nro
    for nc, nt, extra in ((4, 1, dict(niter=2)), (4, 1, dict(coil_combine=1, walsh_patch=1)), (2, 3, dict())):
        data = synth.kspace(nc, nro, npe * nz, seed=9760 + nc + nt, nt=nt)
        fl = dict(golden_angle=1, prof_slide=npe, data_undersamp=us, **extra)
        plan, dims = _plan(data, **fl)
        with plan:
            plan.retarget(skip)
            got = plan.recon(_flat(data))
        want, _ = _fresh(data, skip, **fl)
        assert np.array_equal(got, want), (nc, nt, extra)
        if extra.get("niter"):
            ref, _ = oracle.recon_cgnr(data, 2, golden=1, prof_slide=npe, data_undersamp=us, skip_angles=skip)
            assert rel_l2(got, _flat(ref)) <= TOL
