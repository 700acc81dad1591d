"""Which gridding kernel a plan runs (`tron_plan_grid_kernel_name`): the table at the top of README.md, row by row, and the shapes next to the
run tables' limits (a flat-table overflow used to send a plan straight to the 2x slower binned kernel, reported on
stderr only; the thresholds -- 512 run entries, 32 767 records per run, the member tables' group cap -- must not regress silently).
Plan creation only: no k-space is gridded here (the kernels' parity tests are tests/test_gpu_arc.py, test_gpu_scatter.py, ...)."""
import pytest

from tron_amd import lib

pytestmark = pytest.mark.gpu


def _name(shape, **flags):
    cfg = lib.default_config(adjoint=1, **flags)
    dims = lib.derive_dims(cfg, shape)
    with lib.Plan(cfg, dims) as plan:
        return plan.grid_kernel_name()


def _us(npe, nro):
    return (npe + 0.5) / nro          # npe1work = int(nro * data_undersamp), src/tron.cu:916


README_ROWS = [
    # (nc, nro, npe, flags) -> kernel                                                      README.md, "Which gridding kernel a plan runs"
    ((1, 512, 402, dict(golden_angle=1)), "grid_scatter_kernel"),                           # fp32, 1 channel, W = 2
    ((1, 512, 402, dict(golden_angle=1, input_half=1)), "grid_scatter_kernel"),             # complex-half, 1 channel
    ((1, 256, 120, dict(golden_angle=1, kernwidth=2.5)), "grid_arc_kernel"),                # 1 channel, other widths: arc for fp32 ...
    ((1, 256, 120, dict(golden_angle=1, kernwidth=2.5, input_half=1)), "grid_binned_kernel"),    # ... binned for complex-half
    ((2, 512, 402, dict(golden_angle=1)), "grid_arc_kernel"),                               # fp32, even counts
    ((8, 512, 402, dict(golden_angle=1)), "grid_arc_kernel"),
    ((2, 512, 402, dict(golden_angle=1, input_half=1)), "grid_scatter_kernel"),             # complex-half, 2 channels
    ((2, 256, 120, dict(golden_angle=1, input_half=1, kernwidth=2.5)), "grid_binned_kernel"),
    ((6, 512, 402, dict(golden_angle=1, input_half=1)), "grid_arc_kernel"),                 # complex-half, even counts >= 4 (6, 10, ... since round 5)
    ((8, 256, 120, dict(golden_angle=1, kernwidth=1.0)), "grid_binned_kernel"),             # W <= 1
    ((8, 256, 120, dict(golden_angle=1, kernwidth=2.3)), "grid_binned_kernel"),             # W 2^k no integer
    ((8, 256, 120, dict(golden_angle=1, kernwidth=3.5)), "grid_tile_kernel"),               # W > 3
    ((8, 160, 60, dict(golden_angle=1)), "grid_binned_kernel"),                             # a grid that is no multiple of 64
    ((8, 512, 402, dict(golden_angle=1, kb_mode=lib.KB_EXACT)), "grid_tile_kernel"),        # the reference's sums bit for bit
]


@pytest.mark.parametrize("case,kernel", README_ROWS, ids=[f"{c[0]}ch-nro{c[1]}-{'half-' if c[3].get('input_half') else ''}W{c[3].get('kernwidth', 2.0)}-{k}" for c, k in README_ROWS])
def test_readme_kernel_table(case, kernel):
    nc, nro, npe, flags = case
    assert kernel in _name((nc, 1, nro, npe * 2, 1), data_undersamp=_us(npe, nro), prof_slide=npe, **flags)


def test_linear_angle_slice_groups_and_odd_channel_counts():
    assert "linear-angle slice groups" in _name((1, 1, 512, 402 * 4, 1), golden_angle=0, data_undersamp=_us(402, 512), prof_slide=402)
    assert "grid_scatter_kernel" in _name((1, 3, 256, 120 * 2, 1), golden_angle=1, data_undersamp=_us(120, 256), prof_slide=120)   # one coil x nt = 3


@pytest.mark.parametrize("npe,kernel", [(640, "grid_scatter_kernel"), (800, "grid_scatter_kernel"), (804, "grid_scatter_kernel"), (812, "grid_scatter_kernel"),
                                        (900, "grid_scatter_kernel"), (1024, "grid_scatter_kernel"), (1300, "grid_scatter_kernel")])
def test_one_channel_plans_next_to_the_run_tables_limits(npe, kernel, capfd):
    """One channel, 512^2 grid: 640 spokes per window is the most for which the plan leaves the centre kernel only |r| < 5 (a centre tile's run then
    holds 0.76 x 640 = 486 of its 512 entries); above that the radius is 14 again; 800 is the 64-tiles' limit (0.59 x 800 = 472 entries, ~30 k of
    32 767 records); windows of more than 812 spokes go in passes of at most 812 (until round 6: of more than 1 024, and 813 .. 1 024 spokes overflowed
    the run tables and fell back to the binned kernel).  None of them may fall back, and none may say so on stderr."""
    name = _name((1, 1, 512, npe * 2, 1), golden_angle=1, data_undersamp=_us(npe, 512), prof_slide=npe)
    err = capfd.readouterr().err
    assert kernel in name, (npe, name, err)
    assert "overflow" not in err, err


@pytest.mark.parametrize("nc,npe", [(8, 804), (8, 900), (8, 1024), (8, 2100), (2, 4096)])
def test_many_spokes_stay_on_the_arc_kernel(nc, npe, capfd):
    name = _name((nc, 1, 256, npe * 2, 1), golden_angle=1, data_undersamp=_us(npe, 256), prof_slide=npe)
    err = capfd.readouterr().err
    assert "grid_arc_kernel" in name and "overflow" not in err, (name, err)
