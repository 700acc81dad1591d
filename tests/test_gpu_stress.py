"""Regression guard for the lazy code-object-load fault of round 1: a kernel launched on the plan's
non-blocking stream right after process start could run before its code object was resident ("memory access fault
... address (nil)", ~15 % of processes).  tron_plan_create now force-loads every translation unit; this test starts
many short-lived processes that create a plan and launch a metric-size reconstruction immediately."""
import os
import subprocess

import numpy as np
import pytest

import synth
from tron_amd import ra

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRON = os.path.join(ROOT, "tron_amd", "bin", "tron")
NPROC = 40


@pytest.mark.timeout(1200)
def test_fresh_processes_launch_immediately_without_faults(tmp_path):
    data = synth.kspace(8, 512, 402, seed=synth.SEED_BASE + 31)        # one metric-shape slice, 8 coils
    src = str(tmp_path / "in.ra")
    ra.write(src, data)
    first = None
    for i in range(NPROC):
        dst = str(tmp_path / f"out{i % 2}.ra")
        r = subprocess.run([TRON, "-a", "-G", "-u", "0.7852", src, dst], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            # name the failing stage: same run, synchronising after every launch
            dbg = subprocess.run([TRON, "-a", "-G", "-u", "0.7852", src, dst], capture_output=True, text=True, timeout=120,
                                 env=dict(os.environ, TRON_DEBUG="sync"))
            pytest.fail(f"process {i} of {NPROC} failed (rc {r.returncode}): {r.stderr[-400:]}\n"
                        f"TRON_DEBUG=sync rerun rc {dbg.returncode}: {dbg.stderr[-400:]}")
        out = open(dst, "rb").read()
        if first is None:
            first = out
            img = ra.read(dst)
            assert img.shape == (1, 1, 256, 256, 1) and np.isfinite(img).all() and np.abs(img).max() > 0
        else:
            assert out == first, f"process {i}: output bytes differ from the first run"


def test_plans_created_used_retargeted_and_destroyed_in_random_order():
    """tests/soak.py: plans of random shapes (adjoint and forward, fp32 and complex-half, flat and scan-like k-space, scaled by 2^-40 .. 2^40,
    host arrays on the heap and in mappings of their own, pinned or pageable) created, run on slice sub-ranges, retargeted and destroyed in
    random order in one process: whatever a plan returns is, bit for bit, what a plan created fresh for that job returns (linear-angle
    slice groups: to fp32 rounding).  Round 6 ran 19 000 operations of it clean (CGNR, the Walsh combination, the exact kernels, other oversampling ratios and window widths among them); the suite keeps 600."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tron_soak", os.path.join(ROOT, "tests", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    with np.errstate(over="ignore"):
        failures = soak.run(600, 20261005)
    assert not failures, failures


def test_the_same_from_three_threads_at_once():
    """The soak above from three threads of one process, each with its own plans (ctypes releases the interpreter lock inside the library):
    plans are independent objects, what the library shares between them (code objects, the launchers' occupancy caches, the last-error
    string per thread) must not show.  Round 6 ran 8 threads x 800 operations clean."""
    import importlib.util
    import threading
    spec = importlib.util.spec_from_file_location("tron_soak", os.path.join(ROOT, "tests", "soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    results = {}

    def work(seed):
        try:
            with np.errstate(over="ignore"):
                results[seed] = soak.run(150, seed)
        except Exception as e:                                   # noqa: BLE001 -- whatever it is, it is the finding
            results[seed] = [repr(e)]

    threads = [threading.Thread(target=work, args=(20261010 + i,)) for i in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(results) == [20261010, 20261011, 20261012] and not any(results.values()), results


def test_a_plan_that_does_not_fit_the_device_is_refused_cleanly():
    """A work grid of 1.2 TB, then one of 300 GiB (just over the device's 288 GB): tron_plan_create returns TRON_ERR_NOMEM with the size in
    the message, leaves nothing allocated, and the next plan of the process works."""
    from tron_amd import lib
    for nc, chunk in ((64, 600), (8, 1200)):
        cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=0.05, prof_slide=100, chunk_slices=chunk)
        dims = lib.derive_dims(cfg, (nc, 1, 2048, 100 * chunk, 1))
        with pytest.raises(lib.TronError, match="cannot allocate"):
            lib.Plan(cfg, dims)
    data = synth.kspace(8, 256, 200, seed=5)
    fl = dict(golden_angle=1, data_undersamp=100.5 / 256, prof_slide=100)
    a, dims = lib.recon(data, adjoint=True, **fl)
    b, _ = lib.recon(data, adjoint=True, **fl)
    assert dims.nz == 2 and np.isfinite(a).all() and np.array_equal(a, b)
