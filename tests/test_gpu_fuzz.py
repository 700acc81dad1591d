"""Random-shape sweep of the fast GPU paths against the reference-order exact mode and, for small cases, the CPU
oracle (tests/fuzz_shapes.py).  A fixed seed keeps it reproducible; the sweep found the odd-nxos/2 store-alignment bug that
the hand-picked parity cases missed."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_fast_vs_exact_vs_oracle(oracle):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tron_fuzz", os.path.join(root, "tests", "fuzz_shapes.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    worst, failures = fuzz.run(40, 20261002, verbose=False)
    assert not failures, failures
    assert worst <= 1e-5


def test_random_shapes_on_scan_like_data(oracle):
    """The same sweep on k-space under a scanner's envelope and on smooth images (synth.scan_envelope): the samples next to the origin carry
    the result, which is how round 6 found the centre kernel's window edge (tests/test_gpu_arc.py); flat random fields average such a thing away."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tron_fuzz", os.path.join(root, "tests", "fuzz_shapes.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    worst, failures = fuzz.run(24, 20261004, verbose=False, scan=True)
    assert not failures, failures
    assert worst <= 1e-5


@pytest.mark.slow      # 112 s, nearly all of it the CPU oracle's CGNR; tests/test_gpu_cgnr.py, test_gpu_round2.py keep each feature's own parity tests
def test_random_shapes_round2_features_vs_oracle(oracle):
    """CGNR, Walsh combination, nt > 1, chunked / pinned host pipeline on random shapes, against the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tron_fuzz", os.path.join(root, "tests", "fuzz_shapes.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    worst, failures = fuzz.run2(48, 20261003, verbose=False)
    assert not failures, failures


def test_tiny_problem_sizes(oracle):
    """Readouts of 2..8 samples and images of 1..5 pixels: grids far smaller than one tile, single spokes."""
    import numpy as np
    import synth
    from conftest import rel_l2
    from tron_amd import lib
    for kb in (lib.KB_EXACT, lib.KB_FAST):
        for nro in (2, 3, 5, 8):
            for npe in (1, 7):
                for nc in (1, 2):
                    data = synth.kspace(nc, nro, npe, seed=nro * 100 + npe)
                    want, _ = oracle.recon(data, adjoint=1, golden=1, data_undersamp=10.0)
                    got, _ = lib.recon(data, adjoint=True, kb_mode=kb, golden_angle=1, data_undersamp=10.0)
                    assert got.shape == want.shape
                    assert not want.size or rel_l2(got, want) <= 1e-5, (nro, npe, nc, kb)
        for nx in (1, 2, 3, 5):
            for nc in (1, 2):
                img = synth.image(nc, nx, seed=nx)
                want, _ = oracle.recon(img, adjoint=0, golden=1)
                got, _ = lib.recon(img, adjoint=False, kb_mode=kb, golden_angle=1)
                assert got.shape == want.shape
                assert not want.size or rel_l2(got, want) <= 1e-5, (nx, nc, kb)
