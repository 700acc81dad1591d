"""Random-shape sweep of the fast GPU paths against the reference-order exact mode and, for small cases, the CPU
oracle (tools/fuzz.py).  A fixed seed keeps it reproducible; the sweep found the odd-nxos/2 store-alignment bug that
the hand-picked parity cases missed."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_random_shapes_fast_vs_exact_vs_oracle(oracle):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tron_fuzz", os.path.join(root, "tools", "fuzz.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    worst, failures = fuzz.run(40, 20261002, verbose=False)
    assert not failures, failures
    assert worst <= 1e-5
