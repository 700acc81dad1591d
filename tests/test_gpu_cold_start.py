"""Cold-start faults must be SEEN.  tests/conftest.py runs every pipeline family in a child process before the first GPU test and
retries up to three times, so that a fault of the first GPU process on a fresh box does not take the session
with it; this file turns a failed first attempt into a failed test, and proves with an injected fault that it would."""
import os
import tempfile

import pytest

import conftest

pytestmark = pytest.mark.gpu


def test_the_first_gpu_process_of_this_session_ran_clean(_first_process_on_a_fresh_box):
    ok, msg = conftest.cold_start_verdict(_first_process_on_a_fresh_box)
    assert ok, msg


def test_an_injected_first_process_fault_is_reported():
    """TRON_DEBUG=cold_fault=<path> (tron_plan.cpp): the first plan creation that finds <path> missing fails, later ones run.
    The burn-in then needs a second attempt -- and the verdict says so instead of passing."""
    with tempfile.TemporaryDirectory() as tmp:
        attempts = conftest.run_burn_in(dict(TRON_DEBUG="cold_fault=" + os.path.join(tmp, "first_plan_failed")))
    assert len(attempts) == 2 and attempts[0]["rc"] not in (0, None) and attempts[1]["rc"] == 0, attempts
    assert "injected cold-start fault" in attempts[0]["stderr"]
    ok, msg = conftest.cold_start_verdict(attempts)
    assert not ok and "FAILED" in msg
