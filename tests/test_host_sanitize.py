"""The host-only code that parses untrusted input (.ra reader / writer / converters, src/ra.cu's replacement) and the
host table builders, compiled with AddressSanitizer + UBSan (GPU sanitizers are not available; the host build is)
and driven through truncated and bit-flipped files by tests/native/ra_sanitize.cpp."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_ra_io_and_host_tables_under_sanitizers(tmp_path):
    exe = str(tmp_path / "ra_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "native", "ra_sanitize.cpp"),
           os.path.join(ROOT, "tron_amd", "csrc", "rawarray.cpp"),
           os.path.join(ROOT, "tron_amd", "csrc", "tron_hostmath.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert "ra_sanitize: ok" in r.stdout
