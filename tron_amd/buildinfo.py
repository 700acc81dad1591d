"""Identity of the kernel sources a measurement was taken on.  The GPU box receives a snapshot
without .git, so captures (rocprof PMC traffic) are stamped with a hash of the sources under tron_amd/csrc that decide the device work instead
of a commit id; bench.py recomputes it and refuses a capture taken on other sources."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


# Host-buffer staging, the command-line driver and the .ra reader decide nothing about what the kernels do on device-resident
# data (what a capture measures): an edit there leaves the captures valid.
_NOT_DEVICE_WORK = {"tron_hostio.cpp", "tron_main.cpp", "rawarray.cpp", "exports.map"}


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_CSRC, "*"))):
        if os.path.isfile(f) and os.path.basename(f) not in _NOT_DEVICE_WORK:
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
