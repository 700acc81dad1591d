"""Identity of the kernel sources a measurement was taken on.  The GPU box receives a snapshot
without .git, so captures (rocprof PMC traffic) are stamped with a hash of tron_amd/csrc instead
of a commit id; bench.py recomputes it and refuses a capture taken on other sources."""
import glob
import hashlib
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_CSRC, "*"))):
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
