"""Environment probe (tooling): does our HIP .so coexist with PyTorch in one process?

Usage: python tools/probe/probe.py [torch_first|lib_first|notorch]
"""
import ctypes
import os
import sys

mode = sys.argv[1] if len(sys.argv) > 1 else "torch_first"
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libprobe.so")


def loaded_libs():
    out = []
    with open("/proc/self/maps") as f:
        for line in f:
            p = line.strip().split()[-1]
            if any(k in p for k in ("amdhip64", "rocfft", "hsa-runtime")) and p not in out:
                out.append(p)
    return out


if mode == "notorch":
    lib = ctypes.CDLL(so)
    lib.probe_run.argtypes = [ctypes.c_void_p, ctypes.c_int]
    print("rc", lib.probe_run(None, 1))
    print(loaded_libs())
    sys.exit(0)

if mode == "torch_first":
    import torch
    torch.cuda.init()
    lib = ctypes.CDLL(so)
else:
    lib = ctypes.CDLL(so)
    import torch
    torch.cuda.init()

lib.probe_run.argtypes = [ctypes.c_void_p, ctypes.c_int]
t = torch.ones(1024, device="cuda")
torch.cuda.synchronize()
rc = lib.probe_run(ctypes.c_void_p(t.data_ptr()), 1)
torch.cuda.synchronize()
print("rc", rc, "t[0:4] after lib memset:", t[:4].tolist(), "t[4:6]", t[4:6].tolist())
print(loaded_libs())
