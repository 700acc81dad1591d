"""ctypes front-end of oracle/libirt_nufft.so -- the C++/OpenMP restatement of the reference's CPU comparator
(contrib/irt nufft_init('minmax:kb') + nufft_adj).  TEST INFRASTRUCTURE / reported CPU baseline only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libirt_nufft.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "irt_nufft.cpp")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libirt_nufft.so"], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(_SO)
        i, p, d = ctypes.c_int, ctypes.c_void_p, ctypes.c_double
        L.irt_adjoint.restype = i; L.irt_adjoint.argtypes = [i, i, i, i, p, p, p, i, p, i]
        L.irt_bench_golden.restype = d
        L.irt_bench_golden.argtypes = [i, i, i, i, i, i, ctypes.POINTER(d), ctypes.POINTER(d), ctypes.POINTER(d)]
        _lib = L
    return _lib


def adjoint(om, X, N, J=4, K=None, dcf=None, threads=1):
    """x_c = nufft_adj(dcf .* X_c, nufft_init(om, [N N], [J J], [K K], [N/2 N/2])) for every coil c.
    om: (M, 2) radians; X: (nc, M) complex.  Returns (nc, N, N) complex128 indexed [c, n1, n2]."""
    om = np.ascontiguousarray(om, np.float64)
    X = np.ascontiguousarray(np.atleast_2d(X), np.complex128)
    nc, M = X.shape
    K = 2 * N if K is None else K
    img = np.zeros((nc, N * N), np.complex128)
    w = None if dcf is None else np.ascontiguousarray(dcf, np.float64)
    rc = lib().irt_adjoint(N, J, K, M, om.ctypes.data, X.ctypes.data, None if w is None else w.ctypes.data, nc, img.ctypes.data, threads)
    if rc != 0:
        raise ValueError("irt_adjoint: unsupported J / K")
    return img.reshape(nc, N, N, order="C").transpose(0, 2, 1)      # stored n1 fastest


def bench_golden(N, nro, npe, nc, nslices, threads):
    ti, ta, cs = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    wall = lib().irt_bench_golden(N, nro, npe, nc, nslices, threads, ctypes.byref(ti), ctypes.byref(ta), ctypes.byref(cs))
    return dict(wall_s=wall, init_s=ti.value, adj_s=ta.value, checksum=cs.value)
