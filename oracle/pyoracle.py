"""ctypes front-end of oracle/libtron_oracle.so (TEST INFRASTRUCTURE ONLY).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, and
from nowhere else: the shipped package ``tron_amd`` must never import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtron_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libra_ref.so")


class OracleParams(ctypes.Structure):
    """Mirror of ``oracle_params`` in tron_oracle.c (= the globals of tron.cu:54-87)."""
    _fields_ = [
        ("adjoint", ctypes.c_int), ("golden_angle", ctypes.c_int), ("koosh", ctypes.c_int),
        ("gridos", ctypes.c_float), ("kernwidth", ctypes.c_float), ("data_undersamp", ctypes.c_float),
        ("prof_slide", ctypes.c_int), ("skip_angles", ctypes.c_int),
        ("nc", ctypes.c_int), ("nt", ctypes.c_int), ("nro", ctypes.c_int), ("npe1", ctypes.c_int),
        ("npe2", ctypes.c_int), ("npe1work", ctypes.c_int),
        ("nx", ctypes.c_int), ("ny", ctypes.c_int), ("nz", ctypes.c_int),
        ("nxos", ctypes.c_int), ("nyos", ctypes.c_int), ("nzos", ctypes.c_int),
        ("out_dims", ctypes.c_uint64 * 5),
        ("out_bytes", ctypes.c_uint64),
    ]


def build(force: bool = False) -> None:
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "tron_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "libtron_oracle.so"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        f, i, p = ctypes.c_float, ctypes.c_int, ctypes.c_void_p
        L.oracle_besseli0.restype = f; L.oracle_besseli0.argtypes = [f]
        L.oracle_kernel_shape.restype = f; L.oracle_kernel_shape.argtypes = [f, f]
        L.oracle_gridkernel.restype = f; L.oracle_gridkernel.argtypes = [f, f, f]
        L.oracle_gridkernelhat.restype = f; L.oracle_gridkernelhat.argtypes = [f, f, f]
        L.oracle_modang.restype = f; L.oracle_modang.argtypes = [f]
        L.oracle_grid_angle.restype = f; L.oracle_grid_angle.argtypes = [i, i, i, i]
        L.oracle_degrid_angle.restype = f; L.oracle_degrid_angle.argtypes = [i, i, i, i]
        L.oracle_deapod_weight.restype = f; L.oracle_deapod_weight.argtypes = [ctypes.c_size_t, i, f, f]
        L.oracle_fftshift.restype = None; L.oracle_fftshift.argtypes = [p, p, i, i, i]
        L.oracle_crop.restype = None; L.oracle_crop.argtypes = [p, i, p, i, i]
        L.oracle_pad.restype = None; L.oracle_pad.argtypes = [p, i, p, i, i]
        L.oracle_coilcombinesos.restype = None; L.oracle_coilcombinesos.argtypes = [p, p, i, i]
        L.oracle_deapod.restype = None; L.oracle_deapod.argtypes = [p, i, i, f, f]
        L.oracle_precompensate.restype = None; L.oracle_precompensate.argtypes = [p, i, i, i]
        L.oracle_gridradial2d.restype = None; L.oracle_gridradial2d.argtypes = [p, p, i, i, i, i, f, f, i, i]
        L.oracle_degridradial2d.restype = None; L.oracle_degridradial2d.argtypes = [p, p, i, i, i, i, f, f, i, i]
        L.oracle_fft2.restype = None; L.oracle_fft2.argtypes = [p, p, i, i, i]
        L.oracle_derive_dims.restype = i; L.oracle_derive_dims.argtypes = [ctypes.POINTER(OracleParams), ctypes.POINTER(ctypes.c_uint64)]
        L.oracle_nufft_adj_radial2d.restype = None; L.oracle_nufft_adj_radial2d.argtypes = [ctypes.POINTER(OracleParams), p, p, i]
        L.oracle_nufft_radial2d.restype = None; L.oracle_nufft_radial2d.argtypes = [ctypes.POINTER(OracleParams), p, p]
        L.oracle_recon_radial2d.restype = i; L.oracle_recon_radial2d.argtypes = [ctypes.POINTER(OracleParams), p, p, i, i]
        L.oracle_recon_cgnr.restype = i; L.oracle_recon_cgnr.argtypes = [ctypes.POINTER(OracleParams), p, p, i, i, i, i]
        L.oracle_recon_combine.restype = i; L.oracle_recon_combine.argtypes = [ctypes.POINTER(OracleParams), p, p, i, i, i, i]
        L.oracle_params_size.restype = ctypes.c_size_t
        assert L.oracle_params_size() == ctypes.sizeof(OracleParams)
        _lib = L
    return _lib


def set_threads(n: int) -> None:
    """OpenMP threads of the oracle's parallel loops from here on (launchers such as torchrun export OMP_NUM_THREADS=1,
    which would make a metric-size check take minutes)."""
    lib()
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(max(1, int(n)))
    except OSError:
        pass


def _c64(a):
    a = np.ascontiguousarray(a, dtype=np.complex64)
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ----------------------------------------------------------------------------- scalars

def besseli0(x): return lib().oracle_besseli0(float(x))
def gridkernel(x, W=2.0, sigma=2.0): return lib().oracle_gridkernel(float(x), float(W), float(sigma))
def gridkernelhat(u, W=2.0, sigma=2.0): return lib().oracle_gridkernelhat(float(u), float(W), float(sigma))
def modang(x): return lib().oracle_modang(float(x))
def grid_angle(pe, npe, skip, golden): return lib().oracle_grid_angle(int(pe), int(npe), int(skip), int(golden))
def degrid_angle(pe, npe, skip, golden): return lib().oracle_degrid_angle(int(pe), int(npe), int(skip), int(golden))
def deapod_weight(idx, n, W=2.0, sigma=2.0): return lib().oracle_deapod_weight(int(idx), int(n), float(W), float(sigma))


# ----------------------------------------------------------------------------- stages
# Arrays use the reference's device layouts: channel fastest, i.e. numpy shape
# (..., nchan) C-contiguous complex64.

def gridradial2d(nudata, nxos, W=2.0, gridos=2.0, skip_angles=0, golden=1):
    """nudata: (npe, nro, nchan) complex64 (already density-compensated) -> (nxos, nxos, nchan)."""
    nudata = _c64(nudata)
    npe, nro, nchan = nudata.shape
    out = np.empty((nxos, nxos, nchan), np.complex64)
    lib().oracle_gridradial2d(_ptr(out), _ptr(nudata), nxos, nchan, nro, npe, W, gridos, skip_angles, int(golden))
    return out


def degridradial2d(udata, nro, npe, W=2.0, gridos=2.0, skip_angles=0, golden=1):
    """udata: (n, n, nrep) complex64 -> (npe, nro, nrep)."""
    udata = _c64(udata)
    n, _, nrep = udata.shape
    out = np.empty((npe, nro, nrep), np.complex64)
    lib().oracle_degridradial2d(_ptr(out), _ptr(udata), n, nrep, nro, npe, W, gridos, skip_angles, int(golden))
    return out


def precompensate(nudata):
    nudata = _c64(nudata).copy()
    npe, nro, nchan = nudata.shape
    lib().oracle_precompensate(_ptr(nudata), nchan, nro, npe)
    return nudata


def fft2(a, sign):
    a = _c64(a)
    n, _, nchan = a.shape
    out = np.empty_like(a)
    lib().oracle_fft2(_ptr(out), _ptr(a), n, nchan, int(sign))
    return out


def fftshift(a, direction):
    a = _c64(a)
    n, _, nchan = a.shape
    out = np.empty_like(a)
    lib().oracle_fftshift(_ptr(out), _ptr(a), n, nchan, int(direction))
    return out


def crop(a, ndst):
    a = _c64(a)
    n, _, nchan = a.shape
    out = np.empty((ndst, ndst, nchan), np.complex64)
    lib().oracle_crop(_ptr(out), ndst, _ptr(a), n, nchan)
    return out


def pad(a, ndst):
    a = _c64(a)
    n, _, nchan = a.shape
    out = np.empty((ndst, ndst, nchan), np.complex64)
    lib().oracle_pad(_ptr(out), ndst, _ptr(a), n, nchan)
    return out


def deapod(a, W=2.0, sigma=2.0):
    a = _c64(a).copy()
    n, _, nrep = a.shape
    lib().oracle_deapod(_ptr(a), n, nrep, W, sigma)
    return a


def coilcombinesos(a):
    a = _c64(a)
    n, _, nchan = a.shape
    out = np.empty((n, n), np.complex64)
    lib().oracle_coilcombinesos(_ptr(out), _ptr(a), n, nchan)
    return out


# ----------------------------------------------------------------------------- whole runs

def make_params(in_dims, adjoint, golden=0, gridos=2.0, kernwidth=2.0, data_undersamp=1.0,
                prof_slide=0, skip_angles=0, koosh=0):
    """The getopt defaults of tron.cu:66-87 plus the dim logic of tron.cu:905-961."""
    p = OracleParams()
    p.adjoint, p.golden_angle, p.koosh = int(adjoint), int(golden), int(koosh)
    p.gridos, p.kernwidth, p.data_undersamp = gridos, kernwidth, data_undersamp
    p.prof_slide, p.skip_angles = int(prof_slide), int(skip_angles)
    dims = (ctypes.c_uint64 * 5)(*[int(d) for d in in_dims])
    rc = lib().oracle_derive_dims(ctypes.byref(p), dims)
    if rc != 0:
        raise ValueError("nc must be 1 or even (tron.cu:963)")
    return p


def recon(data, adjoint, zfirst=0, zcount=None, **flags):
    """``tron [-a] ...`` on an in-memory array.

    data: numpy array shaped like the .ra dims in FILE order, i.e. (nc, nt, nro, npe1, npe2)
    for the adjoint / (nc, nt, nx, ny, nz) forward, Fortran-ordered or anything
    np.asfortranarray can convert.  Returns (out, params) with out shaped
    ``params.out_dims`` (Fortran order, like the .ra file tron would write); forward
    output carries nc as leading dimension although the header says 1.
    """
    data = np.asfortranarray(data, dtype=np.complex64)
    assert data.ndim == 5
    p = make_params(data.shape, adjoint, **flags)
    flat_in = data.reshape(-1, order="F")
    nout = p.out_bytes // 8
    flat_out = np.zeros(nout, np.complex64)
    if zcount is None:
        zcount = p.nz
    if not adjoint and not p.koosh:
        zcount = min(zcount, 1)  # h_out is sized for npe2 = 1
    rc = lib().oracle_recon_radial2d(ctypes.byref(p), _ptr(flat_out), _ptr(flat_in), int(zfirst), int(zcount))
    if rc != 0:
        raise ValueError(f"oracle_recon_radial2d failed rc={rc}")
    if adjoint:
        shape = tuple(int(d) for d in p.out_dims)
    else:
        shape = (p.nc,) + tuple(int(d) for d in p.out_dims)[1:]
    return flat_out.reshape(shape, order="F"), p


def recon_combine(data, mode=0, npatch=1, zfirst=0, zcount=None, **flags):
    """``tron -a`` with the coil combination chosen: mode 0 = root-sum-of-squares per repetition (tron.cu:255-268 applied
    to each of the nt repetitions), 1 = Walsh adaptive combine (tron.cu:222-302).  Handles nt > 1."""
    data = np.asfortranarray(data, dtype=np.complex64)
    p = make_params(data.shape, 1, **flags)
    flat_in = data.reshape(-1, order="F")
    flat_out = np.zeros(p.out_bytes // 8, np.complex64)
    if zcount is None:
        zcount = p.nz
    rc = lib().oracle_recon_combine(ctypes.byref(p), _ptr(flat_out), _ptr(flat_in), int(zfirst), int(zcount), int(mode), int(npatch))
    if rc != 0:
        raise ValueError(f"oracle_recon_combine failed rc={rc}")
    return flat_out.reshape(tuple(int(d) for d in p.out_dims), order="F"), p


def recon_cgnr(data, niter, consistent=0, zfirst=0, zcount=None, **flags):
    """``tron -a -i niter ...``: CGNR per slice (the algorithm tron.cu:665-720 cites, see tron_oracle.c), then
    root-sum-of-squares.  Returns (out, params) like ``recon``."""
    data = np.asfortranarray(data, dtype=np.complex64)
    p = make_params(data.shape, 1, **flags)
    flat_in = data.reshape(-1, order="F")
    flat_out = np.zeros(p.out_bytes // 8, np.complex64)
    if zcount is None:
        zcount = p.nz
    rc = lib().oracle_recon_cgnr(ctypes.byref(p), _ptr(flat_out), _ptr(flat_in), int(zfirst), int(zcount), int(niter), int(consistent))
    if rc != 0:
        raise ValueError(f"oracle_recon_cgnr failed rc={rc}")
    return flat_out.reshape(tuple(int(d) for d in p.out_dims), order="F"), p


# ----------------------------------------------------------------------------- oracle/_ref

def have_ref() -> bool:
    return os.path.exists(_REF_SO)


class RaT(ctypes.Structure):
    """ra_t of the reference, src/ra.h:38-48."""
    _fields_ = [("flags", ctypes.c_uint64), ("eltype", ctypes.c_uint64), ("elbyte", ctypes.c_uint64),
                ("size", ctypes.c_uint64), ("ndims", ctypes.c_uint64),
                ("dims", ctypes.POINTER(ctypes.c_uint64)), ("data", ctypes.POINTER(ctypes.c_uint8))]


_ref = None


def ref():
    """The reference's own ra.cu + float16.cu (oracle/_ref/libra_ref.so)."""
    global _ref
    if _ref is None:
        R = ctypes.CDLL(_REF_SO)
        R.ra_read.restype = ctypes.c_int; R.ra_read.argtypes = [ctypes.POINTER(RaT), ctypes.c_char_p]
        R.ra_write.restype = ctypes.c_int; R.ra_write.argtypes = [ctypes.POINTER(RaT), ctypes.c_char_p]
        R.ra_free.restype = None; R.ra_free.argtypes = [ctypes.POINTER(RaT)]
        # float16.cu is compiled as C++, so its symbols are mangled
        R.f2h = getattr(R, "_Z21floatbits_to_halfbitsj"); R.f2h.restype = ctypes.c_uint16; R.f2h.argtypes = [ctypes.c_uint32]
        R.d2h = getattr(R, "_Z22doublebits_to_halfbitsm"); R.d2h.restype = ctypes.c_uint16; R.d2h.argtypes = [ctypes.c_uint64]
        R.h2f = getattr(R, "_Z24float16bits_to_floatbitst"); R.h2f.restype = ctypes.c_uint32; R.h2f.argtypes = [ctypes.c_uint16]
        R.h2d = getattr(R, "_Z25float16bits_to_doublebitst"); R.h2d.restype = ctypes.c_uint64; R.h2d.argtypes = [ctypes.c_uint16]
        _ref = R
    return _ref
