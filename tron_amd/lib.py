"""ctypes binding of ``libtronhip.so`` (the C ABI declared in ``include/tron_hip.h``).

There is no CPU fallback: if the shared library is missing or the HIP runtime cannot find a
device, calls raise.  When PyTorch shares the process, import torch and initialise its HIP
context BEFORE loading this module's library (both then use the one HIP runtime torch loaded).
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libtronhip.so")

TRON_OK, TRON_ERR_INVALID, TRON_ERR_UNSUPPORTED, TRON_ERR_HIP, TRON_ERR_FFT, TRON_ERR_NOMEM = range(6)
KB_EXACT, KB_FAST = 0, 1
STAGE_GRID, STAGE_FFT, STAGE_POST, STAGE_PRE, STAGE_DEGRID = range(5)


class TronError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"tronhip error {code}: {msg}")
        self.code = code


class Config(ctypes.Structure):
    """``tron_config``: the getopt-settable globals of src/tron.cu:58-87."""
    _fields_ = [
        ("adjoint", ctypes.c_int), ("golden_angle", ctypes.c_int), ("koosh", ctypes.c_int), ("verbose", ctypes.c_int),
        ("gridos", ctypes.c_float), ("kernwidth", ctypes.c_float), ("data_undersamp", ctypes.c_float),
        ("prof_slide", ctypes.c_int), ("skip_angles", ctypes.c_int), ("niter", ctypes.c_int),
        ("blocks", ctypes.c_int), ("threads", ctypes.c_int), ("device", ctypes.c_int),
        ("kb_mode", ctypes.c_int), ("input_half", ctypes.c_int), ("chunk_slices", ctypes.c_int),
    ]


class Dims(ctypes.Structure):
    """``tron_dims``: what main() derives (src/tron.cu:905-961)."""
    _fields_ = [
        ("nc", ctypes.c_int), ("nt", ctypes.c_int),
        ("nro", ctypes.c_int), ("npe1", ctypes.c_int), ("npe2", ctypes.c_int), ("npe1work", ctypes.c_int),
        ("nx", ctypes.c_int), ("ny", ctypes.c_int), ("nz", ctypes.c_int),
        ("nxos", ctypes.c_int), ("nyos", ctypes.c_int), ("nzos", ctypes.c_int),
        ("prof_slide", ctypes.c_int),
        ("out_dims", ctypes.c_uint64 * 5), ("out_bytes", ctypes.c_uint64), ("in_elems", ctypes.c_uint64),
    ]


# every symbol include/tron_hip.h and include/rawarray.h declare
EXPORTS = [
    "tron_config_default", "tron_derive_dims", "tron_plan_create", "tron_plan_destroy",
    "tron_recon_radial2d", "tron_recon_radial2d_range", "tron_nufft_adj_radial2d", "tron_nufft_radial2d",
    "tron_precompensate", "tron_gridradial2d", "tron_degridradial2d", "tron_plan_sync",
    "tron_plan_timing", "tron_plan_timing_get", "tron_plan_timing_reset",
    "tron_host_trig_table", "tron_host_band_table", "tron_host_deapod_table",
    "tron_device_count", "tron_device_malloc", "tron_device_free", "tron_memcpy_h2d", "tron_memcpy_d2h",
    "tron_last_error", "tron_version",
    "ra_read", "ra_write", "ra_free", "ra_query", "ra_reshape", "ra_convert", "ra_squash", "ra_diff", "ra_read_header",
    "ra_float_to_half_bits", "ra_half_to_float_bits", "ra_double_to_half_bits", "ra_half_to_double_bits",
]

_lib = None


def load():
    """Loads libtronhip.so (raises OSError with a build hint if it is not there)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); there is no CPU fallback")
    L = ctypes.CDLL(LIB_PATH)
    i, f, p, sz = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
    pc, pd = ctypes.POINTER(Config), ctypes.POINTER(Dims)
    L.tron_config_default.restype = None; L.tron_config_default.argtypes = [pc]
    L.tron_derive_dims.restype = i; L.tron_derive_dims.argtypes = [pc, ctypes.POINTER(ctypes.c_uint64), pd]
    L.tron_plan_create.restype = i; L.tron_plan_create.argtypes = [ctypes.POINTER(p), pc, pd]
    L.tron_plan_destroy.restype = i; L.tron_plan_destroy.argtypes = [p]
    L.tron_recon_radial2d.restype = i; L.tron_recon_radial2d.argtypes = [p, p, p]
    L.tron_recon_radial2d_range.restype = i; L.tron_recon_radial2d_range.argtypes = [p, p, p, i, i]
    L.tron_nufft_adj_radial2d.restype = i; L.tron_nufft_adj_radial2d.argtypes = [p, p, p, i, i, i]
    L.tron_nufft_radial2d.restype = i; L.tron_nufft_radial2d.argtypes = [p, p, p, i]
    L.tron_precompensate.restype = i; L.tron_precompensate.argtypes = [p, p]
    L.tron_gridradial2d.restype = i; L.tron_gridradial2d.argtypes = [p, p, p, i]
    L.tron_degridradial2d.restype = i; L.tron_degridradial2d.argtypes = [p, p, p]
    L.tron_plan_sync.restype = i; L.tron_plan_sync.argtypes = [p]
    L.tron_plan_timing.restype = i; L.tron_plan_timing.argtypes = [p, i]
    L.tron_plan_timing_get.restype = i; L.tron_plan_timing_get.argtypes = [p, i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    L.tron_plan_timing_reset.restype = i; L.tron_plan_timing_reset.argtypes = [p]
    L.tron_host_trig_table.restype = i; L.tron_host_trig_table.argtypes = [pc, pd, p, sz]
    L.tron_host_band_table.restype = i; L.tron_host_band_table.argtypes = [i, f, p]
    L.tron_host_deapod_table.restype = i; L.tron_host_deapod_table.argtypes = [i, f, f, p]
    L.tron_device_count.restype = i; L.tron_device_count.argtypes = [ctypes.POINTER(i)]
    L.tron_device_malloc.restype = i; L.tron_device_malloc.argtypes = [ctypes.POINTER(p), sz]
    L.tron_device_free.restype = i; L.tron_device_free.argtypes = [p]
    L.tron_memcpy_h2d.restype = i; L.tron_memcpy_h2d.argtypes = [p, p, sz]
    L.tron_memcpy_d2h.restype = i; L.tron_memcpy_d2h.argtypes = [p, p, sz]
    L.tron_last_error.restype = ctypes.c_char_p; L.tron_last_error.argtypes = []
    L.tron_version.restype = ctypes.c_char_p; L.tron_version.argtypes = []
    L.ra_float_to_half_bits.restype = ctypes.c_uint16; L.ra_float_to_half_bits.argtypes = [ctypes.c_uint32]
    L.ra_half_to_float_bits.restype = ctypes.c_uint32; L.ra_half_to_float_bits.argtypes = [ctypes.c_uint16]
    L.ra_double_to_half_bits.restype = ctypes.c_uint16; L.ra_double_to_half_bits.argtypes = [ctypes.c_uint64]
    L.ra_half_to_double_bits.restype = ctypes.c_uint64; L.ra_half_to_double_bits.argtypes = [ctypes.c_uint16]
    _lib = L
    return L


def check(rc):
    if rc != TRON_OK:
        raise TronError(rc, load().tron_last_error().decode(errors="replace"))


def default_config(**kw) -> Config:
    cfg = Config()
    load().tron_config_default(ctypes.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def derive_dims(cfg: Config, in_dims) -> Dims:
    d = Dims()
    arr = (ctypes.c_uint64 * 5)(*[int(x) for x in in_dims])
    check(load().tron_derive_dims(ctypes.byref(cfg), arr, ctypes.byref(d)))
    return d


def device_count() -> int:
    n = ctypes.c_int(0)
    rc = load().tron_device_count(ctypes.byref(n))
    return n.value if rc == TRON_OK else 0


class DeviceBuffer:
    """A device allocation owned through the C ABI (no torch needed)."""

    def __init__(self, nbytes: int):
        self.ptr = ctypes.c_void_p()
        self.nbytes = int(nbytes)
        check(load().tron_device_malloc(ctypes.byref(self.ptr), self.nbytes))

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(load().tron_memcpy_h2d(b.ptr, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
        return b

    def to_numpy(self, dtype, count) -> np.ndarray:
        out = np.empty(count, dtype)
        assert out.nbytes <= self.nbytes
        check(load().tron_memcpy_d2h(out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            load().tron_device_free(self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Plan:
    """``tron_plan``: = tron_init()/tron_shutdown() of the reference (src/tron.cu:579-620)."""

    def __init__(self, cfg: Config, dims: Dims):
        self.cfg, self.dims = cfg, dims
        self._h = ctypes.c_void_p()
        check(load().tron_plan_create(ctypes.byref(self._h), ctypes.byref(cfg), ctypes.byref(dims)))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            load().tron_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # host buffers in, host buffers out (= recon_radial2d, src/tron.cu:726-786)
    def recon(self, flat_in: np.ndarray, zfirst=0, zcount=None, out=None) -> np.ndarray:
        d = self.dims
        if out is None:
            out = np.zeros(d.out_bytes // 8, np.complex64)
        if zcount is None:
            zcount = d.nz
        check(load().tron_recon_radial2d_range(self._h, out.ctypes.data_as(ctypes.c_void_p),
                                               flat_in.ctypes.data_as(ctypes.c_void_p), int(zfirst), int(zcount)))
        return out

    def adjoint_device(self, d_out, d_in, zfirst, zcount, combine=1):
        check(load().tron_nufft_adj_radial2d(self._h, d_out, d_in, int(zfirst), int(zcount), int(combine)))

    def forward_device(self, d_out, d_in, nimg):
        check(load().tron_nufft_radial2d(self._h, d_out, d_in, int(nimg)))

    def precompensate_device(self, d_nudata):
        check(load().tron_precompensate(self._h, d_nudata))

    def grid_device(self, d_udata, d_nudata, skip):
        check(load().tron_gridradial2d(self._h, d_udata, d_nudata, int(skip)))

    def degrid_device(self, d_nudata, d_udata):
        check(load().tron_degridradial2d(self._h, d_nudata, d_udata))

    def sync(self):
        check(load().tron_plan_sync(self._h))

    def timing(self, enable=True):
        check(load().tron_plan_timing(self._h, int(enable)))

    def timing_reset(self):
        check(load().tron_plan_timing_reset(self._h))

    def timing_get(self, stage):
        ms, n = ctypes.c_double(0), ctypes.c_uint64(0)
        check(load().tron_plan_timing_get(self._h, int(stage), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


def recon(data: np.ndarray, adjoint: bool, **flags):
    """``tron [-a] ...`` on an in-memory array shaped like the input .ra dims in file order
    ((nc, nt, nro, npe1, npe2) adjoint / (nc, nt, nx, ny, nz) forward; Fortran order).
    Returns (out, dims) with ``out`` shaped like the output .ra dims (forward output carries
    nc as its leading dimension although the reference's header says 1)."""
    half = flags.pop("input_half", 0)
    if half:
        assert data.dtype == np.float16 and data.shape[0] == 2
        shape = data.shape[1:]
        flat = np.asfortranarray(data).reshape(-1, order="F")
    else:
        data = np.asfortranarray(data, dtype=np.complex64)
        shape = data.shape
        flat = data.reshape(-1, order="F")
    cfg = default_config(adjoint=int(adjoint), input_half=int(half), **flags)
    dims = derive_dims(cfg, shape)
    with Plan(cfg, dims) as plan:
        out = plan.recon(flat)
    if adjoint:
        oshape = tuple(int(x) for x in dims.out_dims)
    else:
        oshape = (dims.nc,) + tuple(int(x) for x in dims.out_dims)[1:]
    return out.reshape(oshape, order="F"), dims
