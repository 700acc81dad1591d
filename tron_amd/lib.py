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
        ("pin_host", ctypes.c_int), ("cgnr_consistent", ctypes.c_int),
        ("coil_combine", ctypes.c_int), ("walsh_patch", ctypes.c_int),
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
    "tron_recon_radial2d", "tron_recon_radial2d_range", "tron_recon_radial2d_block", "tron_recon_radial2d_multi", "tron_nufft_adj_radial2d", "tron_cgnr_radial2d", "tron_nufft_radial2d",
    "tron_precompensate", "tron_gridradial2d", "tron_degridradial2d", "tron_plan_sync",
    "tron_plan_timing", "tron_plan_timing_get", "tron_plan_timing_reset", "tron_plan_retarget", "tron_plan_retarget_times", "tron_plan_grid_kernel_name", "tron_plan_degrid_kernel_name", "tron_plan_create_times", "tron_plan_shader_clock",
    "tron_host_trig_table", "tron_host_band_table", "tron_host_deapod_table", "tron_host_numa_cpulist",
    "tron_device_count", "tron_device_pci_bus_id", "tron_device_malloc", "tron_device_free", "tron_memcpy_h2d", "tron_memcpy_d2h",
    "tron_last_error", "tron_version",
    "ra_read", "ra_write", "ra_free", "ra_query", "ra_reshape", "ra_convert", "ra_squash", "ra_diff", "ra_read_header", "ra_data_offset", "ra_write_header", "ra_read_range", "ra_write_range",
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

    def sig(name, restype, argtypes):
        # a symbol missing from an OLDER build (tools/ab.sh A/B runs) is simply not bound; build() checks EXPORTS
        try:
            fn = getattr(L, name)
        except AttributeError:
            return
        fn.restype, fn.argtypes = restype, argtypes

    sig("tron_config_default", None, [pc])
    sig("tron_derive_dims", i, [pc, ctypes.POINTER(ctypes.c_uint64), pd])
    sig("tron_plan_create", i, [ctypes.POINTER(p), pc, pd])
    sig("tron_plan_destroy", i, [p])
    sig("tron_recon_radial2d", i, [p, p, p])
    sig("tron_recon_radial2d_range", i, [p, p, p, i, i])
    sig("tron_recon_radial2d_block", i, [p, p, p, i, i])
    sig("tron_recon_radial2d_multi", i, [pc, pd, ctypes.POINTER(i), i, p, p])
    sig("tron_nufft_adj_radial2d", i, [p, p, p, i, i, i])
    sig("tron_cgnr_radial2d", i, [p, p, p, i, i, i])
    sig("tron_nufft_radial2d", i, [p, p, p, i])
    sig("tron_precompensate", i, [p, p])
    sig("tron_gridradial2d", i, [p, p, p, i])
    sig("tron_degridradial2d", i, [p, p, p])
    sig("tron_plan_sync", i, [p])
    sig("tron_plan_retarget", i, [p, i])
    sig("tron_plan_retarget_times", i, [p, ctypes.POINTER(ctypes.c_double)])
    sig("tron_plan_create_times", i, [p, ctypes.POINTER(ctypes.c_double)])
    sig("tron_plan_shader_clock", i, [p, ctypes.POINTER(ctypes.c_double)])
    sig("tron_plan_grid_kernel_name", ctypes.c_char_p, [p])
    sig("tron_plan_degrid_kernel_name", ctypes.c_char_p, [p])
    sig("tron_host_numa_cpulist", i, [ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(i), i])
    sig("tron_plan_timing", i, [p, i])
    sig("tron_plan_timing_get", i, [p, i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)])
    sig("tron_plan_timing_reset", i, [p])
    sig("tron_host_trig_table", i, [pc, pd, p, sz])
    sig("tron_host_band_table", i, [i, f, p])
    sig("tron_host_deapod_table", i, [i, f, f, p])
    sig("tron_device_count", i, [ctypes.POINTER(i)])
    sig("tron_device_pci_bus_id", i, [i, ctypes.c_char_p, i])
    sig("tron_device_malloc", i, [ctypes.POINTER(p), sz])
    sig("tron_device_free", i, [p])
    sig("tron_memcpy_h2d", i, [p, p, sz])
    sig("tron_memcpy_d2h", i, [p, p, sz])
    sig("tron_last_error", ctypes.c_char_p, [])
    sig("tron_version", ctypes.c_char_p, [])
    sig("ra_float_to_half_bits", ctypes.c_uint16, [ctypes.c_uint32])
    sig("ra_half_to_float_bits", ctypes.c_uint32, [ctypes.c_uint16])
    sig("ra_double_to_half_bits", ctypes.c_uint16, [ctypes.c_uint64])
    sig("ra_half_to_double_bits", ctypes.c_uint64, [ctypes.c_uint16])
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


def device_pci_bus_id(device: int) -> str:
    buf = ctypes.create_string_buffer(64)
    check(load().tron_device_pci_bus_id(int(device), buf, 64))
    return buf.value.decode()


def numa_cpulist(sysroot: str, bus_id: str, max_cpus: int = 4096):
    """CPUs of the NUMA node a PCI function hangs off, from a sysfs tree (= tron_host_numa_cpulist); [] when the node is unknown."""
    cpus = (ctypes.c_int * max_cpus)()
    n = load().tron_host_numa_cpulist(sysroot.encode(), bus_id.encode(), cpus, max_cpus)
    if n < 0:
        raise ValueError(f"malformed cpulist under {sysroot} for {bus_id}")
    return [cpus[k] for k in range(n)]


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

    def write(self, a: np.ndarray, offset_bytes: int = 0):
        """Copies a host array into the buffer at a byte offset (a stream assembled block by block)."""
        a = np.ascontiguousarray(a)
        assert 0 <= offset_bytes and offset_bytes + a.nbytes <= self.nbytes
        check(load().tron_memcpy_h2d(ctypes.c_void_p(self.ptr.value + int(offset_bytes)), a.ctypes.data_as(ctypes.c_void_p), a.nbytes))

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
        # the C side can only validate against its own dims: a short or strided array would be read past its end
        elem = 4 if self.cfg.input_half else 8
        if not (isinstance(flat_in, np.ndarray) and (flat_in.flags.c_contiguous or flat_in.flags.f_contiguous)):
            raise ValueError("Plan.recon: input must be a contiguous numpy array")
        if flat_in.nbytes < d.in_elems * elem:
            raise ValueError(f"Plan.recon: input holds {flat_in.nbytes} bytes, the plan's dims need {d.in_elems * elem}")
        if self.cfg.input_half and flat_in.dtype != np.float16:
            raise ValueError("Plan.recon: input_half plans take float16 (re, im) pairs")
        if not self.cfg.input_half and flat_in.dtype not in (np.dtype(np.complex64), np.dtype(np.float32)):
            raise ValueError(f"Plan.recon: expected complex64 input, got {flat_in.dtype}")
        if not (out.flags.c_contiguous or out.flags.f_contiguous) or out.dtype != np.complex64 or out.nbytes < d.out_bytes:
            raise ValueError(f"Plan.recon: output must be contiguous complex64 of at least {d.out_bytes} bytes")
        check(load().tron_recon_radial2d_range(self._h, out.ctypes.data_as(ctypes.c_void_p),
                                               flat_in.ctypes.data_as(ctypes.c_void_p), int(zfirst), int(zcount)))
        return out

    def recon_block(self, block_in: np.ndarray, zfirst: int, zcount: int) -> np.ndarray:
        """Adjoint of slices [zfirst, zfirst+zcount) from a buffer that holds ONLY their spokes (block_in[0] = spoke
        zfirst*prof_slide of the stream); returns the zcount images (= tron_recon_radial2d_block)."""
        d = self.dims
        elem = 4 if self.cfg.input_half else 8
        need = ((zcount - 1) * d.prof_slide + d.npe1work) * d.nro * d.nc * d.nt * elem if zcount > 0 else 0
        if not (block_in.flags.c_contiguous or block_in.flags.f_contiguous) or block_in.nbytes < need:
            raise ValueError(f"Plan.recon_block: input must be contiguous and hold {need} bytes, has {block_in.nbytes}")
        out = np.zeros(zcount * d.nt * d.nx * d.ny, np.complex64)
        check(load().tron_recon_radial2d_block(self._h, out.ctypes.data_as(ctypes.c_void_p),
                                               block_in.ctypes.data_as(ctypes.c_void_p), int(zfirst), int(zcount)))
        return out

    def adjoint_device(self, d_out, d_in, zfirst, zcount, combine=1):
        check(load().tron_nufft_adj_radial2d(self._h, d_out, d_in, int(zfirst), int(zcount), int(combine)))

    def cgnr_device(self, d_out, d_in, zfirst, zcount, combine=1):
        check(load().tron_cgnr_radial2d(self._h, d_out, d_in, int(zfirst), int(zcount), int(combine)))

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

    def grid_kernel_name(self) -> str:
        """Which gridding kernels this plan launches (bench.py's roofline line, the traffic captures)."""
        return load().tron_plan_grid_kernel_name(self._h).decode()

    def degrid_kernel_name(self) -> str:
        """Which degridding kernel the most recent forward launch of this plan ran ("" before the first)."""
        return load().tron_plan_degrid_kernel_name(self._h).decode()

    def create_times(self) -> dict:
        """Seconds tron_plan_create spent: total, HIP runtime + code objects, tables, of which the gridding kernels' run tables, work buffers."""
        t = (ctypes.c_double * 5)()
        check(load().tron_plan_create_times(self._h, t))
        return dict(total=t[0], runtime=t[1], tables=t[2], run_tables=t[3], work_buffers=t[4])

    def shader_clock_mhz(self) -> float:
        """Shader clock (MHz) right behind what the plan has queued (s_memtime / s_memrealtime of a short spin kernel); synchronises."""
        mhz = ctypes.c_double(0.0)
        check(load().tron_plan_shader_clock(self._h, ctypes.byref(mhz)))
        return mhz.value

    def retarget(self, skip_angles: int):
        """Later calls use spoke angles starting at index `skip_angles` (= tron_plan_retarget: the new tables are built on the
        device beside the work already queued; results equal those of a plan created with this skip_angles)."""
        check(load().tron_plan_retarget(self._h, int(skip_angles)))
        self.cfg.skip_angles = int(skip_angles)

    def retarget_times(self) -> dict:
        """Host seconds of the last retarget() call, of which the (cos, sin) table."""
        t = (ctypes.c_double * 2)()
        check(load().tron_plan_retarget_times(self._h, t))
        return dict(call=t[0], trig=t[1])

    def timing(self, enable=True):
        check(load().tron_plan_timing(self._h, int(enable)))

    def timing_reset(self):
        check(load().tron_plan_timing_reset(self._h))

    def timing_get(self, stage):
        ms, n = ctypes.c_double(0), ctypes.c_uint64(0)
        check(load().tron_plan_timing_get(self._h, int(stage), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


def recon_multi(data: np.ndarray, adjoint: bool, devices=None, n_devices=0, **flags):
    """``recon`` over several GPUs inside this process (= tron_recon_radial2d_multi): one worker thread + plan per device.
    input_half=1 takes a float16 array shaped (2, nc, nt, nro, npe1, npe2) like ``recon``."""
    half = flags.pop("input_half", 0)
    if half:
        if data.dtype != np.float16 or data.shape[0] != 2:
            raise ValueError("input_half=1 needs a float16 array whose first axis is (re, im)")
        shape = data.shape[1:]
        flat = np.asfortranarray(data).reshape(-1, order="F")
    else:
        data = np.asfortranarray(data, dtype=np.complex64)
        shape = data.shape
        flat = data.reshape(-1, order="F")
    cfg = default_config(adjoint=int(adjoint), input_half=int(half), **flags)
    dims = derive_dims(cfg, shape)
    out = np.zeros(dims.out_bytes // 8, np.complex64)
    devs = None
    if devices is not None:
        n_devices = len(devices)
        devs = (ctypes.c_int * n_devices)(*[int(x) for x in devices])
    check(load().tron_recon_radial2d_multi(ctypes.byref(cfg), ctypes.byref(dims), devs, int(n_devices),
                                           out.ctypes.data_as(ctypes.c_void_p), flat.ctypes.data_as(ctypes.c_void_p)))
    oshape = tuple(int(x) for x in dims.out_dims) if adjoint else (dims.nc,) + tuple(int(x) for x in dims.out_dims)[1:]
    return out.reshape(oshape, order="F"), dims


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
