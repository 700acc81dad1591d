"""RawArray (.ra) files in numpy -- the Python-side mirror of the reference's
``src/raread.m`` / ``src/rawrite.m`` (MATLAB) and of ``ra_read`` / ``ra_write``
(``src/ra.cu:87-162``).

File layout (``src/ra.h:38-48``, ``src/ra.cu:131-162``): six little-endian u64
``{magic, flags, eltype, elbyte, size, ndims}``, then ``ndims`` u64 dims, then ``size``
bytes of data with the FIRST dimension varying fastest (MATLAB order,
``src/raread.m:25-57``).  ``eltype`` 0 user, 1 int, 2 uint, 3 float, 4 complex
(``src/ra.h:63-72``); a complex element is ``elbyte`` bytes = (re, im) halves.

This module is host-side plumbing for tests, fixtures and the bench; the C
implementation the ``tron`` CLI uses lives in ``tron_amd/csrc/rawarray.cpp``.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

RA_MAGIC = 0x7961727261776172  # "rawarray" little-endian, src/ra.h:51
RA_FLAG_BIG_ENDIAN = 1 << 0    # src/ra.h:55
RA_FLAG_COMPRESSED = 1 << 1    # src/ra.h:56

RA_TYPE_USER, RA_TYPE_INT, RA_TYPE_UINT, RA_TYPE_FLOAT, RA_TYPE_COMPLEX = range(5)

_DTYPES = {
    (RA_TYPE_INT, 1): np.int8, (RA_TYPE_INT, 2): np.int16, (RA_TYPE_INT, 4): np.int32, (RA_TYPE_INT, 8): np.int64,
    (RA_TYPE_UINT, 1): np.uint8, (RA_TYPE_UINT, 2): np.uint16, (RA_TYPE_UINT, 4): np.uint32, (RA_TYPE_UINT, 8): np.uint64,
    (RA_TYPE_FLOAT, 2): np.float16, (RA_TYPE_FLOAT, 4): np.float32, (RA_TYPE_FLOAT, 8): np.float64,
    (RA_TYPE_COMPLEX, 8): np.complex64, (RA_TYPE_COMPLEX, 16): np.complex128,
}


@dataclass
class RaHeader:
    flags: int
    eltype: int
    elbyte: int
    size: int
    dims: tuple

    @property
    def ndims(self) -> int:
        return len(self.dims)

    @property
    def nbytes_header(self) -> int:
        return 8 * (6 + len(self.dims))


def read_header(path) -> RaHeader:
    with open(path, "rb") as f:
        return _read_header(f)


def _read_header(f) -> RaHeader:
    raw = f.read(48)
    if len(raw) != 48:
        raise ValueError("truncated RawArray header")
    magic, flags, eltype, elbyte, size, ndims = struct.unpack("<6Q", raw)
    if magic != RA_MAGIC:
        raise ValueError("Invalid RA file.")  # same message as src/ra.cu:60
    if ndims > 64:
        raise ValueError(f"implausible ndims {ndims}")
    raw = f.read(8 * ndims)
    if len(raw) != 8 * ndims:
        raise ValueError("truncated RawArray dims")
    dims = struct.unpack(f"<{ndims}Q", raw)
    return RaHeader(flags, eltype, elbyte, size, tuple(dims))


def read(path, with_header: bool = False):
    """Read a .ra file.  Returns an array whose ``shape`` is the file's dims in file
    order and whose memory order is Fortran (first dim fastest), exactly like
    ``raread.m``.  Complex-half files (eltype 4, elbyte 4) come back as an
    ``(2, *dims)`` float16 array (numpy has no complex32), like ``raread.m`` does for
    every complex type."""
    with open(path, "rb") as f:
        h = _read_header(f)
        if h.flags & (RA_FLAG_BIG_ENDIAN | RA_FLAG_COMPRESSED):
            raise NotImplementedError("big-endian / compressed RawArray files are not implemented (nor in the reference)")
        payload = f.read(h.size)
        if len(payload) != h.size:
            raise ValueError(f"Read {len(payload)} B instead of {h.size} B.")
    nel = int(np.prod(h.dims, dtype=np.uint64)) if h.dims else 1
    if h.eltype == RA_TYPE_COMPLEX and h.elbyte == 4:
        arr = np.frombuffer(payload, dtype=np.float16, count=2 * nel).reshape((2,) + h.dims, order="F")
    elif (h.eltype, h.elbyte) in _DTYPES:
        arr = np.frombuffer(payload, dtype=_DTYPES[(h.eltype, h.elbyte)], count=nel).reshape(h.dims, order="F")
    else:
        arr = np.frombuffer(payload, dtype=np.uint8)
    return (arr, h) if with_header else arr


def read_spokes(path, first_spoke: int, nspokes: int):
    """Reads only spokes [first_spoke, first_spoke+nspokes) of a 5-D k-space file [nc, nt, nro, npe1, npe2] (a spoke =
    nc*nt*nro consecutive elements): one seek + one read of exactly those bytes.  Returns (flat array, header,
    bytes_read); complex64 files give complex64, complex-half files float16 (re, im) pairs."""
    with open(path, "rb") as f:
        h = _read_header(f)
        if h.ndims != 5 or h.eltype != RA_TYPE_COMPLEX or h.elbyte not in (4, 8):
            raise ValueError("read_spokes wants a 5-D complex64 / complex-half k-space file")
        spoke_bytes = h.dims[0] * h.dims[1] * h.dims[2] * h.elbyte
        total = h.dims[3] * h.dims[4]
        if first_spoke < 0 or nspokes < 0 or first_spoke + nspokes > total:
            raise ValueError(f"spokes [{first_spoke}, {first_spoke + nspokes}) outside the file's {total}")
        f.seek(h.nbytes_header + first_spoke * spoke_bytes)
        payload = f.read(nspokes * spoke_bytes)
        if len(payload) != nspokes * spoke_bytes:
            raise ValueError(f"Read {len(payload)} B instead of {nspokes * spoke_bytes} B.")
    arr = np.frombuffer(payload, dtype=np.complex64 if h.elbyte == 8 else np.float16)
    return arr, h, len(payload)


def _classify(arr: np.ndarray):
    k = arr.dtype.kind
    if k == "c":
        return RA_TYPE_COMPLEX, arr.dtype.itemsize
    if k == "f":
        return RA_TYPE_FLOAT, arr.dtype.itemsize
    if k == "i":
        return RA_TYPE_INT, arr.dtype.itemsize
    if k in "ub":
        return RA_TYPE_UINT, arr.dtype.itemsize
    return RA_TYPE_USER, arr.dtype.itemsize


def write(path, arr: np.ndarray, ntrailing: int = 0, complex_half: bool = False) -> None:
    """Write ``arr`` with dims = ``arr.shape`` (+ ``ntrailing`` singleton dims, as
    ``rawrite.m:26-28,56-59``), first dimension fastest.  ``complex_half=True`` takes a
    ``(2, *dims)`` float16 array and writes eltype 4 / elbyte 4."""
    arr = np.asarray(arr)
    if complex_half:
        if arr.dtype != np.float16 or arr.shape[0] != 2:
            raise ValueError("complex_half wants a (2, ...) float16 array")
        eltype, elbyte, dims = RA_TYPE_COMPLEX, 4, arr.shape[1:]
    else:
        eltype, elbyte = _classify(arr)
        dims = arr.shape
    dims = tuple(int(d) for d in dims) + (1,) * ntrailing
    payload = np.asfortranarray(arr).tobytes(order="F")
    with open(path, "wb") as f:
        f.write(struct.pack("<6Q", RA_MAGIC, 0, eltype, elbyte, len(payload), len(dims)))
        f.write(struct.pack(f"<{len(dims)}Q", *dims))
        f.write(payload)
