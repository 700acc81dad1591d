"""CPU-only tests of the product's host side: the C-ABI library loads and exports every declared
symbol, the dimension logic and the host tables agree with the oracle BIT FOR BIT, and the
RawArray / binary16 code agrees with the reference's own src/ra.cu and src/float16.cu
(compiled unmodified into oracle/_ref)."""
import ctypes
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import ref_numpy
import synth
from tron_amd import lib, ra

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    L = lib.load()
    declared = set()
    for hdr in ("tron_hip.h", "rawarray.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b((?:tron|ra)_[a-z0-9_]+)\s*\(", text))
    declared -= {"ra_t", "ra_type"}
    assert declared, "no declarations parsed"
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.tron_version()


def test_product_does_not_touch_the_oracle():
    """The shipped package must never import, link or load anything under oracle/."""
    needles = ("pyoracle", "libtron_oracle", "tron_oracle", "oracle/", "from oracle", "import oracle", "_ref/")
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tron_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                for n in needles:
                    assert n not in text, (os.path.join(dirpath, f), n)
    out = subprocess.run(["readelf", "-d", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle_" not in syms


DIM_CASES = [
    ((6, 1, 512, 20271, 1), dict(adjoint=1, golden=1, data_undersamp=0.4, prof_slide=21)),   # RUNME3:10
    ((1, 1, 512, 402 * 4, 1), dict(adjoint=1, golden=1, data_undersamp=0.7852, prof_slide=402)),
    ((8, 1, 512, 804 * 3, 1), dict(adjoint=1, golden=1, data_undersamp=1.5704, prof_slide=804)),
    ((1, 1, 512, 512, 1), dict(adjoint=1)),                                                  # RUNME3:6
    ((2, 1, 64, 100, 3), dict(adjoint=1, koosh=1)),
    ((1, 1, 256, 256, 1), dict(adjoint=0)),                                                  # RUNME1:5
    ((2, 1, 100, 100, 5), dict(adjoint=0, gridos=1.25, data_undersamp=0.3)),
    ((2, 1, 64, 64, 4), dict(adjoint=0, koosh=1)),
    ((4, 1, 96, 77, 1), dict(adjoint=1, gridos=1.5, data_undersamp=0.33, prof_slide=5)),
]


@pytest.mark.parametrize("in_dims,flags", DIM_CASES)
def test_derive_dims_matches_oracle(oracle, in_dims, flags):
    p = oracle.make_params(in_dims, **flags)
    kw = {("golden_angle" if k == "golden" else k): v for k, v in flags.items()}
    d = lib.derive_dims(lib.default_config(**kw), in_dims)
    for f in ("nc", "nt", "nro", "npe1", "npe2", "npe1work", "nx", "ny", "nz", "nxos", "nyos", "nzos", "prof_slide"):
        assert getattr(d, f) == getattr(p, f), f
    assert tuple(d.out_dims) == tuple(p.out_dims)
    assert d.out_bytes == p.out_bytes


def test_defaults_are_the_references():
    cfg = lib.default_config()
    assert (cfg.gridos, cfg.kernwidth, cfg.data_undersamp) == (2.0, 2.0, 1.0)       # tron.cu:67-69
    assert (cfg.prof_slide, cfg.skip_angles, cfg.niter, cfg.adjoint, cfg.golden_angle) == (0, 0, 0, 0, 0)
    assert (cfg.blocks, cfg.threads) == (4096, 128)                                  # tron.cu:58-59


@pytest.mark.parametrize("adjoint,golden", [(1, 1), (1, 0), (0, 1), (0, 0)])
def test_trig_table_bitexact(oracle, adjoint, golden):
    if adjoint:
        in_dims, kw = (2, 1, 64, 300, 1), dict(data_undersamp=0.5, prof_slide=9, skip_angles=11)
    else:
        in_dims, kw = (2, 1, 32, 32, 1), dict(data_undersamp=0.75, skip_angles=11)
    cfg = lib.default_config(adjoint=adjoint, golden_angle=golden, **kw)
    d = lib.derive_dims(cfg, in_dims)
    n = (d.nz - 1) * d.prof_slide + d.npe1work if (adjoint and golden) else d.npe1work
    tab = np.zeros((n, 2), np.float32)
    lib.check(lib.load().tron_host_trig_table(ctypes.byref(cfg), ctypes.byref(d), tab.ctypes.data_as(ctypes.c_void_p), n))
    for i in list(range(0, n, 7)) + [n - 1]:
        if adjoint:
            # slice z, spoke pe uses angle index pe + skip + z*slide (tron.cu:629-630)
            t = oracle.grid_angle(i, d.npe1work, 11, golden)
        else:
            t = oracle.degrid_angle(i, d.npe1work, 11, golden)
        s, c = ref_numpy.sincosf(t)
        assert tab[i, 0] == c and tab[i, 1] == s, i
    if adjoint and golden:   # index i of the table = spoke (i - z*slide) of slice z
        z, pe = 3, 5
        t = oracle.grid_angle(pe, d.npe1work, 11 + z * d.prof_slide, 1)
        s, c = ref_numpy.sincosf(t)
        assert tuple(tab[z * d.prof_slide + pe]) == (c, s)


@pytest.mark.parametrize("skip", [0, 99_991, 1_000_003, 16_000_000, 250_000_000, -7])
def test_trig_table_bitexact_at_large_angle_indices_and_on_many_threads(skip):
    """The golden-angle wrap (src/tron.cu:372-378: fmodf(PHI * float(pe + skip), 2 pi)) is evaluated by an exact double-precision
    remainder instead of glibc's bit-by-bit fmodf (tron_hostmath.cpp: exact_fmodf_pos), and tables of more than 8 192 entries are
    filled by several threads (tron_plan_retarget's host share): every entry must still be sincosf of libm's fmodf, bit for bit --
    also where fp32 no longer resolves the angle and for a negative index."""
    nz, npe = 130, 402
    cfg = lib.default_config(adjoint=1, golden_angle=1, data_undersamp=0.7852, prof_slide=npe, skip_angles=skip)
    d = lib.derive_dims(cfg, (1, 1, 512, npe * nz, 1))
    n = (d.nz - 1) * d.prof_slide + d.npe1work
    assert n == nz * npe > 8192
    tab = np.zeros((n, 2), np.float32)
    lib.check(lib.load().tron_host_trig_table(ctypes.byref(cfg), ctypes.byref(d), tab.ctypes.data_as(ctypes.c_void_p), n))
    idx = (np.arange(n, dtype=np.int64) + skip).astype(np.float32)
    x = (ref_numpy.PHI * idx).astype(np.float32)
    twopi = np.float32(2.0 * np.pi)
    t = np.fmod(x, twopi).astype(np.float32)                  # numpy's float32 fmod = C fmodf
    t = np.where(t < 0, (t + twopi).astype(np.float32), t)
    for i in list(range(0, n, 53)) + [n - 1]:
        assert np.float32(ref_numpy.modang(x[i])) == t[i]     # (numpy really is libm here)
        s, c = ref_numpy.sincosf(t[i])
        assert tab[i, 0] == c and tab[i, 1] == s, (skip, i)


@pytest.mark.parametrize("nxos,W", [(64, 2.0), (48, 1.5), (33, 2.5)])
def test_band_table_bitexact(nxos, W):
    band = np.zeros((nxos, nxos), np.uint32)
    lib.check(lib.load().tron_host_band_table(nxos, W, band.ctypes.data_as(ctypes.c_void_p)))
    h = nxos // 2
    c = (np.arange(nxos) - h).astype(np.float32)
    R = np.hypot(c[None, :], c[:, None]).astype(np.float32)          # tron.cu:498
    hi = np.minimum(np.floor(R + np.float32(W)), np.float32(h - 1)).astype(np.int64)
    lo = np.maximum(np.ceil(R - np.float32(W)), np.float32(0)).astype(np.int64)
    assert np.array_equal(band & 0xffff, lo.astype(np.uint32))
    assert np.array_equal(band >> 16, hi.astype(np.uint32))


@pytest.mark.parametrize("n,W,sigma", [(64, 2.0, 2.0), (96, 2.0, 1.0), (50, 1.5, 1.25)])
def test_deapod_table_bitexact(oracle, n, W, sigma):
    tab = np.zeros(n * n, np.float32)
    lib.check(lib.load().tron_host_deapod_table(n, W, sigma, tab.ctypes.data_as(ctypes.c_void_p)))
    for idx in list(range(0, n * n, 37)) + [n * n - 1]:
        w = np.float32(oracle.deapod_weight(idx, n, W, sigma))
        want = np.float32(1.0) / (w if w > 0 else np.float32(1.0))
        assert tab[idx] == want, idx


# ----------------------------------------------------------------------------- RawArray

def _ra_struct(L):
    class RaT(ctypes.Structure):
        _fields_ = [("flags", ctypes.c_uint64), ("eltype", ctypes.c_uint64), ("elbyte", ctypes.c_uint64),
                    ("size", ctypes.c_uint64), ("ndims", ctypes.c_uint64),
                    ("dims", ctypes.POINTER(ctypes.c_uint64)), ("data", ctypes.POINTER(ctypes.c_uint8))]
    L.ra_read.restype = ctypes.c_int; L.ra_read.argtypes = [ctypes.POINTER(RaT), ctypes.c_char_p]
    L.ra_write.restype = ctypes.c_int; L.ra_write.argtypes = [ctypes.POINTER(RaT), ctypes.c_char_p]
    L.ra_free.restype = None; L.ra_free.argtypes = [ctypes.POINTER(RaT)]
    L.ra_diff.restype = ctypes.c_int; L.ra_diff.argtypes = [ctypes.POINTER(RaT), ctypes.POINTER(RaT)]
    L.ra_squash.restype = ctypes.c_int; L.ra_squash.argtypes = [ctypes.POINTER(RaT)]
    L.ra_reshape.restype = ctypes.c_int; L.ra_reshape.argtypes = [ctypes.POINTER(RaT), ctypes.POINTER(ctypes.c_uint64), ctypes.c_uint64]
    L.ra_convert.restype = None; L.ra_convert.argtypes = [ctypes.POINTER(RaT), ctypes.c_uint64, ctypes.c_uint64]
    return RaT


def _write_with(writer_lib, RaT, path, arr, eltype, elbyte):
    dims = (ctypes.c_uint64 * arr.ndim)(*arr.shape)
    payload = np.asfortranarray(arr).tobytes(order="F")
    buf = (ctypes.c_uint8 * len(payload)).from_buffer_copy(payload)
    a = RaT(0, eltype, elbyte, len(payload), arr.ndim, dims, ctypes.cast(buf, ctypes.POINTER(ctypes.c_uint8)))
    assert writer_lib.ra_write(ctypes.byref(a), path.encode()) == 0


def test_ra_header_layout(tmp_path):
    data = synth.kspace(2, 8, 3, seed=1)
    path = str(tmp_path / "a.ra")
    ra.write(path, data)
    raw = open(path, "rb").read()
    assert len(raw) == 88 + data.size * 8                     # 5-D header = 88 B
    head = struct.unpack("<11Q", raw[:88])
    assert head[0] == 0x7961727261776172 == 8746397786917265778   # rawrite.m:55
    assert raw[:8] == b"rawarray"
    assert head[1:6] == (0, 4, 8, data.size * 8, 5) and head[6:] == data.shape
    back, h = ra.read(path, with_header=True)
    assert np.array_equal(back, data) and h.dims == data.shape
    # first dimension fastest
    assert np.array_equal(np.frombuffer(raw[88:], np.complex64)[:2], data[:, 0, 0, 0, 0])


def test_ra_c_reader_writer_match_reference(oracle, tmp_path):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref = oracle.ref()
    mine = lib.load()
    RaT = _ra_struct(mine)
    for arr, eltype, elbyte in [(synth.kspace(2, 16, 5, seed=2), 4, 8),
                                (np.arange(24, dtype=np.float32).reshape(2, 3, 4), 3, 4),
                                (np.arange(7, dtype=np.uint16), 2, 2)]:
        p_ref, p_mine, p_py = (str(tmp_path / n) for n in ("ref.ra", "mine.ra", "py.ra"))
        _write_with(ref, oracle.RaT, p_ref, arr, eltype, elbyte)
        _write_with(mine, RaT, p_mine, arr, eltype, elbyte)
        ra.write(p_py, arr)
        b = open(p_ref, "rb").read()
        assert open(p_mine, "rb").read() == b          # our C writer == the reference's writer, byte for byte
        assert open(p_py, "rb").read() == b            # and so is the numpy writer
        # our readers parse what the reference wrote; the reference's reader parses what we wrote
        got = RaT()
        assert mine.ra_read(ctypes.byref(got), p_ref.encode()) == 0
        assert (got.flags, got.eltype, got.elbyte, got.size, got.ndims) == (0, eltype, elbyte, arr.nbytes, arr.ndim)
        assert bytes(ctypes.string_at(got.data, got.size)) == np.asfortranarray(arr).tobytes(order="F")
        assert [got.dims[i] for i in range(arr.ndim)] == list(arr.shape)
        theirs = oracle.RaT()
        assert ref.ra_read(ctypes.byref(theirs), p_mine.encode()) == 0
        assert bytes(ctypes.string_at(theirs.data, theirs.size)) == np.asfortranarray(arr).tobytes(order="F")
        assert np.array_equal(ra.read(p_ref), arr)
        mine.ra_free(ctypes.byref(got))


def test_ra_missing_functions_of_the_reference(tmp_path):
    """ra_reshape / ra_squash / ra_diff / ra_convert are declared at src/ra.h:108-111 and never defined there."""
    L = lib.load()
    RaT = _ra_struct(L)
    arr = synth.kspace(1, 8, 6, seed=3)
    p = str(tmp_path / "x.ra")
    ra.write(p, arr)
    a, b = RaT(), RaT()
    assert L.ra_read(ctypes.byref(a), p.encode()) == 0 and L.ra_read(ctypes.byref(b), p.encode()) == 0
    assert L.ra_diff(ctypes.byref(a), ctypes.byref(b)) == 0
    assert L.ra_squash(ctypes.byref(a)) == 2 and [a.dims[0], a.dims[1]] == [8, 6]
    assert L.ra_diff(ctypes.byref(a), ctypes.byref(b)) != 0
    nd = (ctypes.c_uint64 * 3)(4, 2, 6)
    assert L.ra_reshape(ctypes.byref(a), nd, 3) == 0 and a.ndims == 3
    bad = (ctypes.c_uint64 * 2)(5, 5)
    assert L.ra_reshape(ctypes.byref(a), bad, 2) != 0
    # complex64 -> complex-half -> complex64: equals numpy's float16 rounding
    L.ra_convert(ctypes.byref(b), 4, 4)
    assert (b.elbyte, b.size) == (4, arr.size * 4)
    halves = np.frombuffer(ctypes.string_at(b.data, b.size), np.float16)
    want = np.asfortranarray(arr).reshape(-1, order="F").view(np.float32).astype(np.float16)
    assert np.array_equal(halves.view(np.uint16), want.view(np.uint16))
    L.ra_convert(ctypes.byref(b), 4, 8)
    back = np.frombuffer(ctypes.string_at(b.data, b.size), np.float32)
    assert np.array_equal(back, want.astype(np.float32))
    L.ra_free(ctypes.byref(a)); L.ra_free(ctypes.byref(b))


def test_ra_rejects_garbage(tmp_path):
    L = lib.load()
    RaT = _ra_struct(L)
    p = str(tmp_path / "bad.ra")
    open(p, "wb").write(b"notarawarrayfile" * 8)
    a = RaT()
    assert L.ra_read(ctypes.byref(a), p.encode()) != 0
    with pytest.raises(ValueError):
        ra.read(p)
    good = str(tmp_path / "trunc.ra")
    ra.write(good, synth.kspace(1, 8, 2, seed=4))
    raw = open(good, "rb").read()
    open(good, "wb").write(raw[:-5])
    assert L.ra_read(ctypes.byref(a), good.encode()) != 0
    with pytest.raises(ValueError):
        ra.read(good)
    assert L.ra_read(ctypes.byref(a), str(tmp_path / "missing.ra").encode()) != 0


# ----------------------------------------------------------------------------- binary16

def _edge_floats():
    vals = [0x00000000, 0x80000000, 0x7f800000, 0xff800000, 0x7fc00000, 0x7f800001, 0xffffffff, 0x7f802000,
            0x477fe000, 0x477fefff, 0x477ff000, 0x47800000, 0x38800000, 0x387fffff, 0x387fe000, 0x33000000,
            0x33000001, 0x32ffffff, 0x33800000, 0x33c00000, 0x3f800000, 0x3f801000, 0x3f801001, 0x3f803000, 0x00000001, 0x007fffff]
    return np.array(vals, np.uint32)


def test_half_conversions_match_reference_float16_cu(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    ref, mine = oracle.ref(), lib.load()
    # every half -> float / double
    for h in range(0, 65536):
        assert mine.ra_half_to_float_bits(h) == ref.h2f(h), hex(h)
    for h in range(0, 65536, 17):
        assert mine.ra_half_to_double_bits(h) == ref.h2d(h), hex(h)
    rng = np.random.default_rng(5)
    fbits = np.concatenate([_edge_floats(), rng.integers(0, 2 ** 32, 60000, dtype=np.uint64).astype(np.uint32),
                            # dense sweep through the subnormal-half range, where the tie rule matters
                            (0x33000000 + rng.integers(0, 0x05800000, 40000)).astype(np.uint32)])
    for f in fbits.tolist():
        assert mine.ra_float_to_half_bits(f) == ref.f2h(f), hex(f)
    dbits = rng.integers(0, 2 ** 64, 30000, dtype=np.uint64).tolist()
    dbits += [int(np.float64(np.float32(np.uint32(f).view(np.float32))).view(np.uint64)) for f in _edge_floats() if True]
    dbits += (np.float64(2.0) ** rng.uniform(-26, 17, 20000) * rng.choice([-1, 1], 20000)).view(np.uint64).tolist()
    for d in dbits:
        assert mine.ra_double_to_half_bits(d) == ref.d2h(d), hex(d)


def test_half_matches_numpy_on_normals():
    """Outside the subnormal-half tie quirk the conversion is plain round-to-nearest-even."""
    mine = lib.load()
    rng = np.random.default_rng(6)
    x = (rng.uniform(-4, 4, 20000)).astype(np.float32)
    got = np.array([mine.ra_float_to_half_bits(int(b)) for b in x.view(np.uint32)], np.uint16)
    assert np.array_equal(got, x.astype(np.float16).view(np.uint16))


def _build_c_example(tmp_path):
    import shutil, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "recon_c_abi")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
           os.path.join(root, "examples", "recon_c_abi.c"), "-L" + os.path.join(root, "tron_amd", "lib"), "-ltronhip",
           "-Wl,-rpath," + os.path.join(root, "tron_amd", "lib"), "-Wl,-rpath-link,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def test_headers_are_plain_c99_and_the_c_example_links(tmp_path):
    """include/tron_hip.h and include/rawarray.h compile as pedantic C99 (no HIP or C++ types at the boundary) and a C
    client links against libtronhip.so (examples/recon_c_abi.c)."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    _build_c_example(tmp_path)


def test_numa_cpulist_from_a_fake_sysfs_tree(tmp_path):
    """tron_recon_radial2d_multi binds each per-GPU worker thread to the CPUs of its GPU's NUMA node (the reference has no
    affinity handling): PCI bus id -> numa_node -> cpulist, here against a fake sysfs tree."""
    import ctypes
    L = lib.load()
    dev = tmp_path / "bus" / "pci" / "devices" / "0000:c1:00.0"
    dev.mkdir(parents=True)
    (dev / "numa_node").write_text("1\n")
    node = tmp_path / "devices" / "system" / "node" / "node1"
    node.mkdir(parents=True)
    (node / "cpulist").write_text("32-35,96-97\n")
    buf = (ctypes.c_int * 64)()
    n = L.tron_host_numa_cpulist(str(tmp_path).encode(), b"0000:C1:00.0", buf, 64)       # HIP reports upper-case hex
    assert n == 6 and list(buf[:6]) == [32, 33, 34, 35, 96, 97]
    assert L.tron_host_numa_cpulist(str(tmp_path).encode(), b"0000:c1:00.0", buf, 3) == 3  # truncated, not overrun
    (dev / "numa_node").write_text("-1\n")                                               # single-node host / VM
    assert L.tron_host_numa_cpulist(str(tmp_path).encode(), b"0000:c1:00.0", buf, 64) == 0
    assert L.tron_host_numa_cpulist(str(tmp_path).encode(), b"0000:ff:00.0", buf, 64) == 0  # unknown device
    (dev / "numa_node").write_text("1\n")
    (node / "cpulist").write_text("3-1\n")
    assert L.tron_host_numa_cpulist(str(tmp_path).encode(), b"0000:c1:00.0", buf, 64) == -1


def test_recon_multi_rejects_a_complex64_array_flagged_as_half():
    import numpy as np
    with pytest.raises(ValueError):
        lib.recon_multi(np.zeros((2, 1, 16, 8, 1), np.complex64), adjoint=True, input_half=1)


@pytest.mark.parametrize("nxos", [128, 256, 512, 1024])
def test_band_table_equals_the_scatter_kernels_analytic_test(nxos):
    """grid_scatter_kernel (tron_grid_scatter.hip) tests a point's band, Rlo <= u <= Rhi with R = hypotf(X, Y) in fp32
    (src/tron.cu:498-502), as |X^2 + Y^2 - C| <= D with C = ((u - W)^2 + (u + W)^2) / 2, D = ((u + W)^2 - (u - W)^2) / 2 and u - W
    clamped at 0 -- in fp32, every term exact for W = 2.  The library checks that against its own band table before a plan takes
    the kernel (scatter_band_is_analytic); here the same comparison in numpy float32, for every point and every radius."""
    from tron_amd import lib
    import ctypes
    L = lib.load()
    band = np.zeros(nxos * nxos, dtype=np.uint32)
    L.tron_host_band_table.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    assert L.tron_host_band_table(nxos, ctypes.c_float(2.0), band.ctypes.data_as(ctypes.c_void_p)) == 0
    lo = (band & 0xffff).astype(np.int32).reshape(nxos, nxos)
    hi = (band >> 16).astype(np.int32).reshape(nxos, nxos)
    h = nxos // 2
    X = (np.arange(nxos, dtype=np.float32) - np.float32(h))
    n2 = (X * X)[None, :] + (X * X)[:, None]                     # exact in fp32: integers below 2^24
    W = np.float32(2.0)
    for u in range(0, h):                                        # u <= nxos / 2 - 1, as every sample the kernel sees
        uf = np.float32(u)
        um, up = np.maximum(uf - W, np.float32(0)), uf + W
        A, B = um * um, up * up
        C, D = np.float32(0.5) * (A + B), np.float32(0.5) * (B - A)
        analytic = np.abs(n2 - C) <= D
        assert np.array_equal(analytic, (lo <= u) & (u <= hi)), u
