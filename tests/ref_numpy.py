"""Independent numpy restatement of the reference's gridding / degridding arithmetic in
SCATTER form (one pass over samples instead of the reference's pass over grid points).

Purpose: (1) a second, structurally different statement of tron.cu:465-577 to check the C
oracle against, (2) executable proof that a sample-driven formulation with the
inclusion predicates of SURVEY Q1-Q4 reproduces the reference's point-driven gather --
which is the formulation the HIP kernel uses.  Small sizes only (pure numpy).
"""
import ctypes

import numpy as np

_libm = ctypes.CDLL("libm.so.6")
_libm.sincosf.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
_libm.sincosf.restype = None
_libm.fmodf.argtypes = [ctypes.c_float, ctypes.c_float]
_libm.fmodf.restype = ctypes.c_float

f32 = np.float32
PHI = f32(1.9416089796736116)  # tron.cu:90


def sincosf(t):
    s, c = ctypes.c_float(), ctypes.c_float()
    _libm.sincosf(ctypes.c_float(float(t)), ctypes.byref(s), ctypes.byref(c))
    return f32(s.value), f32(c.value)


def modang(x):  # tron.cu:372-378
    twopi = f32(2.0 * np.pi)
    y = f32(_libm.fmodf(ctypes.c_float(float(x)), ctypes.c_float(float(twopi))))
    return f32(y + twopi) if y < 0 else y


def grid_angle(pe, npe, skip, golden):  # tron.cu:509
    if golden:
        return modang(PHI * f32(pe + skip))
    return f32(float(f32(pe) * f32(2.0)) * np.pi / float(f32(npe)) + np.pi * 0.5)


def degrid_angle(pe, npe, skip, golden):  # tron.cu:555
    if golden:
        return modang(PHI * f32(pe + skip))
    return f32(pe * np.pi / float(f32(npe)))


_NUM = [0.210580722890567e-22, 0.380715242345326e-19, 0.479440257548300e-16, 0.435125971262668e-13,
        0.300931127112960e-10, 0.160224679395361e-7, 0.654858370096785e-5, 0.202591084143397e-2,
        0.463076284721000e0, 0.754337328948189e2, 0.830792541809429e4, 0.571661130563785e6,
        0.216415572361227e8, 0.356644482244025e9, 0.144048298227235e10]


def besseli0(x):  # tron.cu:304-321, vectorised; x float32 array
    x = np.asarray(x, f32)
    z = (x * x).astype(f32).astype(np.float64)
    num = np.full_like(z, _NUM[0])
    for c in _NUM[1:]:
        num = z * num + c
    # the reference's nested expression is Horner with c0 innermost and "+ c14" last, in double
    num = num.astype(f32)
    den = (z * (z * (z - 0.307646912682801e4) + 0.347626332405882e7) - 0.144048298227235e10).astype(f32)
    out = (-num / den).astype(f32)
    return np.where(x == 0, f32(1.0), out)


def gridkernel(x, W=2.0):  # tron.cu:338-349
    x = np.asarray(x, f32)
    W = f32(W)
    beta = f32(f32(2.34) * f32(2.0)) * W
    inside = np.abs(x) < W
    r = (x / W).astype(f32)
    f = np.sqrt(np.maximum(f32(1.0) - r * r, f32(0))).astype(f32)
    val = (f32(0.5) * besseli0((beta * f).astype(f32)) / W).astype(f32)
    return np.where(inside, val, f32(0))


def precompensate(nudata):  # tron.cu:405-416; nudata (npe, nro, nchan)
    npe, nro, _ = nudata.shape
    a = f32(f32(2.0) - f32(2.0) / f32(npe)) / f32(nro)
    b = f32(1.0) / f32(npe)
    r = np.arange(nro, dtype=f32)
    sdc = (a * np.abs(r - f32(nro // 2)) + b).astype(f32)
    out = nudata.astype(np.complex64).copy()
    out.real *= sdc[None, :, None]
    out.imag *= sdc[None, :, None]
    return out


def grid_scatter(nudata, nxos, W=2.0, skip=0, golden=1):
    """Sample-driven restatement of gridradial2d (tron.cu:465-536).
    nudata (npe, nro, nchan) already density-compensated -> (nxos, nxos, nchan)."""
    nudata = np.asarray(nudata, np.complex64)
    npe, nro, nchan = nudata.shape
    h = nxos // 2
    Wf = f32(W)
    cw = int(np.ceil(W))
    # per-point radial band (tron.cu:498-502)
    coords = np.arange(nxos, dtype=np.int64) - h
    Rg = np.hypot(coords[None, :].astype(f32), coords[:, None].astype(f32)).astype(f32)  # [Y, X]
    Rhi = np.minimum(np.floor(Rg + Wf), f32(h - 1)).astype(np.int64)
    Rlo = np.maximum(np.ceil(Rg - Wf), f32(0)).astype(np.int64)
    acc_re = np.zeros((nxos, nxos, nchan), f32)
    acc_im = np.zeros((nxos, nxos, nchan), f32)
    rr = np.arange(-(h - 1), h, dtype=np.int64)           # Q3: |r| <= nxos/2-1
    ridx = np.trunc(rr * nro / nxos).astype(np.int64)     # Q4: C division truncates toward 0
    ridx = np.where(rr * nro % nxos == 0, rr * nro // nxos, ridx)
    for pe in range(npe):
        st, ct = sincosf(grid_angle(pe, npe, skip, golden))
        kx = (rr.astype(f32) * ct).astype(f32)
        ky = (rr.astype(f32) * st).astype(f32)
        fx = np.floor(kx).astype(np.int64)
        fy = np.floor(ky).astype(np.int64)
        d = nudata[pe, ridx + nro // 2, :]                # (nr, nchan)
        for j in range(2 * cw):
            Y = fy - cw + 1 + j
            wy = gridkernel((ky - Y.astype(f32)).astype(f32), W)
            for i in range(2 * cw):
                X = fx - cw + 1 + i
                wx = gridkernel((kx - X.astype(f32)).astype(f32), W)
                wgt = (wx * wy).astype(f32)
                ok = (wgt > 0) & (X >= -h) & (X < nxos - h) & (Y >= -h) & (Y < nxos - h)
                Xi = np.clip(X + h, 0, nxos - 1)
                Yi = np.clip(Y + h, 0, nxos - 1)
                ar = np.abs(rr)
                ok &= (ar >= Rlo[Yi, Xi]) & (ar <= Rhi[Yi, Xi])          # Q1 band predicate
                mult = np.where((rr == 0) & (Rlo[Yi, Xi] == 0), 2, 1)    # Q2: r = 0 visited by both loops
                sel = np.nonzero(ok)[0]
                for m in (1, 2):
                    s2 = sel[mult[sel] >= m]
                    np.add.at(acc_re, (Yi[s2], Xi[s2]), (d[s2].real * wgt[s2, None]).astype(f32))
                    np.add.at(acc_im, (Yi[s2], Xi[s2]), (d[s2].imag * wgt[s2, None]).astype(f32))
    scale = f32(f32(1.0) / f32(nxos)) / f32(npe)
    return ((acc_re * scale) + 1j * (acc_im * scale)).astype(np.complex64)


def degrid_gather(udata, nro, npe, W=2.0, skip=0, golden=1):
    """degridradial2d (tron.cu:540-577).  udata (n, n, nrep) -> (npe, nro, nrep)."""
    udata = np.asarray(udata, np.complex64)
    n, _, nrep = udata.shape
    Wf = f32(W)
    out = np.zeros((npe, nro, nrep), np.complex64)
    ro = np.arange(nro)
    R = (ro.astype(f32) / f32(nro) - f32(0.5)).astype(f32)
    half = f32((n + 1) // 2)
    for pe in range(npe):
        s, c = sincosf(degrid_angle(pe, npe, skip, golden))
        X = ((f32(n) * R).astype(f32) * s + half).astype(f32)
        Y = ((f32(n) * R).astype(f32) * c + half).astype(f32)
        x0 = np.ceil(X - Wf).astype(np.int64)
        y0 = np.ceil(Y - Wf).astype(np.int64)
        acc_re = np.zeros((nro, nrep), f32)
        acc_im = np.zeros((nro, nrep), f32)
        span = int(np.floor(2 * W)) + 1
        for a in range(span):
            xu = x0 + a
            okx = xu.astype(f32) <= (X + Wf).astype(f32)
            wx = gridkernel((xu.astype(f32) - X).astype(f32), W)
            for b in range(span):
                yu = y0 + b
                oky = yu.astype(f32) <= (Y + Wf).astype(f32)
                wgt = (wx * gridkernel((yu.astype(f32) - Y).astype(f32), W)).astype(f32)
                wgt = np.where(okx & oky, wgt, f32(0))
                i = (xu + n) % n
                j = (yu + n) % n
                u = udata[i, j, :]
                acc_re = (acc_re + (u.real * wgt[:, None]).astype(f32)).astype(f32)
                acc_im = (acc_im + (u.imag * wgt[:, None]).astype(f32)).astype(f32)
        out[pe] = acc_re + 1j * acc_im
    return out
