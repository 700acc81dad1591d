"""CPU simulation of grid_arc_kernel's gather loops on the metric trajectory (512^2 grid, 402 golden-angle spokes, W = 2):
per tile, per comb (batch), per thread (2x2 block) the members of its angular window and their radius counts; a wave runs
max-over-lanes.  Reproduces the wave-level counters of tools/arcprof.py (member loop 7.9 k iterations per slice at 0.70 lanes,
radius loop 36.5 k at 0.50) and prices loop structures before they are built:
  base     member loop / radius loop as in the kernel
  sorted   each lane visits its members longest chord first
  flat     one loop over a lane's visits of a batch (max over lanes of the sum)
  whole    flat over the whole tile (no batches): the bound set by the tile itself
  sorted in chunks of n: the members are clipped n at a time, sorted by chord length in registers, then visited
usage: python tools/probe/arc_util_sim.py [records per batch] [wave shape: 16x4 | 8x8 | 4x16 | auto] [npe]"""
import sys
import numpy as np

n = 512; h = 256; rmax = 255; W = 2.0; T = 32; R0 = 14
NREC = int(sys.argv[1]) if len(sys.argv) > 1 else 608
SHAPE = sys.argv[2] if len(sys.argv) > 2 else "16x4"
npe = int(sys.argv[3]) if len(sys.argv) > 3 else 402
PHI = np.float32(1.9416089796736116)
pe = np.arange(npe, dtype=np.float32)
t = np.fmod((PHI * pe).astype(np.float32).astype(np.float64), 2 * np.pi)
phi = np.mod(t, np.pi)
order = np.argsort(phi, kind="stable")
phis = phi[order]; cs_c = np.cos(t)[order]; cs_s = np.sin(t)[order]


def rcp(x):
    return np.where(np.abs(x) > 1e-12, 1.0 / np.where(x == 0, 1, x), np.copysign(1e12, x))


def tile(tx, ty):
    x0 = tx * T - h; y0 = ty * T - h
    outer = (x0 in (0, -T)) and (y0 in (0, -T))
    ax = max(max(x0, -(x0 + T - 1)), 0); ay = max(max(y0, -(y0 + T - 1)), 0)
    if ax * ax + ay * ay > (rmax + W + 1) ** 2:
        return None
    eps = 0.01
    bxl, bxh, byl, byh = x0 - W - eps, x0 + T - 1 + W + eps, y0 - W - eps, y0 + T - 1 + W + eps
    ic, isn = rcp(cs_c), rcp(cs_s)
    xa, xb, ya, yb = bxl * ic, bxh * ic, byl * isn, byh * isn
    lo = np.maximum(np.maximum(np.minimum(xa, xb), np.minimum(ya, yb)), -rmax)
    hi = np.minimum(np.minimum(np.maximum(xa, xb), np.maximum(ya, yb)), rmax)
    rlo = np.ceil(lo).astype(int); rhi = np.floor(hi).astype(int)
    ok = lo <= hi
    if outer:
        a = ok & (rhi >= R0); rlo = np.where(a, np.maximum(rlo, R0), rlo)
        b = ok & ~a & (rlo <= -R0); rhi = np.where(b, np.minimum(rhi, -R0), rhi)
        ok = ok & (a | b)
    ok &= rhi >= rlo
    neg = rhi < 0
    ulo = np.where(neg, -rhi, rlo); ln = np.where(ok, rhi - rlo + 1, 0)
    c = np.where(neg, -cs_c, cs_c); s = np.where(neg, -cs_s, cs_s)
    wrap = np.mod(np.arctan2(y0 + 15.5, x0 + 15.5) + 0.5 * np.pi, np.pi)
    ph = np.where(phis < wrap, phis + np.pi, phis)
    pos = np.where(phis < wrap, np.arange(npe) + npe, np.arange(npe))
    sel = np.where(ln > 0)[0]
    if sel.size == 0:
        return None
    umin, umax = pos[sel].min(), pos[sel].max()
    keep = (pos >= umin) & (pos <= umax)
    idx = np.argsort(pos[keep], kind="stable")
    ulo, ln, c, s, ph = ulo[keep][idx], ln[keep][idx], c[keep][idx], s[keep][idx], ph[keep][idx]
    ns = ulo.size; total = ln.sum()
    K = max(1, -(-total // NREC))
    while True:
        if max(ln[b::K].sum() for b in range(K)) <= NREC:
            break
        K += 1
    # threads
    bx, by = np.meshgrid(np.arange(16), np.arange(16))     # block col, row
    X0 = (x0 + 2 * bx).ravel().astype(float); Y0 = (y0 + 2 * by).ravel().astype(float)
    lo_b = np.full(256, 1 << 20); hi_b = np.full(256, -1)
    for q in range(4):
        X = X0 + (q & 1); Y = Y0 + (q >> 1)
        R = np.hypot(X, Y)
        l = np.maximum(np.ceil(R - W), 0); u = np.minimum(np.floor(R + W), rmax)
        v = l <= u
        lo_b = np.where(v, np.minimum(lo_b, l), lo_b); hi_b = np.where(v, np.maximum(hi_b, u), hi_b)
    umin_t = max(R0, 1) if outer else 1
    has = (lo_b <= hi_b) & (hi_b >= umin_t)
    blo = np.maximum(lo_b, umin_t); bhi = hi_b
    Xc = X0 + 0.5; Yc = Y0 + 0.5
    Tt = np.mod(np.arctan2(Yc, Xc), np.pi); Tt = np.where(Tt < wrap, Tt + np.pi, Tt)
    Rr = np.hypot(Xc, Yc)
    sd0 = (W + 0.52) * 1.41421356 / Rr
    wcs = np.minimum(1.41421356, (np.abs(Xc) + np.abs(Yc)) / Rr + 1.5 * sd0)
    sd = (W + 0.52) * wcs / Rr
    D = np.where(sd < 0.999, np.arcsin(np.minimum(sd, 0.999)) + 2e-3, 4.0)
    inwin = (ph[None, :] >= (Tt - D)[:, None]) & (ph[None, :] <= (Tt + D)[:, None]) & has[:, None]
    We = W + 1e-3
    ic, isn = rcp(c)[None, :], rcp(s)[None, :]
    xa, xb = (X0 - We)[:, None] * ic, (X0 + 1 + We)[:, None] * ic
    ya, yb = (Y0 - We)[:, None] * isn, (Y0 + 1 + We)[:, None] * isn
    lo = np.maximum(np.maximum(np.minimum(xa, xb), np.minimum(ya, yb)), np.maximum(ulo[None, :], blo[:, None]))
    hi = np.minimum(np.minimum(np.maximum(xa, xb), np.maximum(ya, yb)), np.minimum((ulo + ln - 1)[None, :], bhi[:, None]))
    cnt = np.floor(hi).astype(int) - np.ceil(lo).astype(int) + 1
    cnt = np.where(inwin & (cnt > 0), cnt, 0)                 # [thread, entry]
    return dict(x0=x0, y0=y0, K=K, ns=ns, total=total, cnt=cnt, inwin=inwin, bx=bx.ravel(), by=by.ravel())


def wave_of(shape, bx, by, x0, y0):
    if shape == "auto":
        shape = "4x16" if abs(x0 + 16) >= abs(y0 + 16) else "16x4"
    if shape == "16x4":
        return by // 4
    if shape == "4x16":
        return bx // 4
    if shape == "8x8":
        return (by // 8) * 2 + bx // 8
    if shape == "rows4":                      # wave w owns the block rows w, w + 4, w + 8, w + 12 (interleaved strips)
        return by % 4
    if shape == "cols4":
        return bx % 4
    if shape == "checker":                    # 2x2 super-blocks dealt round the four waves
        return (by % 2) * 2 + bx % 2
    raise ValueError(shape)


tot = dict(visits=0, mem_it=0, mem_act=0, rad_it=0, sorted_it=0, flat_it=0, whole_it=0, K=0, tiles=0)
for ty in range(16):
    for tx in range(16):
        r = tile(tx, ty)
        if r is None:
            continue
        wv = wave_of(SHAPE, r["bx"], r["by"], r["x0"], r["y0"])
        cnt, inwin, K = r["cnt"], r["inwin"], r["K"]
        tot["tiles"] += 1; tot["K"] += K
        tot["visits"] += cnt.sum()
        per_wave = np.zeros((4, K))
        for w in range(4):
            lanes = wv == w
            cw, iw = cnt[lanes], inwin[lanes]
            tot["whole_it"] += cw.sum(1).max()
            for b in range(K):
                cb, ib = cw[:, b::K], iw[:, b::K]
                nm = ib.sum(1)                      # members per lane: the window is contiguous, so are its comb members
                M = nm.max()
                tot["mem_it"] += M; tot["mem_act"] += nm.sum()
                if M == 0:
                    continue
                # k-th member of each lane
                lens = np.zeros((cb.shape[0], M), int)
                for li in range(cb.shape[0]):
                    lens[li, :nm[li]] = cb[li, ib[li]]
                tot["rad_it"] += lens.max(0).sum()
                tot["sorted_it"] += (-np.sort(-lens, axis=1)).max(0).sum()
                tot["flat_it"] += lens.sum(1).max()
                for mm in (3, 4, 6):
                    for c0 in range(0, M, mm):
                        ch = -np.sort(-lens[:, c0:c0 + mm], axis=1)
                        mx = ch.max(0)
                        tot[f"s{mm}_it"] = tot.get(f"s{mm}_it", 0) + mx.sum()
                        tot[f"s{mm}_mem"] = tot.get(f"s{mm}_mem", 0) + (mx > 0).sum()
                        if mm == 4:
                            per_wave[w, b] += mx.sum()
                            tot["flat4_it"] = tot.get("flat4_it", 0) + lens[:, c0:c0 + mm].sum(1).max()     # one loop over a lane's visits of the chunk
        tot["wg_sum"] = tot.get("wg_sum", 0) + per_wave.sum()
        tot["wg_max"] = tot.get("wg_max", 0) + 4 * per_wave.max(0).sum()
v = tot["visits"]
print(f"NREC {NREC} shape {SHAPE} npe {npe}: tiles {tot['tiles']}, batches {tot['K']}, visits {v}")
print(f"  member loop  {tot['mem_it']} wave iterations, lanes active {tot['mem_act'] / (64 * tot['mem_it']):.3f}")
for k, name in (("rad_it", "base"), ("sorted_it", "sorted"), ("flat_it", "flat"), ("whole_it", "whole")):
    print(f"  radius loop {name:7s} {tot[k]:7d} wave iterations, lanes active {v / (64 * tot[k]):.3f}")
print(f"  waves of a workgroup meet at a barrier after every batch: sum of their iterations {tot['wg_sum']:.0f}, 4 x the slowest wave's {tot['wg_max']:.0f} "
      f"-> {tot['wg_sum'] / tot['wg_max']:.3f} of the workgroup's gather time is work")
print(f"  flat over chunks of 4 (a member switch inside the hot loop): {tot['flat4_it']:7d} wave iterations, lanes active {v / (64 * tot['flat4_it']):.3f}")
for mm in (3, 4, 6):
    print(f"  sorted in chunks of {mm}: {tot[f's{mm}_it']:7d} wave iterations, lanes active {v / (64 * tot[f's{mm}_it']):.3f}; member visits {tot[f's{mm}_mem']}")
