"""LDS bank conflicts of grid_scatter_kernel's ds_add_u64 by row stride of the tile of sums (tooling; CPU only).

A wave instruction adds at one footprint offset (i, j) for 64 consecutive records along a spoke (a run is dealt to the waves in record
order): lane l's 8-byte word is  (by + i) * P + bx + j,  (bx, by) = the first column / row of its footprint.  A 64-bit LDS operation is
served in two groups of 32 lanes, one LDS cycle per lane on a group's busiest bank pair (word mod 32) -- an atomic serialises lanes on
one address too.  Prints the mean cycles per group for the metric trajectory (512^2 grid, 402 golden-angle spokes, radii 5 .. 255) by
row stride P, so that a stride can be priced before it is built (the round-5 counters: 51 % of the kernel's LDS-active
cycles are conflicts at P = 74).
    python tools/probe/scatter_bank_sim.py [tile = 64]"""
import sys
import numpy as np

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 64
halo, W, n, npe = 4, 2.0, 512, 402
PHI = np.float32(1.9416089796736116)
rng = np.random.default_rng(1)


def groups(P):
    cyc, cnt = 0.0, 0
    for pe in range(0, npe, 3):
        t = np.float32(np.fmod(PHI * np.float32(pe), np.float32(2 * np.pi)))
        c, s = np.cos(t), np.sin(t)
        for sign in (1.0, -1.0):
            u = np.arange(5, 256, dtype=np.float64)
            kx, ky = sign * u * c, sign * u * s
            ix, iy = np.floor(kx - W).astype(np.int64) + 1, np.floor(ky - W).astype(np.int64) + 1
            # records of one tile: consecutive radii whose footprint origin lies in the same tile (+ halo): cut the spoke where the tile changes
            tx, ty = (ix + 256) // tile, (iy + 256) // tile
            key = tx * 64 + ty
            start = 0
            for k in range(1, len(u) + 1):
                if k == len(u) or key[k] != key[start]:
                    bx = ix[start:k] + 256 - tx[start] * tile + halo
                    by = iy[start:k] + 256 - ty[start] * tile + halo
                    a = by * P + bx
                    for g0 in range(0, len(a), 32):
                        g = a[g0:g0 + 32]
                        if len(g) < 8:
                            continue
                        banks = g % 32
                        cyc += np.bincount(banks, minlength=32).max()
                        cnt += 1
                    start = k
    return cyc / max(cnt, 1)


for P in range(tile + 2 * halo, tile + 2 * halo + 14):
    print(f"row stride {P:3d} (8-byte words): {groups(P):.2f} LDS cycles per group of <= 32 lanes")
