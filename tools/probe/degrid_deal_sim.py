"""How should the streaming degridding kernel deal a tile's records to its lanes?  (round 5)

A ds_read_b64 serves the lanes of a wave in two groups of 32; a group takes one LDS cycle per distinct address on its busiest
pair of banks (bank pair = (byte address / 8) mod 32).  All 16 x coils reads of a record sit at FIXED offsets from the point
its footprint starts at, so a group is conflict-free for every read iff its lanes' start points are distinct mod 32.
This script replays the forward bench's trajectory (512 golden-angle spokes x 512 samples on the 512^2 grid, W = 2) tile by
tile and counts, per dealing, the 32-lane groups (VALU issue) and the LDS cycles per read (the groups' worst multiplicities):
  spoke     records in list order (64 consecutive samples of a spoke)
  sorted    counting-sorted by start point (what round 3 built)
  rows(T)   rank r within the class (start point mod 32, ranked in start-point order) -> row r; rows at least T wide are padded
            to a whole group, narrower ones follow each other without gaps
"""
import numpy as np, sys

n = 512; nro = 512; npe = int(sys.argv[1]) if len(sys.argv) > 1 else 512; W = 2.0
HALO, PITCH, TILE = 2, 38, 32
ga = np.float32(111.246117975 * np.pi / 180)
pe = np.arange(npe)
t = np.mod(ga.astype(np.float64) * pe, 2 * np.pi)
ro = np.arange(nro)
R = ro / nro - 0.5
X = n * R[None, :] * np.sin(t)[:, None] + 256
Y = n * R[None, :] * np.cos(t)[:, None] + 256
fx = np.clip(np.floor(X).astype(int), 0, n - 1); fy = np.clip(np.floor(Y).astype(int), 0, n - 1)
tile = (fx // TILE) * (n // TILE) + fy // TILE
xu0 = np.ceil(X - W).astype(int); yu0 = np.ceil(Y - W).astype(int)
order = np.arange(npe * nro).reshape(npe, nro)          # list order: spoke by spoke

def group_cycles(slots):
    """slots: start points in lane order; groups of 32; cycles = max over bank pairs of distinct addresses"""
    cyc = 0; ng = 0
    for g0 in range(0, len(slots), 32):
        s = slots[g0:g0 + 32]
        s = s[s >= 0]
        if len(s) == 0:
            continue
        ng += 1
        u = np.unique(s)
        cyc += np.bincount(u % 32).max()
    return ng, cyc

tot = {}
def add(name, ng, cyc, nrec):
    a = tot.setdefault(name, [0, 0, 0]); a[0] += ng; a[1] += cyc; a[2] += nrec

ntile = (n // TILE) ** 2
kept = 0
for tl in range(ntile):
    m = tile == tl
    if not m.any():
        continue
    tx0 = (tl // (n // TILE)) * TILE; ty0 = (tl % (n // TILE)) * TILE
    slot = ((xu0[m] + HALO - tx0) * 1 + (yu0[m] + HALO - ty0) * PITCH)
    nrec = len(slot)
    if nrec > 1536:
        continue                       # the centre tiles: not kept in registers
    kept += nrec
    add("spoke", *group_cycles(slot[np.argsort(order[m], kind="stable")]), nrec)
    ss = np.sort(slot)
    add("sorted", *group_cycles(ss), nrec)
    cls = ss % 32
    rank = np.zeros(nrec, int)
    cnt = np.zeros(32, int)
    for i in range(nrec):
        rank[i] = cnt[cls[i]]; cnt[cls[i]] += 1
    for T in (33, 28, 24, 20, 16, 1):
        width = np.array([(cnt > r).sum() for r in range(cnt.max())])
        pos0 = np.zeros(len(width) + 1, int)
        p = 0
        for r, w in enumerate(width):
            if w >= T:
                p = (p + 31) // 32 * 32
                pos0[r] = p; p += 32
            else:
                pos0[r] = p; p += w
        lanes = -np.ones((p + 31) // 32 * 32, int)
        for i in range(nrec):
            r, c = rank[i], cls[i]
            if width[r] >= T:
                lanes[pos0[r] + c] = ss[i]
            else:
                lanes[pos0[r] + (cnt[:c] > r).sum()] = ss[i]
        ng, cyc = group_cycles(lanes)
        add(f"rows({T})", ng, cyc, nrec)
        a = tot.setdefault(f"rows({T})", None)
        if len(lanes) > 1536:
            tot.setdefault(f"rows({T}) overflow tiles", [0, 0, 0])[0] += 1
print(f"{npe} spokes: {kept} records in tiles of <= 1536")
for k, (ng, cyc, nrec) in tot.items():
    if nrec:
        print(f"{k:10s} groups {ng:6d} ({ng * 32 / nrec:5.2f} lane slots per record)  LDS cycles per read {cyc:6d} ({cyc / ng:4.2f} per group, {cyc * 32 / nrec:5.2f} per 32 records)")
    else:
        print(f"{k}: {ng}")
