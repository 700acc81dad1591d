// Gridding kernel of the TRON_KB_FAST path for radial trajectories: every thread WALKS THE ARC of spokes that can reach
// its 2x2 grid points.
//
// Same (sample, point) pairs and the same arithmetic per pair as the reference's gridradial2d (src/tron.cu:465-536: kx, ky
// :514-515, weights :516, band :498-502,:512, density compensation :412-414, scale :532), found the way the reference
// finds them -- per point, per spoke, the radii inside the band -- but with the spokes SORTED BY LINE ANGLE: the spokes
// whose samples can fall inside a point's Kaiser-Bessel footprint are then one contiguous run of the sorted list (those
// within asin((W + 1/2)(|cos| + |sin|) / R) of the point's own direction), and on each of them the radii are one interval
// that the thread clips analytically.  No sort of samples by cell, no per-sample staging pass, no per-batch planning: the
// only per-sample work is one LDS-DMA copy.  Sums run spoke-angle-major instead of acquisition-major and the window comes
// from a table (quadratic interpolation, 1e-7 of the reference's expression), so results agree with the reference to fp32
// summation-order noise (~1e-7 relative L2), not bit for bit: TRON_KB_FAST only.
//
// Everything that depends on the trajectory alone is worked out ONCE, when the plan is created (arc_prep_kernel: the angle
// sets of all slices are known then), and kept in HBM, ~20 bytes per (slice, tile, crossing spoke):
//   clip    thread = spoke of the angle-sorted list (host tables).  The spokes that cross tile + halo are one run of the
//           list (a convex region subtends an interval of directions); the run is cut at the direction perpendicular to the
//           tile centre so that it never wraps.  Segment = u in [ulo, ulo + len) with the spoke's direction flipped where
//           needed so that u = |r| > 0 (the tiles of this kernel never hold r = 0: the k-space centre belongs to the inner
//           tile of the binned kernel, GridParams::inner_r0).
//   window  thread = 2x2 block of the tile (the gridding kernel's layout): its run of the list, the spokes within
//           asin((W + 1/2)(|cos| + |sin|) / R) of the block's own direction, by binary search on the line angles;
//   deal    the run is dealt round-robin into K batches ("combs": batch b = entries b, b + K, ...), K = the fewest batches
//           whose records fit in half the LDS sample space.  A comb covers the whole tile evenly, so every thread has work
//           in every batch; each entry gets its record offset inside its comb.
// grid_arc_kernel, one workgroup = 4 waves = one 32x32 tile, each thread 2x2 points, CPB coils in registers; per slice:
//   table   the tile's run (<= 512 entries of 16 bytes) is copied to LDS; each thread reads its own run [jlo, jhi] of the list
//           (found by arc_prep_kernel, once per plan, by binary search on the line angles: 1 KB per tile and slice);
//   batch   the comb's samples are copied global -> LDS by global_load_lds_dwordx4 (one wave instruction per spoke
//           segment and coil pair; everything but the lane's byte offset in scalar registers, the segments' table entries read 64
//           at a time).  The LDS-DMA path moves ~8 bytes per clock and CU (a batch of 34 KB: 4.3 k cycles, 16 % of the wave
//           cycles) however the bytes are packed into instructions -- measured in round 4, all +-1 %: record-major copies (16
//           samples x 4 coil pairs per instruction, a quad of lanes = one 64-byte line), an L2 prefetch of the next batch, and
//           full-lane instructions over a per-record source table in LDS (36 instead of ~60 instructions per batch);
//   gather  thread: its comb members four at a time: clip each spoke against the 2x2 block + footprint (a packed descriptor:
//           radii, first record, run entry), sort the four by chord length in registers (a wave runs the longest chord of
//           its lanes: with every lane's longest first the k-th chords of a wave match far better), then per chord and radius:
//           (kx, ky) as one packed product, ONE table position per axis serving both columns (rows) of the block (pair table,
//           build_kb_pair_lut), band tests, 4 points x CPB coils of packed FMAs from four 16-byte LDS reads.
// No floating-point atomics; the output is written once, coil-planar, in FFT-native order (tron_grid_store.h).
#include <stdlib.h>

#include "tron_device.h"
#include "tron_grid_store.h"

namespace tron {

constexpr int kArcTile = 32;
constexpr int kArcThreads = 256;
constexpr int kArcMaxSpokes = 512;     // spokes of one tile's run
constexpr int kArcMaxBatches = 96;
constexpr int kArcSeg = 64;            // longest spoke segment through tile + halo: (32 + 2 * 3) sqrt(2) = 54
constexpr float kPi = 3.14159265358979f;

// one and two coils: five workgroups per CU (83 / 94 VGPRs) beat four with larger batches by 3-4 % / 1 % (same-box A/B)
constexpr int kArcNrec1 = 1776, kArcWaves1 = 5, kArcNrec2 = 888, kArcWaves2 = 5;
template <int CPB>
struct ArcCfg {
    // sample records per batch.  One buffer: same-box A/B at 8 coils, gridding us per coil-slice: two buffers of 304 records
    // (batch b + 1 copied while batch b is gathered) 1.69-1.70, of 256 1.72-1.75; ONE buffer of 608 1.58-1.59 -- half the
    // batches means twice the comb members per thread and batch, and the lanes of a wave run far more evenly.
    // Three workgroups per CU at 6 and 8 coils (the accumulators allow 3 waves per SIMD anyway), four below.
    // LDS is handed out in units of 1280 bytes on gfx950 (160 KiB / 128): WAVES workgroups per CU need ArcLds <= 42 / 32 / 25 units
    // for 3 / 4 / 5 (a first build of this table at 54 144 bytes ran TWO workgroups per CU, not three, and hid a 25 % saving)
    static constexpr int NREC = CPB >= 8 ? 560 : (CPB >= 6 ? 744 : (CPB >= 4 ? 720 : (CPB >= 2 ? kArcNrec2 : kArcNrec1)));
    static constexpr int WAVES = CPB >= 6 ? 3 : (CPB >= 4 ? 4 : (CPB >= 2 ? kArcWaves2 : kArcWaves1));
    static constexpr int NBUF = 1;
};

// samples in LDS: [coil pair][record] float4 for even coil counts, [re | im][record] float for one coil
template <int CPB>
constexpr int arc_rec_bytes() { return CPB * 8; }

template <int CPB>
struct ArcLds;
template <int CPB>
constexpr bool arc_lds_fits() { return (sizeof(ArcLds<CPB>) + 1279) / 1280 * ArcCfg<CPB>::WAVES <= 128; }
template <int CPB>
struct ArcLds {
    float2 lut[3 * kArcLutEntries];        // Kaiser-Bessel pair table, planes c0 | c1 | c2: entry = (this column, the next one), build_kb_pair_lut
    unsigned s_a[kArcMaxSpokes];           // run entry: sample index of its first record | (records run downwards) << 31
    unsigned s_b[kArcMaxSpokes];           //            ulo | len << 10 | record offset inside its batch << 17
    float2 s_cs[kArcMaxSpokes];            //            (cos, sin) of the (flipped) direction
    float4 d[ArcCfg<CPB>::NBUF * ArcCfg<CPB>::NREC * CPB / 2];   // samples [buffer][coil pair][record] (one coil: [buffer][re | im][record] floats)
};

int grid_arc_nrec(int nchan, int half_in)  // records per batch for a plan of nchan channels (must match launch_grid_arc's choice of CPB)
{
    if (half_in && nchan < 5) return ArcCfg<4>::NREC;
    if (nchan >= 5) {
        if (half_in) return nchan == 6 ? ArcCfg<6>::NREC : ArcCfg<8>::NREC;
        const int pad8 = (nchan + 7) / 8 * 8, pad6 = (nchan + 5) / 6 * 6;
        return pad6 < pad8 ? ArcCfg<6>::NREC : ArcCfg<8>::NREC;
    }
    return nchan >= 3 ? ArcCfg<4>::NREC : (nchan >= 2 ? ArcCfg<2>::NREC : ArcCfg<1>::NREC);
}

// Phase clock of tools/arcprof.py (-DTRON_PHASE_CLOCK builds only): shader-clock cycles per wave and phase plus loop
// counters, summed over all waves of all launches since the last read; production builds carry none of it.
#ifdef TRON_PHASE_CLOCK
constexpr int kArcProfSlots = 16, kArcProfCopies = 4096;
__device__ unsigned long long g_arc_prof[kArcProfCopies * kArcProfSlots];
#define APROF_DECL unsigned prof_acc[kArcProfSlots] = {}; unsigned long long prof_t = __builtin_readcyclecounter(); const unsigned long long prof_c0 = prof_t, prof_r0 = __builtin_amdgcn_s_memrealtime()
#define APROF_MARK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[i] += (unsigned)(t_ - prof_t); prof_t = t_; } while (0)
#define APROF_COUNT(i, v) do { prof_acc[i] += (unsigned)(v); } while (0)
#define APROF_FLUSH do { prof_acc[8] = (unsigned)(__builtin_readcyclecounter() - prof_c0); prof_acc[9] = (unsigned)(__builtin_amdgcn_s_memrealtime() - prof_r0); for (int i_ = 12; i_ < 16; ++i_) for (int o_ = 32; o_ > 0; o_ >>= 1) prof_acc[i_] += __shfl_xor(prof_acc[i_], o_); if (lane == 0) { for (int i_ = 0; i_ < kArcProfSlots; ++i_) if (prof_acc[i_]) atomicAdd(&g_arc_prof[((blockIdx.x * 4 + wave) % kArcProfCopies) * kArcProfSlots + i_], (unsigned long long)prof_acc[i_]); } } while (0)
#else
#define APROF_DECL
#define APROF_MARK(i)
#define APROF_COUNT(i, v)
#define APROF_FLUSH
#endif

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) v4f *lds_f4p;
typedef const __attribute__((address_space(3))) v2f *lds_f2p;

// A thread's 2x2 block of its tile (the layout both kernels below use) and the block's angular window: a spoke of direction phi
// reaches the block's footprint only if its line passes within (W + 1/2)(|cos phi| + |sin phi|) of the block centre; phi is
// within asin((W + 1/2) sqrt(2) / R) =: D0 of the block's own direction, and |cos| + |sin| changes by at most sqrt(2) D0 over
// that range.  Line angles are unwrapped at `wrap`, the direction perpendicular to the tile centre.
__device__ __forceinline__ void arc_block_window(int X0, int Y0, float W, float wrap, float &tlo, float &thi)
{
    const float Xc = (float)X0 + 0.5f, Yc = (float)Y0 + 0.5f;
    float T = atan2f(Yc, Xc);
    T -= floorf(T / kPi) * kPi;
    if (T < wrap) T += kPi;
    const float R = sqrtf(Xc * Xc + Yc * Yc);
    const float sd0 = (W + 0.52f) * 1.41421356f / R;
    const float wcs = fminf(1.41421356f, (fabsf(Xc) + fabsf(Yc)) / R + 1.5f * sd0);
    const float sd = (W + 0.52f) * wcs / R;
    const float D = sd < 0.999f ? asinf(sd) + 2e-3f : 4.0f;
    tlo = T - D;
    thi = T + D;
}

// ------------------------------------------------------------------------------------------------------------------------
// Plan-time pass: the run of every (window, tile), dealt into combs, and every thread's window of it.  grid = (tiles, windows), block = 256.
struct ArcPrepLds {
    unsigned short wb[1024];               // flat tables: the member that holds record 64 w of the run (w < 1 024: 65 536 records)
    int phi2[kArcMaxSpokes];               // flat tables: every entry's record offset, 32 bits
    unsigned s_a[kArcMaxSpokes];
    unsigned s_b[kArcMaxSpokes];
    float2 s_cs[kArcMaxSpokes];
    float phi[kArcMaxSpokes];
    int bstart[kArcMaxBatches + 1];
    int wsum[8];
    int umin, umax, total, kfail, base;
};

__global__ void __launch_bounds__(kArcThreads)
arc_prep_kernel(const ArcPrepParams p)
{
    __shared__ ArcPrepLds L;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int tile = blockIdx.x;
    const int w = blockIdx.y;
    const int n = p.nxos, h = n / 2, rmax = n / 2 - 1;
    const int T = p.flat && p.tile > 0 ? p.tile : kArcTile;     // (grid_scatter_kernel's tables may be made for 64 x 64 tiles)
    const int tpr = n / T;
    const int x0 = (tile % tpr) * T - h, y0 = (tile / tpr) * T - h;
    const bool outer = (x0 == 0 || x0 == -T) && (y0 == 0 || y0 == -T);
    const unsigned short *order = p.order + (size_t)w * p.npe;
    const float *phis = p.phi + (size_t)w * p.npe;
    const float2 *sorted_cs = p.cs + (size_t)w * p.npe;
    int4 *hdr = p.hdr + (size_t)w * p.ntiles + tile;

    // line angles: the run is unwrapped at the direction perpendicular to the tile centre (no spoke of the run is near it)
    float wrap = atan2f((float)y0 + 0.5f * (float)(T - 1), (float)x0 + 0.5f * (float)(T - 1)) + 0.5f * kPi;
    wrap -= floorf(wrap / kPi) * kPi;
    const float eps = 0.01f;
    const float bx_lo = (float)x0 - p.W - eps, bx_hi = (float)(x0 + T - 1) + p.W + eps;
    const float by_lo = (float)y0 - p.W - eps, by_hi = (float)(y0 + T - 1) + p.W + eps;
    const int seg_max = T > kArcTile ? 127 : kArcSeg;           // longest segment through tile + halo (7 bits; 64-tiles: (64 + 6) sqrt(2) = 99)

    if (tid == 0) { L.umin = 1 << 30; L.umax = -1; L.total = 0; L.kfail = 0; L.base = 0; }
    __syncthreads();

    // ---- clip: thread = spoke of the angle-sorted list -----------------------------------------------------------
    constexpr int NCH = kArcMaxNpe / kArcThreads;
    int c_src[NCH], c_pos[NCH], c_seg[NCH];
    float c_phi[NCH];
    float2 c_cs[NCH];
    {
        int lmin = 1 << 30, lmax = -1, lsum = 0;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int j = tid + k * kArcThreads;
            c_pos[k] = -1;
            c_seg[k] = 0;
            c_src[k] = 0;
            c_phi[k] = 0.f;
            c_cs[k] = make_float2(1.f, 0.f);
            if (j < p.npe) {
                const int pe = order[j];
                const float ph = phis[j];
                const float2 cs = sorted_cs[j];
                const float ic = safe_rcp(cs.x), is = safe_rcp(cs.y);
                const float xa = bx_lo * ic, xb = bx_hi * ic;
                const float ya = by_lo * is, yb = by_hi * is;
                const float lo = fmaxf(fmaxf(fminf(xa, xb), fminf(ya, yb)), -(float)rmax);
                const float hi = fminf(fminf(fmaxf(xa, xb), fmaxf(ya, yb)), (float)rmax);
                int len = 0, ulo = 0, neg = 0;
                if (lo <= hi) {
                    int rlo = (int)ceilf(lo);
                    int rhi = (int)floorf(hi);
                    if (outer) {
                        // a quadrant tile holds one side of a spoke plus at most W sqrt(2) < inner_r0 beyond the origin
                        if (rhi >= p.inner_r0) {
                            if (rlo <= -p.inner_r0) atomicOr(p.errflag, 16u);
                            rlo = max(rlo, p.inner_r0);
                        } else if (rlo <= -p.inner_r0) {
                            rhi = min(rhi, -p.inner_r0);
                        } else {
                            rhi = rlo - 1;
                        }
                    }
                    if (rhi >= rlo) {
                        if (rlo <= 0 && rhi >= 0) {             // r = 0 never belongs to a tile of this kernel
                            atomicOr(p.errflag, 32u);
                        } else {
                            neg = rhi < 0;
                            ulo = neg ? -rhi : rlo;
                            len = rhi - rlo + 1;
                            if (len > seg_max) { atomicOr(p.errflag, 2u); len = seg_max; }
                        }
                    }
                }
                const bool wrapped = ph < wrap;
                c_pos[k] = j + (wrapped ? p.npe : 0);
                c_phi[k] = ph + (wrapped ? kPi : 0.f);
                // sample of radius r on spoke pe: nudata[nchan * (nro * pe + r + nro / 2) + c]   src/tron.cu:517,519 (nro == nxos)
                // nro != nxos: radius r reads sample nro / 2 + (r nro) / nxos (:517, truncating towards zero): the entry then names the
                // spoke's centre sample and the copy works out every radius' own sample (arc_sample_of)
                c_src[k] = (int)((unsigned)(p.nro * pe + p.nro / 2 + (p.nro == p.nxos ? (neg ? -ulo : ulo) : 0)) | ((unsigned)neg << 31));
                c_seg[k] = ulo | (len << 10);
                c_cs[k] = neg ? make_float2(-cs.x, -cs.y) : cs;
                if (len > 0) {
                    lmin = min(lmin, c_pos[k]);
                    lmax = max(lmax, c_pos[k]);
                    lsum += len;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lmin = min(lmin, __shfl_xor(lmin, o));
            lmax = max(lmax, __shfl_xor(lmax, o));
            lsum += __shfl_xor(lsum, o);
        }
        if (lane == 0 && lmax >= 0) {
            atomicMin(&L.umin, lmin);
            atomicMax(&L.umax, lmax);
            atomicAdd(&L.total, lsum);
        }
    }
    __syncthreads();
    const int umin = L.umin, umax = L.umax, total = L.total;
    int ns = umax >= umin ? umax - umin + 1 : 0;
    if (ns > kArcMaxSpokes) {
        if (tid == 0) atomicOr(p.errflag, 64u);
        ns = kArcMaxSpokes;
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int idx = c_pos[k] - umin;
        if (c_pos[k] >= 0 && idx >= 0 && idx < ns) {
            L.s_a[idx] = (unsigned)c_src[k];
            L.s_b[idx] = (unsigned)c_seg[k];
            L.s_cs[idx] = c_cs[k];
            L.phi[idx] = c_phi[k];
        }
    }

    // ---- deal the run into K combs; record offsets inside each comb (an exclusive scan in comb-major order) ----------
    int K = max(1, (total + p.nrec - 1) / p.nrec);
    for (;;) {
        __syncthreads();                                    // run entries written (first pass) / kfail cleared (later passes)
        K = min(K, kArcMaxBatches);
        const int M = (ns + K - 1) / K;                     // members of the longest comb
        const int QT = K * M;
        const int QPT = (QT + kArcThreads - 1) / kArcThreads;   // <= 3
        int e_i[3], e_b[3], e_len[3], e_first[3];
        int tsum = 0;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const int q = tid * QPT + e;
            e_i[e] = -1; e_b[e] = 0; e_len[e] = 0; e_first[e] = 0;
            if (e < QPT && q < QT) {
                const int b = q / M, m = q - b * M;
                const int i = m * K + b;
                e_b[e] = b;
                e_first[e] = m == 0;
                if (i < ns) {
                    e_i[e] = i;
                    e_len[e] = (int)((L.s_b[i] >> 10) & 127u);
                }
                tsum += e_len[e];
            }
        }
        int v = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
        if (lane == 63) L.wsum[wave] = v;
        __syncthreads();
        int run = v - tsum;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww)
            if (ww < wave) run += L.wsum[ww];
        int e_excl[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            e_excl[e] = run;
            if (e_first[e]) L.bstart[e_b[e]] = run;
            run += e_len[e];
        }
        if (tid == 0) L.bstart[K] = total;
        __syncthreads();
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            if (e_i[e] >= 0) {
                const int off = e_excl[e] - L.bstart[e_b[e]];
                L.s_b[e_i[e]] = (L.s_b[e_i[e]] & 0x1ffffu) | ((unsigned)(off & 0x7fff) << 17);
                if (p.flat) L.phi2[e_i[e]] = off;               // (one batch: the offset can exceed the field's 15 bits; kept whole beside it)
            }
            if (e_first[e] && L.bstart[e_b[e] + 1] - L.bstart[e_b[e]] > p.nrec) bad = true;
        }
        if (bad) L.kfail = 1;
        __syncthreads();
        if (!L.kfail) break;
        if (K >= kArcMaxBatches) {                          // cannot happen for npe <= kArcMaxNpe (grid_arc_supported)
            if (tid == 0) atomicOr(p.errflag, 256u);
            break;
        }
        __syncthreads();
        if (tid == 0) L.kfail = 0;
        ++K;                                                // a comb overflowed its share: one more comb
    }

    // ---- publish ----
    if (tid == 0) {
        int base = ns > 0 ? atomicAdd(p.alloc + w, ns) : 0;
        if (base + ns > p.cap) {
            atomicOr(p.errflag, 512u);
            base = 0;
            L.base = -1;
        } else {
            L.base = base;
        }
        if (!p.flat) *hdr = make_int4(L.base < 0 ? 0 : ns, K, base, total);
        L.umax = 0;                                          // (reused below: the longest window of the tile's blocks, p.flat)
    }
    __syncthreads();
    const int base = L.base;
    if (base < 0) {
        if (p.flat && tid == 0) *hdr = make_int4(0, 0, 0, 0);
        return;
    }
    uint4 *ent = p.ent + (size_t)w * p.cap + base;
    for (int i = tid; i < ns; i += kArcThreads) {
        const float2 cs = L.s_cs[i];
        ent[i] = make_uint4(L.s_a[i], L.s_b[i], __float_as_uint(cs.x), __float_as_uint(cs.y));
        if (p.flat && p.off) p.off[(size_t)w * p.cap + base + i] = (uint32_t)L.phi2[i];
    }
    // ---- flat tables: which member holds each record (grid_scatter_kernel looks it up instead of searching the run) ----
    // Per GROUP of 64 records 80 bytes: 64 x (the record's member minus the member that holds the group's first record), that member
    // (16 bits), 14 bytes of padding -- +1.25 bytes per 8-byte sample in HBM; the kernel copies a round's groups to LDS by one LDS-DMA
    // instruction a round ahead and has no walk over the segments left.
    if (p.flat && p.rec) {
        const int ngrp = (total + 63) >> 6;
        if (tid == 0) {
            int rb = 0;
            if (total > 0) {
                rb = atomicAdd(p.ralloc + w, ngrp);
                if (rb + ngrp > p.rec_cap || total > 65536) { atomicOr(p.errflag, 2048u); rb = -1; }
            }
            L.kfail = rb;
            p.rbase[(size_t)w * p.ntiles + tile] = rb < 0 ? 0 : rb;
        }
        __syncthreads();
        const int rb = L.kfail;
        if (rb >= 0 && total > 0) {
            for (int i = tid; i < ns; i += kArcThreads) {
                const int len = (int)((L.s_b[i] >> 10) & 127u), off = L.phi2[i];
                if (len > 0)
                    for (int g = (off + 63) >> 6; (g << 6) < off + len; ++g) L.wb[g] = (unsigned short)i;
            }
            __syncthreads();
            unsigned char *const rec = p.rec + ((size_t)w * p.rec_cap + rb) * 80;
            for (int g = tid; g < ngrp; g += kArcThreads) *reinterpret_cast<unsigned short *>(rec + (size_t)g * 80 + 64) = L.wb[g];
            for (int i = tid; i < ns; i += kArcThreads) {
                const int len = (int)((L.s_b[i] >> 10) & 127u), off = L.phi2[i];
                for (int k = 0; k < len; ++k) {
                    const int pos = off + k, d = i - (int)L.wb[pos >> 6];
                    if (d < 0 || d > 255) atomicOr(p.errflag, 4096u);
                    rec[(size_t)(pos >> 6) * 80 + (pos & 63)] = (unsigned char)d;
                }
            }
        }
        __syncthreads();
    }
    // ---- every thread's own run [jlo, jhi] of the list (thread = 2x2 block, as in grid_arc_kernel) ----
    int mw_all = 0;
    const int bpr = T / 2, nblocks = bpr * bpr;                 // 2x2 blocks per tile row / per tile (256; 1 024 for 64-tiles: four per thread)
    for (int blk = tid; blk < nblocks; blk += kArcThreads) {
        const int X0 = x0 + 2 * (blk % bpr), Y0 = y0 + 2 * (blk / bpr);     // (32-tiles: blk = tid = the arc kernel's thread layout)
        int bandlo = 1 << 20, bandhi = -1;                      // radial band of the block's points, src/tron.cu:498-502
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int X = X0 + (q & 1), Y = Y0 + (q >> 1);
            if (X + h < n && Y + h < n) {
                const uint32_t bnd = p.band[(size_t)(Y + h) * n + (X + h)];
                const int lo = (int)(bnd & 0xffffu), hi = (int)(bnd >> 16);
                if (lo <= hi) { bandlo = min(bandlo, lo); bandhi = max(bandhi, hi); }
            }
        }
        const bool has_work = bandlo <= bandhi && bandhi >= (outer ? max(p.inner_r0, 1) : 1);
        int jlo = 1, jhi = 0;
        if (has_work && ns > 0) {
            float tlo, thi;
            arc_block_window(X0, Y0, p.W, wrap, tlo, thi);
            int lo = 0, cnt = ns;
            while (cnt > 0) {
                const int step = cnt >> 1;
                if (L.phi[lo + step] < tlo) { lo += step + 1; cnt -= step + 1; } else cnt = step;
            }
            jlo = lo;
            lo = 0; cnt = ns;
            while (cnt > 0) {
                const int step = cnt >> 1;
                if (!(thi < L.phi[lo + step])) { lo += step + 1; cnt -= step + 1; } else cnt = step;
            }
            jhi = lo - 1;
        }
        if (!p.flat) p.win[((size_t)w * p.ntiles + tile) * kArcThreads + tid] = (uint32_t)jlo | ((uint32_t)(jhi + 1) << 16);
        mw_all = max(mw_all, jhi - jlo + 1);
    }
    if (p.flat) {
        // grid_scatter_kernel (tron_grid_scatter.hip): ONE batch per run (the record offsets then number the run's samples) and, in
        // the header's second word, the most spokes whose line can pass one of the tile's blocks -- the bound behind its fixed-point scale
        int mw = max(mw_all, 0);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mw = max(mw, __shfl_xor(mw, o));
        if (lane == 0) atomicMax(&L.umax, mw);
        __syncthreads();
        if (tid == 0) {
            if (K != 1) atomicOr(p.errflag, 1024u);
            *hdr = make_int4(ns, L.umax, base, total);
        }
    }
}

hipError_t launch_arc_prep(const ArcPrepParams &p, int nwindows, hipStream_t s)
{
    const int T = p.flat && p.tile > 0 ? p.tile : kArcTile;
    if (p.npe > kArcMaxNpe || p.nrec < kArcSeg || (T != 32 && T != 64) || (p.nxos / 2) % T != 0 || p.ntiles != (p.nxos / T) * (p.nxos / T)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(arc_prep_kernel, dim3((unsigned)p.ntiles, (unsigned)nwindows), dim3(kArcThreads), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------------
// HALF: k-space stored as complex-half.  The copy brings the halves into the upper half of each record's fp32 slots (four
// coils per 16 bytes) and one pass over the records converts them in place -- each thread reads all of a record's halves,
// then writes its floats, so no record is touched by two threads -- and the gather is the fp32 one.
bool arc_resample_exact(int nxos, int nro)
{
    const float nro_f = (float)nro, inv = 1.0f / (float)nxos;
    for (int u = 0; u < nxos / 2; ++u)
        if ((int)arc_sample_of((float)u, nro_f, inv) != (int)(((long long)u * nro) / nxos)) return false;
    return true;
}

// RS: nro != nxos, the truncating resample of the readout (src/tron.cu:517, 526): the copy fetches every radius' own sample (one
// record per radius as ever, so the gather indexes as before) and the density compensation follows the sample, not the radius.
template <int CPB, bool HALF, bool RS>
__global__ void __launch_bounds__(kArcThreads, ArcCfg<CPB>::WAVES)
grid_arc_kernel(const GridParams p)
{
    using C = ArcCfg<CPB>;
    static_assert(CPB == 1 || CPB % 2 == 0, "the samples are copied as 16-byte coil pairs (or one coil as two 4-byte planes)");
    static_assert(!HALF || CPB >= 4, "complex-half samples are copied four coils at a time (the last piece of 6 coils holds coils 2..5)");
    static_assert(arc_lds_fits<CPB>(), "ArcCfg::WAVES workgroups of ArcLds do not fit a CU's 128 LDS units of 1280 bytes");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ArcLds<CPB> &L = *reinterpret_cast<ArcLds<CPB> *>(lds_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int zper = p.arc_zper > 0 ? p.arc_zper : 1;
    const int ngroups = (p.nslices + zper - 1) / zper;
    const int zg = blockIdx.x % ngroups;
    const int tile = p.tile_order[blockIdx.x / ngroups] & 0xffff;
    if (tile >= p.ntiles) {
        if (tid == 0) atomicOr(p.errflag, 128u);
        return;
    }
    const int c0 = p.coil0 + blockIdx.y * CPB;
    const int ncb = min(CPB, p.nchan - c0);
    const int n = p.nxos;
    const int h = n / 2;
    const int rmax = n / 2 - 1;
    const int x0 = (tile % p.tiles_per_row) * kArcTile - h;     // tile origin, centred coordinates
    const int y0 = (tile / p.tiles_per_row) * kArcTile - h;
    // the four tiles that meet at the k-space centre leave |r| < inner_r0 to the inner tile (tron_grid_binned.hip)
    const bool outer = (x0 == 0 || x0 == -kArcTile) && (y0 == 0 || y0 == -kArcTile);
    if (p.skip_outside) {
        // nearest point of the tile to the k-space centre; beyond rmax + W every band is empty (src/tron.cu:498-502,512)
        const int ax = max(max(x0, -(x0 + kArcTile - 1)), 0), ay = max(max(y0, -(y0 + kArcTile - 1)), 0);
        const float lim = (float)rmax + p.W + 1.0f;
        if ((float)(ax * ax + ay * ay) > lim * lim) return;
    }
    const int mx = 2 * (lane & 15);                             // this thread's 2x2 points, tile-relative
    const int my = 8 * wave + 2 * (lane >> 4);
    const int X0 = x0 + mx, Y0 = y0 + my;
    int Rlo[4], Rhi[4];                                         // radial band per point, src/tron.cu:498-502
    int bandlo = 1 << 20, bandhi = -1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int X = X0 + (q & 1), Y = Y0 + (q >> 1);
        Rlo[q] = 1; Rhi[q] = 0;
        if (X + h < n && Y + h < n) {
            const uint32_t bnd = p.band[(size_t)(Y + h) * n + (X + h)];
            const int lo = (int)(bnd & 0xffffu), hi = (int)(bnd >> 16);
            if (lo <= hi) { Rlo[q] = lo; Rhi[q] = hi; bandlo = min(bandlo, lo); bandhi = max(bandhi, hi); }
        }
    }
    // the bands as bit masks over u - bandlo (the four bands of a 2x2 block span at most 2W + 3 radii)
    unsigned bmask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bmask[q] = 0u;
        if (Rlo[q] <= Rhi[q] && Rhi[q] - bandlo < 31) bmask[q] = (0xffffffffu >> (31 - (Rhi[q] - Rlo[q]))) << (Rlo[q] - bandlo);
    }
    const int umin_tile = outer ? max(p.inner_r0, 1) : 1;
    const float blo_f = (float)max(bandlo, umin_tile), bhi_f = (float)bandhi;

    const float X0f = (float)X0, X1f = (float)(X0 + 1), Y0f = (float)Y0, Y1f = (float)(Y0 + 1);
    const v2f p0v = {X0f, Y0f}, lscale2 = {p.lut_scale, p.lut_scale};
    const float We = p.W + 1e-3f;
    const float xlo = X0f - We, xhi = X1f + We, ylo = Y0f - We, yhi = Y1f + We;

    for (int i = tid; i < 3 * kArcLutEntries; i += kArcThreads) L.lut[i] = p.kb_lut[i];
    const lds_f2p lutq = (lds_f2p)(__attribute__((address_space(3))) const void *)L.lut + p.lut_bias;      // entry of table position 0
    const unsigned dbase = lds_addr(L.d);
    constexpr unsigned kBufBytes = (unsigned)(C::NREC * CPB * 8);
    // complex-half: pieces of four coils (16 bytes) wait in the last coil-pair planes of a record's fp32 slots until they are converted in place
    constexpr int kHalfPieces = (CPB + 3) / 4, kHalfPlane0 = CPB / 2 - kHalfPieces;
    constexpr unsigned kRecStep = CPB == 1 ? 4u : 16u;              // bytes between consecutive records of one plane
    const float dcf_a = p.apply_dcf ? p.dcf_a : 0.0f, dcf_b = p.apply_dcf ? p.dcf_b : 1.0f;
    const float rs_nro = (float)p.nro, rs_inv = 1.0f / (float)p.nxos;

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned nchan8 = (unsigned)p.nchan * (HALF ? 4u : 8u);       // bytes per sample (all coils)
    const unsigned lane_step = (unsigned)lane * nchan8;
    // output: the block's two rows as byte offsets inside a coil plane (the tile lies inside the grid, its columns are even)
    unsigned out_off[2];
#pragma unroll
    for (int qy = 0; qy < 2; ++qy) {
        const int Y = Y0 + qy;
        const int row = p.out_shift ? (Y < 0 ? Y + n : Y) : Y + h;     // both fftshifts of src/tron.cu:631 folded in
        const int col = p.out_shift ? (X0 < 0 ? X0 + n : X0) : X0 + h;
        out_off[qy] = (unsigned)(row * n + col) * 8u;
    }

    APROF_DECL;
    // The run table of a slice (<= 512 entries, two per thread) and this thread's window of it are asked for one slice ahead and
    // wait in registers: they used to be loaded when the slice began (6 % of the wave cycles, all latency).
    uint4 pf_ent[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    uint32_t pf_wj = 1u << 16;                                  // jlo = 0, jhi = 0 - ... (overwritten below)
    int4 hdr_next = make_int4(0, 1, 0, 0);
    auto fetch_table = [&](const int z, const int4 h) {
        const size_t win = (size_t)z * p.arc_slice_stride;
        const uint4 *ent = p.arc_ent + win * p.arc_cap + h.z;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (tid + k * kArcThreads < h.x) pf_ent[k] = ent[tid + k * kArcThreads];
        pf_wj = p.arc_win[(win * p.ntiles + tile) * kArcThreads + tid];                 // this thread's run of the list: arc_prep_kernel
    };
    if (zg * zper < p.nslices) {
        hdr_next = p.arc_hdr[(size_t)(zg * zper) * p.arc_slice_stride * p.ntiles + tile];
        fetch_table(zg * zper, hdr_next);
    }
    for (int iz = 0; iz < zper; ++iz) {
        const int z = zg * zper + iz;
        if (z >= p.nslices) break;
        const int4 hdr = hdr_next;
        const int ns = hdr.x, K = hdr.y;
        const bool more = iz + 1 < zper && z + 1 < p.nslices;
        if (more) hdr_next = p.arc_hdr[(size_t)(z + 1) * p.arc_slice_stride * p.ntiles + tile];   // (wave-uniform: a scalar load, back long before it is used)
        const uint32_t wj = pf_wj;
        const int jlo = (int)(wj & 0xffffu), jhi = (int)(wj >> 16) - 1;
        const unsigned char *in = reinterpret_cast<const unsigned char *>(p.nudata) + ((size_t)z * (size_t)p.in_slice_stride + c0) * (HALF ? 4 : 8);

        v2f acc[4][CPB];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < CPB; ++c) acc[q][c] = (v2f){0.f, 0.f};

        __syncthreads();                                        // the last slice's gather has ended: run table and buffers are free
        // ---- the tile's run -> LDS ----
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * kArcThreads;
            if (i < ns) {
                const uint4 e = pf_ent[k];
                L.s_a[i] = e.x;
                L.s_b[i] = e.y;
                L.s_cs[i] = make_float2(__uint_as_float(e.z), __uint_as_float(e.w));
            }
        }
        APROF_MARK(0);                                          // tile setup, run table
        __syncthreads();
        APROF_MARK(1);

        // samples of batch b -> buffer b & 1: one LDS-DMA instruction per member and coil pair, lane = radius.  Everything but
        // the lane's own byte offset is wave-uniform and kept in scalar registers.
        auto issue = [&](const int b) {
            const int nmem = (ns - b + K - 1) / K;
            const unsigned buf = dbase + (unsigned)(b & (C::NBUF - 1)) * kBufBytes;
            // the members' run entries 64 at a time, one per lane (ONE round trip to LDS instead of one per member: the copies
            // used to be issued at the pace of those reads), then by v_readlane
            for (int mb = 0; mb < nmem; mb += 64) {
                unsigned va = 0u, vsb = 0u;
                if (mb + lane < nmem) {
                    va = L.s_a[b + K * (mb + lane)];
                    vsb = L.s_b[b + K * (mb + lane)];
                }
                const int mend = min(nmem - mb, 64);
                for (int m = wave_u; m < mend; m += 4) {
                    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)va, m);
                    const unsigned sb = (unsigned)__builtin_amdgcn_readlane((int)vsb, m);
                    const int len = (int)((sb >> 10) & 127u);
                    const unsigned dst = buf + (sb >> 17) * kRecStep;
                    const unsigned first = (a & 0x7fffffffu) * nchan8;                     // byte offset of the first record's coil 0
                    if (lane < len) {
                        unsigned voff;
                        if constexpr (RS) {                                                    // `first` is the spoke's centre sample here
                            const unsigned so = (unsigned)(int)arc_sample_of((float)((int)(sb & 1023u) + lane), rs_nro, rs_inv) * nchan8;
                            voff = (a >> 31) ? first - so : first + so;
                        } else {
                            voff = (a >> 31) ? first - lane_step : first + lane_step;
                        }
                        if constexpr (CPB == 1) {
                            lds_dma4_s(in, voff, dst);                                     // real parts, imaginary parts
                            lds_dma4_s(in + 4, voff, dst + (unsigned)(C::NREC * 4));
                        } else if constexpr (HALF) {
#pragma unroll
                            for (int hq = 0; hq < kHalfPieces; ++hq)                       // four coils per piece, parked in the LAST coil-pair planes
                                // (a piece with only two coils left -- channel counts 6, 10, ...: 24-byte records -- is read 8 bytes early, coils
                                // 4 hq - 2 .. 4 hq + 1: inside the record, where a piece read straight would run 8 bytes past the last sample of the
                                // buffer; the LDS-DMA path takes 4-byte-aligned sources, tools/probe/dma_align.hip)
                                if (4 * hq < ncb) lds_dma16_s(in + 16 * hq - (4 * hq + 4 > ncb ? 8 : 0), voff, dst + (unsigned)((kHalfPlane0 + hq) * C::NREC * 16));
                        } else {
#pragma unroll
                            for (int c = 0; c < CPB / 2; ++c)
                                if (2 * c < ncb) lds_dma16_s(in + 16 * c, voff, dst + (unsigned)(c * C::NREC * 16));
                        }
                    }
                }
            }
        };
        APROF_MARK(2);
        APROF_MARK(3);
        const float rcpK = 1.0f / (float)K;
        for (int b = 0; b < K && ns > 0; ++b) {
            if (C::NBUF == 1) {
                if (b > 0) lds_barrier();                       // everyone has left the buffer
                issue(b);
            } else if (b == 0) {
                issue(0);
            }
            APROF_MARK(2);                                      // DMA issue
            // ---- this thread's members of the comb, and the first four of them clipped while the copy flies (the clip needs the
            // run table only) ----
            int mlo = 0, mhi = -1;
            if (jhi >= jlo && jhi >= b) {
                mlo = jlo <= b ? 0 : (int)(((float)(jlo - b + K - 1) + 0.5f) * rcpK);
                mhi = (int)(((float)(jhi - b) + 0.5f) * rcpK);
            }
            // member -> descriptor: radii << 28 | first record << 17 | run entry << 8 | first radius - bandlo (0: no visit)
            auto clip = [&](const int m) -> unsigned {
                unsigned dsc = 0u;
                if (m <= mhi) {
#ifdef TRON_PHASE_CLOCK
                    { const unsigned long long bm_ = __ballot(1); if (lane == __builtin_ctzll(bm_)) { APROF_COUNT(12, 1); APROF_COUNT(13, __popcll(bm_)); } }
#endif
                    const int i = b + K * m;
                    const float2 cs = L.s_cs[i];
                    const unsigned sb = L.s_b[i];
                    const int s_ulo = (int)(sb & 1023u), s_len = (int)((sb >> 10) & 127u);
                    // 1 / 0 = inf is fine here: the box edges are never 0 (W + 1e-3 is no integer), inf clips like a huge number
                    const float ic = __builtin_amdgcn_rcpf(cs.x), is = __builtin_amdgcn_rcpf(cs.y);
                    // the radii whose sample lies inside the block's footprint: x0 - W < u cos < x1 + W, likewise y  (src/tron.cu:514-516)
                    const float xa = xlo * ic, xb = xhi * ic;
                    const float ya = ylo * is, yb = yhi * is;
                    const float lo = fmaxf(fmaxf(fminf(xa, xb), fminf(ya, yb)), fmaxf((float)s_ulo, blo_f));
                    const float hi = fminf(fminf(fmaxf(xa, xb), fmaxf(ya, yb)), fminf((float)(s_ulo + s_len - 1), bhi_f));
                    const int ua = (int)ceilf(lo), ub = (int)floorf(hi);
                    if (ua <= ub)
                        dsc = ((unsigned)(ub - ua + 1) << 28) | ((unsigned)((int)(sb >> 17) - s_ulo + ua) << 17) | ((unsigned)i << 8) | (unsigned)(ua - bandlo);
                }
                return dsc;
            };
            int m0 = mlo;
            const bool any_member = __ballot(m0 <= mhi) != 0ull;
            unsigned desc[4] = {0u, 0u, 0u, 0u};
            if (any_member) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm volatile("" ::: "memory");              // one clip after the other: four at once spill accumulators
                    desc[k] = clip(m0 + k);
                }
            }
            APROF_MARK(6);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of batch b have landed
            APROF_MARK(4);                                      // DMA wait
            lds_barrier();                                      // ... everyone's have; everyone has left batch b - 1's buffer
            if constexpr (HALF) {
                // complex-half -> fp32 in place, record by record (slots beyond the batch's records hold stale bits: never read)
                const unsigned cbuf = dbase + (unsigned)(b & (C::NBUF - 1)) * kBufBytes;
                for (int rec = tid; rec < C::NREC; rec += kArcThreads) {
                    v4u hv[kHalfPieces];
#pragma unroll
                    for (int hq = 0; hq < kHalfPieces; ++hq)
                        hv[hq] = *(const __attribute__((address_space(3))) v4u *)(size_t)(cbuf + (unsigned)((kHalfPlane0 + hq) * C::NREC * 16 + rec * 16));
#pragma unroll
                    for (int hq = 0; hq < kHalfPieces; ++hq) {
                        const bool early = 4 * hq < ncb && 4 * hq + 4 > ncb;                   // (the piece was read 8 bytes early: its coils are words 2, 3)
                        const unsigned w[4] = {early ? hv[hq].z : hv[hq].x, early ? hv[hq].w : hv[hq].y, hv[hq].z, hv[hq].w};
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            if (2 * hq + k >= CPB / 2) continue;                               // (six coils: the second piece fills one plane)
                            __half2 h0, h1;
                            __builtin_memcpy(&h0, &w[2 * k], 4);
                            __builtin_memcpy(&h1, &w[2 * k + 1], 4);
                            const float2 f0 = __half22float2(h0), f1 = __half22float2(h1);
                            *(__attribute__((address_space(3))) v4f *)(size_t)(cbuf + (unsigned)((2 * hq + k) * C::NREC * 16 + rec * 16)) = (v4f){f0.x, f0.y, f1.x, f1.y};
                        }
                    }
                }
                lds_barrier();
            }
            APROF_MARK(5);
            if (C::NBUF == 2 && b + 1 < K) issue(b + 1);
            APROF_MARK(2);
            const unsigned buf = dbase + (unsigned)(b & (C::NBUF - 1)) * kBufBytes;

            if (b == 0 && more) fetch_table(z + 1, hdr_next);   // the next slice's table and window: on their way while this slice is gathered

            // ---- gather ----
            // Members four at a time: clipped (above / at the bottom of this loop), sorted descending by their radii count, visited.
            // Simulated on the metric trajectory (tools/probe/arc_util_sim.py, which reproduces the kernel's wave-level counters):
            // radius loop 36.0 k -> 30.4 k wave iterations per slice, lanes active 0.51 -> 0.60; a flat loop over a lane's visits
            // would reach 0.69 but pays a member switch inside the hot loop (measured in round 3: slower).
            // (bottom-tested: hipcc keeps two copies of the accumulators across a top-tested loop with a wave-uniform exit)
            if (any_member) do {
#define TRON_ARC_CSWAP(a, b) { const unsigned hi_ = max(desc[a], desc[b]), lo_ = min(desc[a], desc[b]); desc[a] = hi_; desc[b] = lo_; }
                TRON_ARC_CSWAP(0, 1) TRON_ARC_CSWAP(2, 3) TRON_ARC_CSWAP(0, 2) TRON_ARC_CSWAP(1, 3) TRON_ARC_CSWAP(1, 2)
#undef TRON_ARC_CSWAP
#pragma unroll 1
                for (int k = 0; k < 4; ++k) {                                                 // (one copy of the radius loop; k is wave-uniform)
                    const unsigned dk = k == 0 ? desc[0] : (k == 1 ? desc[1] : (k == 2 ? desc[2] : desc[3]));
                    if (dk != 0u) {                                                           // (no lane left: the compiler's execz branch skips the body)
                        const float2 cs = L.s_cs[(dk >> 8) & 511u];
                        unsigned addr = buf + ((dk >> 17) & 2047u) * kRecStep;
                        int bit = (int)(dk & 255u);
                        const int bit_end = bit + (int)(dk >> 28);
                        const float uf0 = (float)(bit + bandlo);
                        v2f ufv = {uf0, uf0};
                        const v2f csv = {cs.x, cs.y};
                        do {
#ifdef TRON_PHASE_CLOCK
                            { const unsigned long long bm_ = __ballot(1); if (lane == __builtin_ctzll(bm_)) { APROF_COUNT(14, 1); APROF_COUNT(15, __popcll(bm_)); } }
#endif
                            // (kx, ky) = u (cos, sin) and the distances to the block's first column / row, op for op src/tron.cu:514-516;
                            // their table positions as a packed pair: one position serves both columns (rows) of the block
                            const v2f kxy = ufv * csv;
                            const v2f tp = (kxy - p0v) * lscale2;                                      // exact: the scale is a power of two
                            const v2f tt = {__builtin_truncf(tp.x), __builtin_truncf(tp.y)};
                            const v2f fv = tp - tt;                                                    // in (-1, 1)
                            const lds_f2p lx = lutq + (int)tt.x, ly = lutq + (int)tt.y;
                            const v2f x0c = lx[0], x1c = lx[kArcLutEntries], x2c = lx[2 * kArcLutEntries];
                            const v2f y0c = ly[0], y1c = ly[kArcLutEntries], y2c = ly[2 * kArcLutEntries];
                            v4f dd[CPB / 2 > 0 ? CPB / 2 : 1];
                            v2f d1 = {0.f, 0.f};
                            if constexpr (CPB == 1) {
                                d1.x = *(const __attribute__((address_space(3))) float *)(size_t)addr;
                                d1.y = *(const __attribute__((address_space(3))) float *)(size_t)(addr + (unsigned)(C::NREC * 4));
                            } else {
#pragma unroll
                                for (int c = 0; c < CPB / 2; ++c) dd[c] = *(lds_f4p)(size_t)(addr + (unsigned)(c * C::NREC * 16));
                            }
                            const float sdc = fmaf(dcf_a, RS ? arc_sample_of(ufv.x, rs_nro, rs_inv) : ufv.x, dcf_b);   // src/tron.cu:412 (|ro - nro/2| = u, or u's sample)
                            const v2f fxv = {fv.x, fv.x}, fyv = {fv.y, fv.y}, sdcv = {sdc, sdc};
                            const v2f wx = __builtin_elementwise_fma(fxv, __builtin_elementwise_fma(fxv, x2c, x1c), x0c);
                            const v2f wy = __builtin_elementwise_fma(fyv, __builtin_elementwise_fma(fyv, y2c, y1c), y0c) * sdcv;
                            const v2f w01 = wx * (v2f){wy.x, wy.x}, w23 = wx * (v2f){wy.y, wy.y};    // src/tron.cu:516
                            float wq[4] = {w01.x, w01.y, w23.x, w23.y};
#pragma unroll
                            for (int q = 0; q < 4; ++q)                                                // src/tron.cu:512,521: bit u - bandlo of the point's band mask
                                wq[q] = __uint_as_float(__float_as_uint(wq[q]) & (unsigned)__builtin_amdgcn_sbfe((int)bmask[q], bit, 1));
                            if constexpr (CPB == 1) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    acc[q][0].x = fmaf(d1.x, wq[q], acc[q][0].x);                      // src/tron.cu:519
                                    acc[q][0].y = fmaf(d1.y, wq[q], acc[q][0].y);
                                }
                            }
#pragma unroll
                            for (int c = 0; c < CPB / 2; ++c) {
                                const v4f d = dd[c];
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    acc[q][2 * c].x = fmaf(d.x, wq[q], acc[q][2 * c].x);              // src/tron.cu:519
                                    acc[q][2 * c].y = fmaf(d.y, wq[q], acc[q][2 * c].y);
                                    acc[q][2 * c + 1].x = fmaf(d.z, wq[q], acc[q][2 * c + 1].x);
                                    acc[q][2 * c + 1].y = fmaf(d.w, wq[q], acc[q][2 * c + 1].y);
                                }
                            }
                            ufv += (v2f){1.0f, 1.0f};
                            addr += kRecStep;
                            ++bit;
                        } while (bit < bit_end);
                    }
                }
                m0 += 4;
                if (__ballot(m0 <= mhi) == 0ull) break;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    asm volatile("" ::: "memory");
                    desc[k] = clip(m0 + k);
                }
            } while (true);
            APROF_MARK(6);                                      // gather
        }
        // an empty run (a rim tile no spoke of this window crosses) has no batch to hide the request behind: without it the next
        // slice of this workgroup would use this slice's (empty) table and window and drop its samples
        if (more && ns <= 0) fetch_table(z + 1, hdr_next);

        {
            unsigned char *zbase = reinterpret_cast<unsigned char *>(p.udata + (size_t)z * p.out_z + (size_t)c0 * p.out_c);
#pragma unroll
            for (int c = 0; c < CPB; ++c)
                if (c < ncb) {
#pragma unroll
                    for (int qy = 0; qy < 2; ++qy) {
                        float4 v;
                        v.x = acc[2 * qy][c].x * p.scale;                   // src/tron.cu:532-534
                        v.y = acc[2 * qy][c].y * p.scale;
                        v.z = acc[2 * qy + 1][c].x * p.scale;
                        v.w = acc[2 * qy + 1][c].y * p.scale;
                        float4 *const o = reinterpret_cast<float4 *>(zbase + (size_t)c * p.out_c * 8 + out_off[qy]);
                        if (p.arc_accumulate) {                             // a later pass over more than kArcMaxNpe spokes per window
                            const float4 old = *o;
                            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                        }
                        *o = v;
                    }
                }
        }
        APROF_MARK(7);                                          // output store
    }
    APROF_FLUSH;
}

#ifdef TRON_PHASE_CLOCK
extern "C" __attribute__((visibility("default"))) int tron_debug_arc_profile(unsigned long long *out, int n)   // reads and clears the phase clock
{
    static unsigned long long h[kArcProfCopies * kArcProfSlots];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_arc_prof), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < n && i < kArcProfSlots; ++i) {
        out[i] = 0;
        for (int c = 0; c < kArcProfCopies; ++c) out[i] += h[c * kArcProfSlots + i];
    }
    for (size_t i = 0; i < sizeof(h) / sizeof(h[0]); ++i) h[i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_arc_prof), h, sizeof(h)) != hipSuccess;
}
#endif

template <int CPB, bool HALF, bool RS>
static hipError_t launch_arc_rs(const GridParams &p, int first_plain, hipStream_t s)
{
    if (p.arc_nrec != ArcCfg<CPB>::NREC) return hipErrorInvalidValue;       // the run tables were dealt for another batch size
    const int tpr = (p.nxos + kArcTile - 1) / kArcTile;
    GridParams q = p;
    q.tiles_per_row = tpr;
    q.ntiles = tpr * tpr;
    q.tile_order = p.tile_order + first_plain;
    q.arc_zper = p.arc_zper > 0 ? p.arc_zper : 1;
    const int ngroups = (p.nslices + q.arc_zper - 1) / q.arc_zper;
    const int chunks = (p.nchan - p.coil0 + CPB - 1) / CPB;
    dim3 grid((unsigned)((size_t)q.ntiles * ngroups), (unsigned)chunks);
    const size_t lds = sizeof(ArcLds<CPB>);
    if (lds > 64 * 1024) {
        const hipError_t once = allow_dynamic_lds(reinterpret_cast<const void *>(grid_arc_kernel<CPB, HALF, RS>), (int)sizeof(ArcLds<CPB>));
        if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL((grid_arc_kernel<CPB, HALF, RS>), grid, dim3(kArcThreads), lds, s, q);
    return hipGetLastError();
}

template <int CPB, bool HALF>
static hipError_t launch_arc_cpb(const GridParams &p, int first_plain, hipStream_t s)
{
    return p.nro != p.nxos ? launch_arc_rs<CPB, HALF, true>(p, first_plain, s) : launch_arc_rs<CPB, HALF, false>(p, first_plain, s);
}

// fp32 k-space: one coil or an even coil count (16-byte coil pairs); complex-half: an even coil count >= 4 (16-byte pieces of four
// coils; the last piece of a count that is no multiple of four is read 8 bytes early)
bool grid_arc_supported(int nchan, int nxos, int nro, int npe, float W, int half_in)
{
    const bool coils = half_in ? (nchan >= 4 && (nchan & 1) == 0) : (nchan == 1 || (nchan & 1) == 0);
    // widths without a Kaiser-Bessel pair table (W <= 1, W 2^k no integer; build_kb_pair_lut) stay on the binned kernel
    // nro != nxos (any -o but 2): the truncating resample of src/tron.cu:517, where its float form is exact (it is for every size tried)
    return nchan >= 1 && coils && (nro == nxos || (nro >= 2 && arc_resample_exact(nxos, nro))) && nxos <= 2048 && npe <= kArcMaxWindow && W <= 3.0f && kb_pair_lut_scale(W, kArcLutEntries) > 0
           && (nxos / 2) % kArcTile == 0 && nxos >= 4 * kArcTile && (long long)nro * npe * nchan < (1ll << 28);
}

// p.tile_order[first_plain ...] must list the 32x32 tiles (see build_tile_order(nxos, 32, ...)); p.inner_r0 > 0.
hipError_t launch_grid_arc(const GridParams &p, int half_in, int first_plain, hipStream_t s)
{
    const int gran = half_in ? 3 : (p.nchan == 1 ? 0 : 1);
    if (p.out_p != 1 || p.inner_r0 <= 0 || !p.arc_hdr || !p.arc_ent || !p.arc_win || !p.kb_lut || p.lut_entries > kArcLutEntries || (p.coil0 & gran)
        || p.npe > kArcMaxNpe || !grid_arc_supported(p.nchan, p.nxos, p.nro, p.npe, p.W, half_in) || (reinterpret_cast<uintptr_t>(p.nudata) & 15) != 0)
        return hipErrorInvalidValue;
    const int nc = p.nchan - p.coil0;
    if (nc >= 5) {
        const int pad8 = (nc + 7) / 8 * 8, pad6 = (nc + 5) / 6 * 6;
        // (complex-half chunks start on a multiple of four coils: 6-coil chunks only for exactly 6 coils)
        if (half_in) return nc == 6 ? launch_arc_cpb<6, true>(p, first_plain, s) : launch_arc_cpb<8, true>(p, first_plain, s);
        return pad6 < pad8 ? launch_arc_cpb<6, false>(p, first_plain, s) : launch_arc_cpb<8, false>(p, first_plain, s);
    }
    if (half_in) return launch_arc_cpb<4, true>(p, first_plain, s);
    if (nc >= 3) return launch_arc_cpb<4, false>(p, first_plain, s);
    if (nc >= 2) return launch_arc_cpb<2, false>(p, first_plain, s);
    return launch_arc_cpb<1, false>(p, first_plain, s);
}

__global__ void warm_grid_arc_tu() {}

hipError_t warm_grid_arc()   // see warm_kernels() in tron_kernels.hip
{
    hipLaunchKernelGGL(warm_grid_arc_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
