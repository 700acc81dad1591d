// Spoke clipping and the sample loop of the tiled degridding kernels (degridradial2d, src/tron.cu:540-577), shared by
// degrid_tile_kernel (one tile and coil chunk per workgroup) and degrid_stream_kernel (one tile, a run of images).
//
// Both work on a 32x32 tile of the Cartesian grid held in LDS with a halo: every spoke is clipped against the tile
// (thread = spoke, ballot compaction), and the samples that fall inside it are dealt out flat over the workgroup's
// threads.  A sample is owned by the tile that holds floor(X), floor(Y), so every sample is produced exactly once, with
// the reference's own coordinate arithmetic, weights and accumulation order (xu outer, yu inner): TRON_KB_EXACT is
// bit-identical to the reference loop.
#pragma once

#include "tron_device.h"

namespace tron {

constexpr int kDgTile = 32;

// spoke lists of one tile and round: MAXSP accepted spokes, MAXB 64-record blocks indexed by the inverse map
template <int MAXSP, int MAXB, int NWAVES>
struct DgLists {
    int sp_pe[MAXSP];
    int sp_seg[MAXSP];             // ro_lo | len << 16
    int sp_start[MAXSP + 1];       // exclusive scan of len
    float2 sp_cs[MAXSP];           // (cos, sin) of the accepted spokes: the sample loop stays off global memory
    int wcnt[2 * NWAVES];
    unsigned short first[MAXB];    // spoke slot holding record 64*b
    alignas(8) unsigned int segbits[2 * MAXB];   // bit r: record r is the first of its spoke's segment (64 records per wave pass: one 8-byte word)
};

struct DgRound {
    int nacc, nrec;
    bool mapped;
};

// Clip spokes [round0, round0 + MAXSP) of image k against tile (tx0, ty0): X(ro) = nr*(ro/nro - 1/2)*sin + halfr,
// Y likewise with cos (src/tron.cu:554-561).  All NT threads call it; ends with the lists visible to all of them.
template <int NT, int MAXSP, int MAXB, class Lists>
__device__ __forceinline__ DgRound dg_clip_round(const DegridParams &p, Lists &L, const int k, const int round0, const int tid,
                                                const int tx0, const int ty0, const int n, const int nr)
{
    constexpr int NW = NT / 64;
    const int lane = tid & 63, wave = tid >> 6;
    const float half = (float)((n + 1) / 2), halfr = (float)((nr + 1) / 2);   // src/tron.cu:560-561
    const float eps = 0.01f;
    const float bx_lo = (float)tx0 - eps, bx_hi = (float)(tx0 + kDgTile) + eps;
    const float by_lo = (float)ty0 - eps, by_hi = (float)(ty0 + kDgTile) + eps;
    if (tid == 0) L.sp_start[0] = 0;
    int nacc = 0;
    for (int chunk0 = round0; chunk0 < min(p.npe, round0 + MAXSP); chunk0 += NT) {
        const int pe = chunk0 + tid;
        bool accept = false;
        int rlo = 0, len = 0;
        float2 cs = make_float2(0.f, 0.f);
        if (pe < p.npe && pe < round0 + MAXSP) {
            cs = p.trig[(size_t)k * p.trig_img_stride + pe];
            const float ax = (float)nr * cs.y / (float)p.nro, ay = (float)n * cs.x / (float)p.nro;  // d/d(ro)
            const float ox = halfr - 0.5f * (float)nr * cs.y, oy = half - 0.5f * (float)n * cs.x;   // value at ro = 0
            const float ix = safe_rcp(ax), iy = safe_rcp(ay);
            const float xa = (bx_lo - ox) * ix, xb = (bx_hi - ox) * ix;
            const float ya = (by_lo - oy) * iy, yb = (by_hi - oy) * iy;
            const float lo = fmaxf(fmaxf(fminf(xa, xb), fminf(ya, yb)) - 1.0f, 0.0f);
            const float hi = fminf(fminf(fmaxf(xa, xb), fmaxf(ya, yb)) + 1.0f, (float)(p.nro - 1));
            if (lo <= hi) {
                rlo = (int)floorf(lo);
                const int rhi = (int)ceilf(hi);
                len = min(rhi, p.nro - 1) - rlo + 1;
                if (len > 0x7fff) len = 0x7fff;
                accept = len > 0;
            }
        }
        const unsigned long long m = __ballot(accept);
        if (lane == 0) L.wcnt[wave] = __popcll(m);
        __syncthreads();
        int base = nacc, total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int cnt = L.wcnt[w];
            if (w < wave) base += cnt;
            total += cnt;
        }
        if (accept) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            L.sp_pe[slot] = pe;
            L.sp_seg[slot] = (rlo & 0xffff) | (len << 16);
            L.sp_cs[slot] = cs;
        }
        nacc += total;
        __syncthreads();
    }
    {   // exclusive scan of the segment lengths, two per thread (2 NT >= MAXSP)
        static_assert(2 * NT >= MAXSP, "two list entries per thread");
        const int i0 = 2 * tid, i1 = 2 * tid + 1;
        const int l0 = i0 < nacc ? (L.sp_seg[i0] >> 16) : 0;
        const int l1 = i1 < nacc ? (L.sp_seg[i1] >> 16) : 0;
        int v = l0 + l1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(v, o);
            if (lane >= o) v += t;
        }
        if (lane == 63) L.wcnt[NW + wave] = v;
        __syncthreads();
        int wbase = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if (w < wave) wbase += L.wcnt[NW + w];
        const int excl = wbase + v - (l0 + l1);
        if (i0 < nacc) L.sp_start[i0 + 1] = excl + l0;
        if (i1 < nacc) L.sp_start[i1 + 1] = excl + l0 + l1;
        __syncthreads();
    }
    DgRound r;
    r.nacc = nacc;
    r.nrec = L.sp_start[nacc];
    r.mapped = r.nrec <= 64 * MAXB;
    if (r.mapped) {
        // inverse map: which spoke holds record 64*b (a wave pass covers exactly one such block), and where the segments
        // start inside the block: a lane's spoke is first[b] + the segment starts at records (64 b, rec]
        for (int w = tid; w < 2 * ((r.nrec + 63) >> 6); w += NT) L.segbits[w] = 0u;
        __syncthreads();
        for (int sidx = tid; sidx < nacc; sidx += NT) {
            const int st = L.sp_start[sidx], en = L.sp_start[sidx + 1];
            for (int b = (st + 63) >> 6; 64 * b < en; ++b) L.first[b] = (unsigned short)sidx;
            atomicOr(&L.segbits[st >> 5], 1u << (st & 31));
        }
        __syncthreads();
    }
    return r;
}

// What a sample needs from its record, fast weights: the 2 ceil(W) + 2 ceil(W) window weights (zero where the reference skips
// a point), where its footprint starts in a tile buffer, where the sample goes, and whether this tile owns it at all.
// It does not depend on the image: the streaming kernel keeps it in registers over a run of images with the same angles.
template <int NF>
struct DgPrep {
    float wx[NF], wy[NF];
    int t0;                         // byte offset of the footprint's first point from the start of a tile buffer
    int soff;                       // sample's index in the image's output: (pe * nro + ro) * nrep + c0
    bool own;
};

// spoke holding record `rec` of the round's list: largest s with sp_start[s] <= rec (lane = rec & 63)
template <int MAXSP, class LdsT>
__device__ __forceinline__ int dg_spoke_of(LdsT &L, const DgRound rd, const int rec, const int lane)
{
    if (rd.mapped) {
        // the spoke of the pass's first record plus the segment starts between it and `rec`: two wave-uniform LDS
        // reads and a population count (a search over sp_start cost two more dependent LDS round trips per pass)
        const int b = rec >> 6;
        const unsigned long long starts = *reinterpret_cast<const unsigned long long *>(&L.segbits[2 * b]);
        const unsigned long long upto = ((2ull << lane) - 1ull) & ~1ull;           // records (64 b, rec]
        return L.first[b] + __popcll(starts & upto);
    }
    // 8-ary search: the seven splitters of a round are independent LDS reads
    int lo = 0;
    int span = rd.nacc;
    while (span > 1) {
        const int step = (span + 7) >> 3;
        const int end = lo + span;
        int sv[7];
#pragma unroll
        for (int j = 1; j < 8; ++j) sv[j - 1] = L.sp_start[min(lo + j * step, MAXSP)];
        int cnt = 0;
#pragma unroll
        for (int j = 1; j < 8; ++j) cnt += (lo + j * step < end && sv[j - 1] <= rec) ? 1 : 0;
        lo += cnt * step;
        span = min(step, end - lo);
    }
    return lo;
}

// Coordinates of record `rec` (src/tron.cu:554-561) and whether the tile at (tx0, ty0) owns the sample.
template <int MAXSP, class LdsT>
__device__ __forceinline__ bool dg_coords(const DegridParams &p, LdsT &L, const DgRound rd, const int rec, const int lane,
                                          const int tx0, const int ty0, const int n, const int nr, int &pe, int &ro, float &X, float &Y)
{
    const float half = (float)((n + 1) / 2), halfr = (float)((nr + 1) / 2);   // src/tron.cu:560-561
    const float inv_nro = 1.0f / (float)p.nro;
    const bool nro_pow2 = (p.nro & (p.nro - 1)) == 0;               // then ro / nro == ro * (1 / nro) exactly
    const int lo = dg_spoke_of<MAXSP>(L, rd, rec, lane);
    pe = L.sp_pe[lo];
    ro = (L.sp_seg[lo] & 0xffff) + (rec - L.sp_start[lo]);
    // thread's polar and Cartesian coordinates, src/tron.cu:554-561
    const float R = (nro_pow2 ? (float)ro * inv_nro : (float)ro / (float)p.nro) - 0.5f;
    const float2 cs = L.sp_cs[lo];
    X = cs.y; Y = cs.x;                                             // X = sin, Y = cos (src/tron.cu:559)
    X = (float)nr * R * X + halfr;
    Y = (float)n * R * Y + half;
    const int fx = min(max((int)floorf(X), 0), nr - 1);              // owner cell
    const int fy = min(max((int)floorf(Y), 0), n - 1);
    return (unsigned)(fx - tx0) < (unsigned)kDgTile && (unsigned)(fy - ty0) < (unsigned)kDgTile;
}

template <int CW, int MAXSP, int HALO, int SX, int SY, class LdsT>
__device__ __forceinline__ DgPrep<2 * CW> dg_prep(const DegridParams &p, const KbCoef &kb, LdsT &L, const DgRound rd, const int rec,
                                                  const int lane, const int tx0, const int ty0, const int n, const int nr, const int c0)
{
    constexpr int NF = 2 * CW;
    DgPrep<NF> P;
    const float W = p.W;
    int pe, ro;
    float X, Y;
    P.own = dg_coords<MAXSP>(p, L, rd, rec, lane, tx0, ty0, n, nr, pe, ro, X, Y);
    P.soff = (pe * p.nro + ro) * p.nrep + c0;
    // all weights first, as interleaved packed polynomials (x and y of a slot share an instruction).
    // 2*CW slots suffice: [X-W, X+W] holds more integers only when both end points sit at distance
    // exactly W, where the weight is 0; a slot with |d| >= W gets weight 0, which is what skipping it
    // (src/tron.cu:563,566) amounts to.  Then NF x NF fixed-offset LDS reads.
    const int xu0 = (int)ceilf(X - W), yu0 = (int)ceilf(Y - W);
    v2f sxy[NF], wxy[NF];
    const v2f one = {1.0f, 1.0f};
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        const v2f dxy = {(float)(xu0 + t) - X, (float)(yu0 + t) - Y};
        const v2f r = dxy * kb.invW;
        sxy[t] = __builtin_elementwise_fma(-r, r, one);
        // outside the window: s <- 1 - (W/W)^2 keeps the polynomial finite, the weight is zeroed below
        wxy[t] = (v2f){kb.poly[kKbPolyTerms - kb_terms(CW)], kb.poly[kKbPolyTerms - kb_terms(CW)]};
    }
#pragma unroll
    for (int k = kKbPolyTerms - kb_terms(CW) + 1; k < kKbPolyTerms; ++k) {
        const v2f c = {kb.poly[k], kb.poly[k]};
#pragma unroll
        for (int t = 0; t < NF; ++t) wxy[t] = __builtin_elementwise_fma(wxy[t], sxy[t], c);
    }
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        P.wx[t] = fabsf((float)(xu0 + t) - X) < W ? wxy[t].x : 0.0f;
        P.wy[t] = fabsf((float)(yu0 + t) - Y) < W ? wxy[t].y : 0.0f;
    }
    P.t0 = ((xu0 + HALO - tx0) * SX + (yu0 + HALO - ty0) * SY) * (int)sizeof(float2);
    return P;
}

// The sample of a prepared record from the tile buffer at L.tile[tile_off]: NF x NF points x CPB coils, stored to dst.
// ROLLED: one row of the footprint per iteration of a loop that is NOT unrolled -- a wave cannot have more than 15 LDS reads
// in flight anyway (lgkmcnt), the unrolled form holds ~150 registers, this one ~100 (same speed where both fit).
// PAIRS (CPB = 4; the WHOLE wave must call, lanes without a sample with P.own = false): the 32 bytes of a sample are stored by
// two neighbouring lanes, 16 bytes each, instead of by one lane in two instructions -- a store instruction then writes 32-byte
// pieces at the samples' 64-byte stride, not 16-byte ones (measured with the addresses faked: 1.72 -> 1.52 us per coil image;
// 16 bytes per lane with the lanes consecutive: 1.45).  The sums cross the lanes through `stage`, 128 float4 of LDS private
// to the wave.
template <int CPB, int CW, int PLANE, int SX, int SY, bool ROLLED, bool PAIRS = false, class LdsT>
__device__ __forceinline__ void dg_gather_store(const DegridParams &p, LdsT &L, const int tile_off, const DgPrep<2 * CW> &P,
                                                float2 *dst, const int ncb, float4 *stage = nullptr)
{
    constexpr int NF = 2 * CW;
    float2 acc[CPB];
#pragma unroll
    for (int c = 0; c < CPB; ++c) acc[c] = make_float2(0.f, 0.f);
    unsigned t0 = lds_addr(L.tile) + (unsigned)(tile_off * (int)sizeof(float2) + P.t0);
    // volatile keeps hipcc from pairing the reads into ds_read2_b64, which moves 128 B per clock on a 32-bank modulus;
    // ds_read_b64 moves 256 on 64 banks
    if (PAIRS && !P.own) {
        // no sample: nothing to gather, but this lane stores its half of a neighbour's below
    } else if (ROLLED) {
        float wx[NF];
#pragma unroll
        for (int t = 0; t < NF; ++t) wx[t] = P.wx[t];
#pragma unroll 1
        for (int sx = 0; sx < NF; ++sx) {
            const float wxs = wx[0];
#pragma unroll
            for (int t = 0; t < NF; ++t) {
                const float wgt = wxs * P.wy[t];                            // src/tron.cu:568
#pragma unroll
                for (int c = 0; c < CPB; ++c) {
                    const v2f v = *(const volatile __attribute__((address_space(3))) v2f *)(size_t)(
                        t0 + (unsigned)((c * PLANE + t * SY) * (int)sizeof(float2)));
                    acc[c].x = fmaf(v.x, wgt, acc[c].x);                    // src/tron.cu:573
                    acc[c].y = fmaf(v.y, wgt, acc[c].y);
                }
            }
#pragma unroll
            for (int t = 0; t + 1 < NF; ++t) wx[t] = wx[t + 1];
            t0 += (unsigned)(SX * (int)sizeof(float2));
        }
    } else {
#pragma unroll
        for (int sx = 0; sx < NF; ++sx)
#pragma unroll
            for (int t = 0; t < NF; ++t) {
                const float wgt = P.wx[sx] * P.wy[t];                       // src/tron.cu:568
#pragma unroll
                for (int c = 0; c < CPB; ++c) {
                    const v2f v = *(const volatile __attribute__((address_space(3))) v2f *)(size_t)(
                        t0 + (unsigned)((c * PLANE + sx * SX + t * SY) * (int)sizeof(float2)));
                    acc[c].x = fmaf(v.x, wgt, acc[c].x);                    // src/tron.cu:573
                    acc[c].y = fmaf(v.y, wgt, acc[c].y);
                }
            }
    }
    if constexpr (PAIRS && CPB == 4) {
        const int lane = threadIdx.x & 63;
        stage[2 * lane] = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
        stage[2 * lane + 1] = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
        const unsigned long long owners = __ballot(P.own);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int smp = 32 * i + (lane >> 1);                           // the sample this lane stores half of
            const int so = __shfl(P.soff, smp);
            const float4 v = stage[64 * i + lane];                          // = stage[2 * smp + (lane & 1)]
            if ((owners >> smp) & 1ull) *reinterpret_cast<float4 *>(dst + so + 2 * (lane & 1)) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");              // the next call's stage writes come after these reads
        __builtin_amdgcn_wave_barrier();
        return;
    }
    float2 *o = dst + P.soff;
    if (CPB % 2 == 0 && ncb == CPB && (p.nrep & 1) == 0) {                  // c0 is a multiple of CPB: 16-byte aligned
#pragma unroll
        for (int c = 0; c < CPB; c += 2)
            *reinterpret_cast<float4 *>(o + c) = make_float4(acc[c].x, acc[c].y, acc[c + 1].x, acc[c + 1].y);
    } else {
#pragma unroll
        for (int c = 0; c < CPB; ++c)
            if (c < ncb) o[c] = acc[c];
    }
}

// The samples of one round, dealt out flat over the NT threads.  The tile (CPB coil planes of PLANE points each, HALO points
// before the tile's first row and column) starts at L.tile[tile_off]; SX / SY are the strides of a step along the sine
// ("X") and cosine ("Y") axes.
// `tile_off` is an index, not a pointer: a pointer handed through a call can lose the LDS address space (flat loads).
template <int CPB, int CW, int KB, int NT, int MAXSP, int PLANE, int HALO, int SX, int SY, bool ROLLED, class LdsT>
__device__ __forceinline__ void dg_sample_loop(const DegridParams &p, const KbCoef &kb, LdsT &L, const int tile_off, const DgRound rd,
                                               const int tid, const int tx0, const int ty0, const int n, const int nr,
                                               float2 *dst, const int c0, const int ncb, const int first_rec = 0)
{
    const float W = p.W;
    const int nrec = rd.nrec;
    for (int rec = first_rec + tid; rec < nrec; rec += NT) {
        if (KB == TRON_KB_FAST) {
            const DgPrep<2 * CW> P = dg_prep<CW, MAXSP, HALO, SX, SY>(p, kb, L, rd, rec, tid & 63, tx0, ty0, n, nr, c0);
            if (P.own) dg_gather_store<CPB, CW, PLANE, SX, SY, ROLLED>(p, L, tile_off, P, dst, ncb);
            continue;
        }
        int pe, ro;
        float X, Y;
        if (!dg_coords<MAXSP>(p, L, rd, rec, tid & 63, tx0, ty0, n, nr, pe, ro, X, Y)) continue;

        constexpr int NS = 2 * CW + 1;                              // at most floor(2W)+1 integers in [X-W, X+W]
        const int xu0 = (int)ceilf(X - W), yu0 = (int)ceilf(Y - W);
        const int lrow0 = HALO - tx0, lcol0 = yu0 + HALO - ty0;
        float2 acc[CPB];
#pragma unroll
        for (int c = 0; c < CPB; ++c) acc[c] = make_float2(0.f, 0.f);
        {
            float wy[NS];
            int ny = 0;
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                wy[t] = 0.f;
                if ((float)(yu0 + t) <= (Y + W)) {                      // src/tron.cu:566
                    wy[t] = kb_weight<KB>((float)(yu0 + t) - Y, kb);
                    ny = t + 1;
                }
            }
            for (int xu = xu0; (float)xu <= (X + W); ++xu) {               // src/tron.cu:563
                const float wgtx = kb_weight<KB>((float)xu - X, kb);
                const int trow = tile_off + (xu + lrow0) * SX + lcol0 * SY;
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    if (t < ny) {
                        const float wgt = wgtx * wy[t];                         // src/tron.cu:568
#pragma unroll
                        for (int c = 0; c < CPB; ++c) {
                            const float2 v = L.tile[trow + c * PLANE + t * SY];
                            acc[c].x += v.x * wgt;                              // src/tron.cu:573, unfused
                            acc[c].y += v.y * wgt;
                        }
                    }
                }
            }
        }
        float2 *o = dst + ((size_t)pe * p.nro + ro) * p.nrep + c0;
        if (CPB % 2 == 0 && ncb == CPB && (p.nrep & 1) == 0) {          // c0 is a multiple of CPB: 16-byte aligned
#pragma unroll
            for (int c = 0; c < CPB; c += 2)
                *reinterpret_cast<float4 *>(o + c) = make_float4(acc[c].x, acc[c].y, acc[c + 1].x, acc[c + 1].y);
        } else {
#pragma unroll
            for (int c = 0; c < CPB; ++c)
                if (c < ncb) o[c] = acc[c];
        }
    }
}

}  // namespace tron
