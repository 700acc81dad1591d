// Tiled degridding (forward interpolation) = degridradial2d of the reference, src/tron.cu:540-577.
//
// The reference runs one thread per k-space sample and reads its <= (2W+1)^2 Cartesian neighbours
// straight from global memory, re-evaluating the y weight inside the x loop.  Here one workgroup owns a
// 32x32 tile of the Cartesian grid: it copies the tile plus a (ceil(W)+1)-point halo into LDS once (for CPB
// coils, periodic wrap and the second fftshift of src/tron.cu:646 folded into the load index), clips every
// spoke against the tile (thread = spoke, ballot compaction), and then deals the samples that fall inside
// the tile out flat over its 256 threads.  A sample is owned by the tile that holds floor(X), floor(Y), so
// every sample is produced exactly once, with the reference's own coordinate arithmetic, weights and
// accumulation order (xu outer, yu inner): TRON_KB_EXACT is bit-identical to the reference loop.
#include <stdlib.h>

#include "tron_device.h"

namespace tron {

constexpr int kDgTile = 32;
constexpr int kDgThreads = 256;
constexpr int kDgMaxSpokes = 256;   // spokes clipped per round (one per thread)
constexpr int kDgMaxBlocks = 256;   // 64-record blocks indexed by the inverse map (more records: 8-ary search)

template <int CPB, int CW>
struct DgLds {
    static constexpr int HALO = CW + 1;
    static constexpr int TS = kDgTile + 2 * HALO;
    int sp_pe[kDgMaxSpokes];
    int sp_seg[kDgMaxSpokes];          // ro_lo | len << 16
    int sp_start[kDgMaxSpokes + 1];   // exclusive scan of len
    float2 sp_cs[kDgMaxSpokes];        // (cos, sin) of the accepted spokes: the sample loop stays off global memory
    int wcnt[8];
    unsigned short first[kDgMaxBlocks];   // spoke slot holding record 64*b: starts the per-lane spoke search
    float2 tile[TS * TS * CPB + 8];    // [coil][row][col]: neighbouring samples read neighbouring banks; zeroed pad
};

template <int CPB, int CW, int KB>
__global__ void __launch_bounds__(kDgThreads) degrid_tile_kernel(const DegridParams p)
{
    using L_t = DgLds<CPB, CW>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    L_t &L = *reinterpret_cast<L_t *>(lds_raw);
    constexpr int HALO = L_t::HALO, TS = L_t::TS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.n;                                          // columns (cosine axis, "Y" of the reference)
    const int nr = p.nrows > 0 ? p.nrows : n;                   // rows (sine axis, "X"): differs for non-square forward transforms
    const int tpr = (n + kDgTile - 1) / kDgTile, tprr = (nr + kDgTile - 1) / kDgTile;
    // centre tiles hold the most samples (density ~ 1/r): they are dispatched first, all images of a tile together
    // Workgroup id -> (tile, image, coil chunk).  The coil chunks of one (tile, image) write interleaved 8*CPB-byte
    // pieces of the same output lines (samples are coil-interleaved, src/tron.cu:550): they are placed 8 ids apart, i.e.
    // on the same XCD and next to each other in time, so its L2 merges the pieces before they reach HBM (dispatched
    // far apart they cost 2 us per coil image in partial-line writes).
    const int chunks = (p.nrep + CPB - 1) / CPB;
    const int grp = blockIdx.x / (8 * chunks), within = blockIdx.x % (8 * chunks);
    const int ti = grp * 8 + (within & 7);                      // (tile, image) index
    if (ti >= tpr * tprr * p.nimg) return;
    const int tile = p.tile_order ? p.tile_order[ti / p.nimg] : ti % (tpr * tprr);
    const int k = p.tile_order ? ti % p.nimg : ti / (tpr * tprr);    // image
    const int c0 = (within >> 3) * CPB;
    const int ncb = min(CPB, p.nrep - c0);
    const int tx0 = (tile / tpr) * kDgTile;                     // first row (sine axis, "X" of the reference)
    const int ty0 = (tile % tpr) * kDgTile;                     // first column (cosine axis, "Y")

    {   // every sample lies within n/2 of the grid centre (src/tron.cu:554-561: |R| <= 1/2): a tile whose nearest cell is
        // farther away owns none -- nothing to load, nothing to produce (the corners of the square: 12 % of the tiles)
        const float hc = (float)((n + 1) / 2);
        const float dx = fmaxf(fmaxf((float)tx0 - hc, hc - (float)(tx0 + kDgTile)), 0.f);
        const float dy = fmaxf(fmaxf((float)ty0 - hc, hc - (float)(ty0 + kDgTile)), 0.f);
        const float lim = 0.5f * (float)n + 1.5f;
        if (nr == n && dx * dx + dy * dy > lim * lim) return;
    }
    KbCoef kb;
    kb.W = p.W; kb.invW = 1.0f / p.W; kb.beta = p.beta;
#pragma unroll
    for (int t = 0; t < kKbPolyTerms; ++t) kb.poly[t] = p.kb_poly[t];
    const float W = p.W;

    // ---- tile + halo -> LDS (periodic wrap of src/tron.cu:569-570; fftshift(INVERSE) of :646 folded in)
    const float2 *src = p.udata + (size_t)k * p.in_z + (size_t)c0 * p.in_c;
    {
        // every load of the tile is issued before the first LDS store: one HBM latency per block, not one per pass
        constexpr int NIT = (TS * TS + kDgThreads - 1) / kDgThreads;
        float2 stage[NIT][CPB];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * kDgThreads;
            const int ea = e / TS, eb = e - ea * TS;
            const int r = p.in_transposed ? eb : ea, col = p.in_transposed ? ea : eb;   // the fastest index follows memory
            int i = tx0 - HALO + r, j = ty0 - HALO + col;               // periodic wrap, src/tron.cu:569-570
            if (n >= TS && nr >= TS) {                                  // -n <= i < 2n: one step each way
                i += i < 0 ? nr : 0; i -= i >= nr ? nr : 0;
                j += j < 0 ? n : 0; j -= j >= n ? n : 0;
            } else {
                i %= nr; i += i < 0 ? nr : 0;
                j %= n; j += j < 0 ? n : 0;
            }
            if (p.in_shift) {
                i += nr / 2; if (i >= nr) i -= nr;
                j += n / 2; if (j >= n) j -= n;
            }
            const float2 *s = src + (p.in_transposed ? (size_t)j * nr + i : (size_t)i * n + j) * p.in_p;
#pragma unroll
            for (int c = 0; c < CPB; ++c)
                stage[it][c] = (e < TS * TS && c < ncb) ? s[(size_t)c * p.in_c] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + it * kDgThreads;
            const int ea = e / TS, eb = e - ea * TS;
            const int slot = p.in_transposed ? eb * TS + ea : e;
            if (e < TS * TS) {
#pragma unroll
                for (int c = 0; c < CPB; ++c) L.tile[c * (TS * TS) + slot] = stage[it][c];
            }
        }
    }
    if (tid < 8) L.tile[TS * TS * CPB + tid] = make_float2(0.f, 0.f);

    const float half = (float)((n + 1) / 2), halfr = (float)((nr + 1) / 2);   // src/tron.cu:560-561
    const float eps = 0.01f;
    const float bx_lo = (float)tx0 - eps, bx_hi = (float)(tx0 + kDgTile) + eps;
    const float by_lo = (float)ty0 - eps, by_hi = (float)(ty0 + kDgTile) + eps;
    float2 *dst = p.nudata + (size_t)k * p.nro * p.npe * p.nrep;

    for (int round0 = 0; round0 < p.npe && TRON_DBG_LT(p, 2); round0 += kDgMaxSpokes) {
        // ---- clip: thread = spoke; X(ro) = n*(ro/nro - 1/2)*sin + half, Y likewise with cos ----------
        if (tid == 0) L.sp_start[0] = 0;
        int nacc = 0;
        for (int chunk0 = round0; chunk0 < min(p.npe, round0 + kDgMaxSpokes); chunk0 += kDgThreads) {
            const int pe = chunk0 + tid;
            bool accept = false;
            int rlo = 0, len = 0;
            float2 cs = make_float2(0.f, 0.f);
            if (pe < p.npe && pe < round0 + kDgMaxSpokes) {
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
            for (int w = 0; w < 4; ++w) {
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
        {   // exclusive scan of the segment lengths
            const int i0 = 2 * tid, i1 = 2 * tid + 1;
            const int l0 = i0 < nacc ? (L.sp_seg[i0] >> 16) : 0;
            const int l1 = i1 < nacc ? (L.sp_seg[i1] >> 16) : 0;
            int v = l0 + l1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(v, o);
                if (lane >= o) v += t;
            }
            if (lane == 63) L.wcnt[4 + wave] = v;
            __syncthreads();
            int wbase = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (w < wave) wbase += L.wcnt[4 + w];
            const int excl = wbase + v - (l0 + l1);
            if (i0 < nacc) L.sp_start[i0 + 1] = excl + l0;
            if (i1 < nacc) L.sp_start[i1 + 1] = excl + l0 + l1;
            __syncthreads();
        }
        const int nrec = L.sp_start[nacc];
        const bool mapped = nrec <= 64 * kDgMaxBlocks;
        if (mapped) {
            // inverse map: which spoke holds record 64*b (a wave pass covers exactly one such block)
            for (int sidx = tid; sidx < nacc; sidx += kDgThreads) {
                const int st = L.sp_start[sidx], en = L.sp_start[sidx + 1];
                for (int b = (st + 63) >> 6; 64 * b < en; ++b) L.first[b] = (unsigned short)sidx;
            }
            __syncthreads();
        }
        const float inv_nro = 1.0f / (float)p.nro;
        const bool nro_pow2 = (p.nro & (p.nro - 1)) == 0;               // then ro / nro == ro * (1 / nro) exactly

        // ---- samples, dealt out flat over the 256 threads -------------------------------------------
        // (keeping the tile loads in flight across the first clip round was tried: the registers it pins cost more
        //  than the exposed latency, 2.78 -> 2.91 us per coil image)
        for (int rec = tid; rec < nrec && TRON_DBG_LT(p, 1); rec += kDgThreads) {
            // spoke holding record `rec`: largest s with sp_start[s] <= rec
            int lo;
            if (mapped) {
                // start at the spoke of the pass's first record, then count the segment starts up to `rec`, four
                // independent LDS reads at a time (segments are tens of records long: one round as a rule)
                lo = L.first[rec >> 6];
                for (;;) {
                    int sv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) sv[j] = L.sp_start[min(lo + 1 + j, nacc)];   // sp_start[nacc] = nrec > rec
                    int cnt = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) cnt += sv[j] <= rec ? 1 : 0;
                    lo += cnt;
                    if (cnt < 4) break;
                }
            } else {
                // 8-ary search: the seven splitters of a round are independent LDS reads
                lo = 0;
                int span = nacc;
                while (span > 1) {
                    const int step = (span + 7) >> 3;
                    const int end = lo + span;
                    int sv[7];
#pragma unroll
                    for (int j = 1; j < 8; ++j) sv[j - 1] = L.sp_start[min(lo + j * step, kDgMaxSpokes)];
                    int cnt = 0;
#pragma unroll
                    for (int j = 1; j < 8; ++j) cnt += (lo + j * step < end && sv[j - 1] <= rec) ? 1 : 0;
                    lo += cnt * step;
                    span = min(step, end - lo);
                }
            }
            const int pe = L.sp_pe[lo];
            const int ro = (L.sp_seg[lo] & 0xffff) + (rec - L.sp_start[lo]);
            // thread's polar and Cartesian coordinates, src/tron.cu:554-561
            const float R = (nro_pow2 ? (float)ro * inv_nro : (float)ro / (float)p.nro) - 0.5f;
            const float2 cs = L.sp_cs[lo];
            float X = cs.y, Y = cs.x;                                   // X = sin, Y = cos (src/tron.cu:559)
            X = (float)nr * R * X + halfr;
            Y = (float)n * R * Y + half;
            const int fx = min(max((int)floorf(X), 0), nr - 1);          // owner cell
            const int fy = min(max((int)floorf(Y), 0), n - 1);
            if ((unsigned)(fx - tx0) >= (unsigned)kDgTile || (unsigned)(fy - ty0) >= (unsigned)kDgTile) continue;

            constexpr int NS = 2 * CW + 1;                              // at most floor(2W)+1 integers in [X-W, X+W]
            const int xu0 = (int)ceilf(X - W), yu0 = (int)ceilf(Y - W);
            const int lrow0 = HALO - tx0, lcol0 = yu0 + HALO - ty0;
            float2 acc[CPB];
#pragma unroll
            for (int c = 0; c < CPB; ++c) acc[c] = make_float2(0.f, 0.f);

            if (KB == TRON_KB_FAST) {
                // all weights first, as interleaved packed polynomials (x and y of a slot share an instruction).
                // 2*CW slots suffice: [X-W, X+W] holds more integers only when both end points sit at distance
                // exactly W, where the weight is 0; a slot with |d| >= W gets weight 0, which is what skipping it
                // (src/tron.cu:563,566) amounts to.  Then NF x NF fixed-offset LDS reads.
                constexpr int NF = 2 * CW;
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
                float wx[NF], wy[NF];
#pragma unroll
                for (int t = 0; t < NF; ++t) {
                    wx[t] = fabsf((float)(xu0 + t) - X) < W ? wxy[t].x : 0.0f;
                    wy[t] = fabsf((float)(yu0 + t) - Y) < W ? wxy[t].y : 0.0f;
                }
                const float2 *t0 = L.tile + (xu0 + lrow0) * TS + lcol0;
#pragma unroll
                for (int sx = 0; sx < NF; ++sx)
#pragma unroll
                    for (int t = 0; t < NF; ++t) {
                        const float wgt = wx[sx] * wy[t];                       // src/tron.cu:568
#pragma unroll
                        for (int c = 0; c < CPB; ++c) {
                            const float2 v = t0[c * (TS * TS) + sx * TS + t];
                            acc[c].x = fmaf(v.x, wgt, acc[c].x);                // src/tron.cu:573
                            acc[c].y = fmaf(v.y, wgt, acc[c].y);
                        }
                    }
            } else {
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
                    const float2 *trow = L.tile + (xu + lrow0) * TS + lcol0;
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        if (t < ny) {
                            const float wgt = wgtx * wy[t];                         // src/tron.cu:568
#pragma unroll
                            for (int c = 0; c < CPB; ++c) {
                                const float2 v = trow[c * (TS * TS) + t];
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
        __syncthreads();
    }
}

template <int CPB, int CW>
static hipError_t launch_degrid_tile_cpb(const DegridParams &p, int kb_mode, hipStream_t s)
{
    const int tpr = (p.n + kDgTile - 1) / kDgTile, tprr = ((p.nrows > 0 ? p.nrows : p.n) + kDgTile - 1) / kDgTile;
    const int chunks = (p.nrep + CPB - 1) / CPB;
    const size_t nti = (size_t)tpr * tprr * p.nimg;
    dim3 grid((unsigned)(((nti + 7) / 8) * 8 * chunks));
    const size_t lds = sizeof(DgLds<CPB, CW>);
    static_assert(sizeof(DgLds<CPB, CW>) <= 64 * 1024, "degrid tile must fit the default dynamic LDS limit");
    if (kb_mode == TRON_KB_EXACT)
        hipLaunchKernelGGL((degrid_tile_kernel<CPB, CW, TRON_KB_EXACT>), grid, dim3(kDgThreads), lds, s, p);
    else
        hipLaunchKernelGGL((degrid_tile_kernel<CPB, CW, TRON_KB_FAST>), grid, dim3(kDgThreads), lds, s, p);
    return hipGetLastError();
}

template <int CW>
static hipError_t launch_degrid_tile_cw(const DegridParams &p, int kb_mode, hipStream_t s)
{
    static const int force = tuning_env("TRON_DEGRID_CPB") ? atoi(tuning_env("TRON_DEGRID_CPB")) : 0;   // tuning knob
    if (force == 2) return launch_degrid_tile_cpb<2, CW>(p, kb_mode, s);
    if (force == 1) return launch_degrid_tile_cpb<1, CW>(p, kb_mode, s);
    if (p.nrep >= 4) return launch_degrid_tile_cpb<4, CW>(p, kb_mode, s);
    if (p.nrep >= 2) return launch_degrid_tile_cpb<2, CW>(p, kb_mode, s);
    return launch_degrid_tile_cpb<1, CW>(p, kb_mode, s);
}

// Requires W <= 4 (any grid size, square or not: the halo wraps periodically, onto the tile itself when the grid is small).
// A non-square grid (p.nrows) must come with p.tile_order == nullptr (raster order).
hipError_t launch_degrid_tile(const DegridParams &p, int kb_mode, hipStream_t s)
{
    const int cw = (int)ceilf(p.W);
    switch (cw) {
        case 1: return launch_degrid_tile_cw<1>(p, kb_mode, s);
        case 2: return launch_degrid_tile_cw<2>(p, kb_mode, s);
        case 3: return launch_degrid_tile_cw<3>(p, kb_mode, s);
        case 4: return launch_degrid_tile_cw<4>(p, kb_mode, s);
        default: return hipErrorInvalidValue;
    }
}

__global__ void warm_degrid_tile_tu() {}

hipError_t warm_degrid_tile()   // see warm_kernels() in tron_kernels.hip
{
    hipLaunchKernelGGL(warm_degrid_tile_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
