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

#include "tron_degrid_sample.h"

namespace tron {

constexpr int kDgThreads = 256;
constexpr int kDgMaxSpokes = 256;   // spokes clipped per round (one per thread)
constexpr int kDgMaxBlocks = 192;   // 64-record blocks indexed by the inverse map (more records: 8-ary search); 256 spokes x 47 samples
                                    // fit, and three workgroups' LDS still fits the CU (1280-byte allocation granules)

template <int CPB, int CW>
struct DgLds : DgLists<kDgMaxSpokes, kDgMaxBlocks, kDgThreads / 64> {
    static constexpr int HALO = CW + 1;
    static constexpr int TS = kDgTile + 2 * HALO;
    float2 tile[TS * TS * CPB + 8];    // [coil][row][col]: neighbouring samples read neighbouring banks; zeroed pad
};

template <int CPB, int CW, int KB>
__global__ void __launch_bounds__(kDgThreads) degrid_tile_kernel(const DegridParams p)
{
    using L_t = DgLds<CPB, CW>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    L_t &L = *reinterpret_cast<L_t *>(lds_raw);
    constexpr int HALO = L_t::HALO, TS = L_t::TS;

    const int tid = threadIdx.x;
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
            if (p.in_transposed) { i += p.in_rot; if (i >= nr) i -= nr; } else { j += p.in_rot; if (j >= n) j -= n; }
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

    float2 *dst = p.nudata + (size_t)k * p.nro * p.npe * p.nrep;
    for (int round0 = 0; round0 < p.npe; round0 += kDgMaxSpokes) {
        const DgRound rd = dg_clip_round<kDgThreads, kDgMaxSpokes, kDgMaxBlocks>(p, L, k, round0, tid, tx0, ty0, n, nr);
        // (keeping the tile loads in flight across the first clip round was tried: the registers it pins cost more
        //  than the exposed latency, 2.78 -> 2.91 us per coil image; two records per thread side by side, to overlap one's
        //  LDS reads with the other's arithmetic: 2.18 -> 2.40)
        dg_sample_loop<CPB, CW, KB, kDgThreads, kDgMaxSpokes, TS * TS, HALO, TS, 1, false>(p, kb, L, 0, rd, tid, tx0, ty0, n, nr, dst, c0, ncb);
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
    if (p.nrep >= 4) return launch_degrid_tile_cpb<4, CW>(p, kb_mode, s);
    if (p.nrep >= 2) return launch_degrid_tile_cpb<2, CW>(p, kb_mode, s);
    return launch_degrid_tile_cpb<1, CW>(p, kb_mode, s);
}

// Requires W <= 4 (any grid size, square or not: the halo wraps periodically, onto the tile itself when the grid is small).
// A non-square grid (p.nrows) must come with p.tile_order == nullptr (raster order).
hipError_t launch_degrid_tile(const DegridParams &p, int kb_mode, hipStream_t s)
{
    const int cw = (int)ceilf(p.W);
    if ((long long)p.nro * p.npe * p.nrep >= (1ll << 31)) return hipErrorInvalidValue;      // a sample's output index is an int (DgPrep::soff)
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
