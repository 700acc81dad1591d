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
#include "tron_device.h"

namespace tron {

constexpr int kDgTile = 32;
constexpr int kDgThreads = 256;
constexpr int kDgMaxSpokes = 512;

template <int CPB, int CW>
struct DgLds {
    static constexpr int HALO = CW + 1;
    static constexpr int TS = kDgTile + 2 * HALO;
    int sp_pe[kDgMaxSpokes];
    int sp_seg[kDgMaxSpokes];          // ro_lo | len << 16
    int sp_start[kDgMaxSpokes + 1];   // exclusive scan of len
    int wcnt[8];
    float2 tile[TS * TS * CPB];        // [coil][row][col]: neighbouring samples read neighbouring banks
};

template <int CPB, int CW, int KB>
__global__ void __launch_bounds__(kDgThreads) degrid_tile_kernel(const DegridParams p)
{
    using L_t = DgLds<CPB, CW>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    L_t &L = *reinterpret_cast<L_t *>(lds_raw);
    constexpr int HALO = L_t::HALO, TS = L_t::TS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.n;
    const int tpr = (n + kDgTile - 1) / kDgTile;
    const int tile = blockIdx.x % (tpr * tpr);
    const int k = blockIdx.x / (tpr * tpr);                    // image
    const int c0 = blockIdx.y * CPB;
    const int ncb = min(CPB, p.nrep - c0);
    const int tx0 = (tile / tpr) * kDgTile;                     // first row (sine axis, "X" of the reference)
    const int ty0 = (tile % tpr) * kDgTile;                     // first column (cosine axis, "Y")

    KbCoef kb;
    kb.W = p.W; kb.invW = 1.0f / p.W; kb.beta = p.beta;
#pragma unroll
    for (int t = 0; t < kKbPolyTerms; ++t) kb.poly[t] = p.kb_poly[t];
    const float W = p.W;

    // ---- tile + halo -> LDS (periodic wrap of src/tron.cu:569-570; fftshift(INVERSE) of :646 folded in)
    const float2 *src = p.udata + (size_t)k * p.in_z;
    for (int e = tid; e < TS * TS * CPB; e += kDgThreads) {
        const int c = e / (TS * TS), rc = e % (TS * TS);
        const int r = rc / TS, col = rc % TS;
        int i = tx0 - HALO + r, j = ty0 - HALO + col;               // wrap without integer division
        while (i < 0) i += n;
        while (i >= n) i -= n;
        while (j < 0) j += n;
        while (j >= n) j -= n;
        if (p.in_shift) {
            i += n / 2; if (i >= n) i -= n;
            j += n / 2; if (j >= n) j -= n;
        }
        float2 v = make_float2(0.f, 0.f);
        if (c < ncb) v = src[((size_t)i * n + j) * p.in_p + (size_t)(c0 + c) * p.in_c];
        L.tile[c * (TS * TS) + r * TS + col] = v;
    }

    const float half = (float)((n + 1) / 2);                    // src/tron.cu:560-561
    const float eps = 0.01f;
    const float bx_lo = (float)tx0 - eps, bx_hi = (float)(tx0 + kDgTile) + eps;
    const float by_lo = (float)ty0 - eps, by_hi = (float)(ty0 + kDgTile) + eps;
    float2 *dst = p.nudata + (size_t)k * p.nro * p.npe * p.nrep;

    for (int round0 = 0; round0 < p.npe; round0 += kDgMaxSpokes) {
        // ---- clip: thread = spoke; X(ro) = n*(ro/nro - 1/2)*sin + half, Y likewise with cos ----------
        if (tid == 0) L.sp_start[0] = 0;
        int nacc = 0;
        for (int chunk0 = round0; chunk0 < min(p.npe, round0 + kDgMaxSpokes); chunk0 += kDgThreads) {
            const int pe = chunk0 + tid;
            bool accept = false;
            int rlo = 0, len = 0;
            if (pe < p.npe && pe < round0 + kDgMaxSpokes) {
                const float2 cs = p.trig[pe];
                const float ax = (float)n * cs.y / (float)p.nro, ay = (float)n * cs.x / (float)p.nro;   // d/d(ro)
                const float ox = half - 0.5f * (float)n * cs.y, oy = half - 0.5f * (float)n * cs.x;     // value at ro = 0
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

        // ---- samples, dealt out flat over the 256 threads -------------------------------------------
        for (int rec = tid; rec < nrec; rec += kDgThreads) {
          {
            int lo = 0, hi = nacc - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (L.sp_start[mid] <= rec) lo = mid; else hi = mid - 1;
            }
            const int pe = L.sp_pe[lo];
            const int ro = (L.sp_seg[lo] & 0xffff) + (rec - L.sp_start[lo]);
            // thread's polar and Cartesian coordinates, src/tron.cu:554-561
            const float R = (float)ro / (float)p.nro - 0.5f;
            const float2 cs = p.trig[pe];
            float X = cs.y, Y = cs.x;                                   // X = sin, Y = cos (src/tron.cu:559)
            X = (float)n * R * X + half;
            Y = (float)n * R * Y + half;
            const int fx = min(max((int)floorf(X), 0), n - 1);           // owner cell
            const int fy = min(max((int)floorf(Y), 0), n - 1);
            if ((unsigned)(fx - tx0) >= (unsigned)kDgTile || (unsigned)(fy - ty0) >= (unsigned)kDgTile) continue;

            const int yu0 = (int)ceilf(Y - W);
            float wy[2 * CW + 1];                                    // at most floor(2W)+1 integers in [Y-W, Y+W]
            int ny = 0;
#pragma unroll
            for (int t = 0; t < 2 * CW + 1; ++t) {
                wy[t] = 0.f;
                if ((float)(yu0 + t) <= (Y + W)) {                      // src/tron.cu:566
                    wy[t] = kb_weight<KB>((float)(yu0 + t) - Y, kb);
                    ny = t + 1;
                }
            }
            float2 acc[CPB];
#pragma unroll
            for (int c = 0; c < CPB; ++c) acc[c] = make_float2(0.f, 0.f);
            const int lrow0 = HALO - tx0, lcol0 = yu0 + HALO - ty0;
            for (int xu = (int)ceilf(X - W); (float)xu <= (X + W); ++xu) {     // src/tron.cu:563
                const float wgtx = kb_weight<KB>((float)xu - X, kb);
                const float2 *trow = L.tile + (xu + lrow0) * TS + lcol0;
#pragma unroll
                for (int t = 0; t < 2 * CW + 1; ++t) {
                    if (t < ny) {
                        const float wgt = wgtx * wy[t];                         // src/tron.cu:568
#pragma unroll
                        for (int c = 0; c < CPB; ++c) {
                            const float2 v = trow[c * (TS * TS) + t];
                            if (KB == TRON_KB_EXACT) {
                                acc[c].x += v.x * wgt;                          // src/tron.cu:573, unfused
                                acc[c].y += v.y * wgt;
                            } else {
                                acc[c].x = fmaf(v.x, wgt, acc[c].x);
                                acc[c].y = fmaf(v.y, wgt, acc[c].y);
                            }
                        }
                    }
                }
            }
            float2 *o = dst + ((size_t)pe * p.nro + ro) * p.nrep + c0;
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
    const int tpr = (p.n + kDgTile - 1) / kDgTile;
    const int chunks = (p.nrep + CPB - 1) / CPB;
    dim3 grid((unsigned)((size_t)tpr * tpr * p.nimg), (unsigned)chunks);
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

// Requires n >= 32 + 2*(ceil(W)+1) (the halo must not wrap onto the tile itself) and W <= 3.
hipError_t launch_degrid_tile(const DegridParams &p, int kb_mode, hipStream_t s)
{
    const int cw = (int)ceilf(p.W);
    switch (cw) {
        case 1: return launch_degrid_tile_cw<1>(p, kb_mode, s);
        case 2: return launch_degrid_tile_cw<2>(p, kb_mode, s);
        case 3: return launch_degrid_tile_cw<3>(p, kb_mode, s);
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
