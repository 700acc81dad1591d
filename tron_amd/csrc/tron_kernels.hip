// HIP kernels of libtronhip for gfx950 (MI355X): radial gridding (adjoint interpolation),
// degridding (forward interpolation) and the fused pad/crop/deapodise/coil-combine passes.
//
// What each kernel computes follows the reference (davidssmith/TRON, src/tron.cu; cited per
// kernel); how it computes it does not.  The reference grids with one thread per Cartesian
// point scanning EVERY spoke (src/tron.cu:507-530, ~425 kernel evaluations per useful
// accumulation).  Here one workgroup owns a 16x16 Cartesian tile and
//   1. clips all spokes against the tile (one lane per spoke, wave64 ballot + popcount
//      compaction keeps the accepted spokes in acquisition order),
//   2. stages the accepted spoke segments in LDS in batches: lanes run ALONG the spoke, so
//      k-space is read coalesced from HBM exactly once per tile-halo, and each sample's
//      2x(2*ceil(W)) separable Kaiser-Bessel weights, density compensation and footprint
//      origin are computed once per sample (shared by all coils and all 16 footprint points),
//   3. lets each thread gather, for its own Cartesian point, the staged samples whose
//      footprint covers it.  Points accumulate in registers: no atomics, deterministic, and
//      the summation order per point (spoke ascending; positive radii ascending, then negative
//      radii ascending) is the reference's own, so TRON_KB_EXACT reproduces the reference's
//      fp32 sums bit for bit.
// Build with -ffp-contract=off: every fused multiply-add in this file is an explicit fmaf().
#include "tron_internal.h"

#include <hip/hip_fp16.h>

#include "../../include/tron_hip.h"

namespace tron {

// ------------------------------------------------------------------------- Kaiser-Bessel

// src/tron.cu:304-321, op for op: the coefficient literals are doubles, so both Horner chains
// run in double (unfused) and are rounded to float; the quotient is an IEEE float division.
__device__ __forceinline__ float besseli0_ref(const float x)
{
    if (x == 0.f) return 1.f;
    float z = x * x;
    float num = (z* (z* (z* (z* (z* (z* (z* (z* (z* (z* (z* (z* (z*
        (z* 0.210580722890567e-22  + 0.380715242345326e-19 ) +
        0.479440257548300e-16) + 0.435125971262668e-13 ) +
        0.300931127112960e-10) + 0.160224679395361e-7  ) +
        0.654858370096785e-5)  + 0.202591084143397e-2  ) +
        0.463076284721000e0)   + 0.754337328948189e2   ) +
        0.830792541809429e4)   + 0.571661130563785e6   ) +
        0.216415572361227e8)   + 0.356644482244025e9   ) +
        0.144048298227235e10);
    float den = (z*(z*(z-0.307646912682801e4)+
        0.347626332405882e7)-0.144048298227235e10);
    return -num/den;
}

struct KbCoef {
    float W, beta;
    int terms;
    const float *poly;
};

// src/tron.cu:338-349.  EXACT: the reference's expression tree.  FAST: I0(beta*sqrt(s)) is
// an entire function of s = 1-(x/W)^2 with positive Taylor coefficients, evaluated by Horner
// in fp32 (no sqrt, no division); coefficients (incl. 0.5/W) come from the host.
template <int KB>
__device__ __forceinline__ float kb_weight(const float x, const KbCoef &k)
{
    if (!(fabsf(x) < k.W)) return 0.0f;
    if (KB == TRON_KB_EXACT) {
        float r = x / k.W;
        float f = sqrtf(1.0f - r * r);
        return 0.5f * besseli0_ref(k.beta * f) / k.W;
    } else {
        float r = x / k.W;
        float s = fmaf(-r, r, 1.0f);
        float acc = k.poly[0];
        for (int t = 1; t < k.terms; ++t) acc = fmaf(acc, s, k.poly[t]);
        return acc;
    }
}

__device__ __forceinline__ float safe_rcp(float c)
{
    return fabsf(c) > 1e-12f ? 1.0f / c : copysignf(1e12f, c);
}

// ------------------------------------------------------------------------- gridding

struct SpokeEntry {   // an accepted spoke of the current clip chunk
    float ct, st;
    int pe;
    int rlo, rhi;
    int pad0, pad1, pad2;
};

struct BatchSpoke {   // a staged spoke
    float ct, st, reach, inv;
    int rlo, rhi, use_x, pad;
};

template <int CW>
struct GridCfg {
    static constexpr int LPS = (CW <= 2) ? 32 : 64;       // lanes (= LDS record slots) per spoke
    static constexpr int SPI = kGridThreads / LPS;        // spokes staged per iteration
    static constexpr int NREC = kBatchSpokes * LPS;
    static constexpr int NW = 2 * CW;                     // footprint points per dimension
};

size_t grid_lds_bytes(int cpb, int cw)
{
    const int lps = (cw <= 2) ? 32 : 64;
    const size_t nrec = (size_t)kBatchSpokes * lps;
    size_t b = 0;
    b += sizeof(SpokeEntry) * kGridThreads;
    b += sizeof(BatchSpoke) * kBatchSpokes;
    b += 16;                                  // wave counters
    b += nrec * sizeof(uint32_t);             // footprint origins
    b += nrec * 2 * cw * sizeof(float) * 2;   // wx, wy
    b += nrec * (size_t)cpb * sizeof(float2); // samples
    return b;
}

template <bool HALF>
__device__ __forceinline__ float2 load_sample(const void *base, size_t idx)
{
    if (HALF) {
        const __half2 h = reinterpret_cast<const __half2 *>(base)[idx];
        return __half22float2(h);
    } else {
        return reinterpret_cast<const float2 *>(base)[idx];
    }
}

// = precompensate + gridradial2d (src/tron.cu:405-416, 465-536) for a batch of slices.
// grid = (ntiles*nslices, coil chunks); block = 256 threads = one 16x16 tile.
template <int CPB, int CW, int KB, bool HALF>
__global__ void __launch_bounds__(kGridThreads)
grid_tile_kernel(const GridParams p)
{
    using C = GridCfg<CW>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    SpokeEntry *s_list = reinterpret_cast<SpokeEntry *>(lds_raw);
    BatchSpoke *s_sp = reinterpret_cast<BatchSpoke *>(s_list + kGridThreads);
    int *s_wcnt = reinterpret_cast<int *>(s_sp + kBatchSpokes);
    uint32_t *s_b = reinterpret_cast<uint32_t *>(s_wcnt + 4);
    float *s_wx = reinterpret_cast<float *>(s_b + C::NREC);
    float *s_wy = s_wx + C::NREC * C::NW;
    float2 *s_d = reinterpret_cast<float2 *>(s_wy + C::NREC * C::NW);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int z = blockIdx.x % p.nslices;
    const int tile = p.tile_order[blockIdx.x / p.nslices];
    const int c0 = p.coil0 + blockIdx.y * CPB;
    const int ncb = min(CPB, p.nchan - c0);
    const int n = p.nxos;
    const int h = n / 2;
    const int rmax = n / 2 - 1;

    // tile origin and this thread's point, centred coordinates (src/tron.cu:495-496)
    const int x0 = (tile % p.tiles_per_row) * kTile - h;
    const int y0 = (tile / p.tiles_per_row) * kTile - h;
    const int X = x0 + (lane & 15);
    const int Y = y0 + wave * 4 + (lane >> 4);
    const bool inside = (X + h < n) && (Y + h < n);
    int Rlo = 1 << 20, Rhi = -1;                // empty band for points outside the grid
    if (inside) {
        const uint32_t bnd = p.band[(size_t)(Y + h) * n + (X + h)];   // src/tron.cu:498-502
        Rlo = (int)(bnd & 0xffffu);
        Rhi = (int)(bnd >> 16);
    }
    const float Xf = (float)X, Yf = (float)Y;

    KbCoef kb;
    kb.W = p.W; kb.beta = p.beta; kb.terms = p.kb_terms; kb.poly = p.kb_poly;

    float2 acc[CPB];
#pragma unroll
    for (int c = 0; c < CPB; ++c) acc[c] = make_float2(0.f, 0.f);

    const unsigned char *in_bytes = reinterpret_cast<const unsigned char *>(p.nudata)
        + (size_t)z * (size_t)p.in_slice_stride * (HALF ? sizeof(__half2) : sizeof(float2));
    const float2 *trig = p.trig + (size_t)z * p.trig_slice_stride;

    const float eps = 0.01f;
    const float bx_lo = (float)x0 - p.W - eps, bx_hi = (float)(x0 + kTile - 1) + p.W + eps;
    const float by_lo = (float)y0 - p.W - eps, by_hi = (float)(y0 + kTile - 1) + p.W + eps;

    for (int chunk0 = 0; chunk0 < p.npe; chunk0 += kGridThreads) {
        // ---- 1. clip: one lane per spoke -------------------------------------------------
        const int pe = chunk0 + tid;
        bool accept = false;
        float ct = 0.f, st = 0.f;
        int rlo = 0, rhi = -1;
        if (pe < p.npe) {
            const float2 cs = trig[pe];
            ct = cs.x; st = cs.y;
            const float ic = safe_rcp(ct), is = safe_rcp(st);
            const float xa = bx_lo * ic, xb = bx_hi * ic;
            const float ya = by_lo * is, yb = by_hi * is;
            const float lo = fmaxf(fmaxf(fminf(xa, xb), fminf(ya, yb)), -(float)rmax);
            const float hi = fminf(fminf(fmaxf(xa, xb), fmaxf(ya, yb)), (float)rmax);
            if (lo <= hi) {
                rlo = (int)ceilf(lo);
                rhi = (int)floorf(hi);
                accept = rlo <= rhi;
                if (accept && rhi - rlo + 1 > C::LPS) {   // cannot happen for a 16x16 tile; flag it if it does
                    atomicOr(p.errflag, 1u);
                    rhi = rlo + C::LPS - 1;
                }
            }
        }
        const unsigned long long m = __ballot(accept);
        if (lane == 0) s_wcnt[wave] = __popcll(m);
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int cnt = s_wcnt[w];
            if (w < wave) base += cnt;
            total += cnt;
        }
        if (accept) {
            SpokeEntry e;
            e.ct = ct; e.st = st; e.pe = pe; e.rlo = rlo; e.rhi = rhi; e.pad0 = e.pad1 = e.pad2 = 0;
            s_list[base + __popcll(m & ((1ull << lane) - 1ull))] = e;
        }
        __syncthreads();

        for (int b0 = 0; b0 < total; b0 += kBatchSpokes) {
            const int nsp = min(kBatchSpokes, total - b0);
            // ---- 2. stage: lanes run along the spoke ----------------------------------
            for (int sidx = tid / C::LPS; sidx < nsp; sidx += C::SPI) {
                const SpokeEntry e = s_list[b0 + sidx];
                const int k = tid % C::LPS;
                const int r = e.rlo + k;
                if (k == 0) {
                    BatchSpoke bs;
                    bs.ct = e.ct; bs.st = e.st;
                    bs.reach = (float)CW * (fabsf(e.ct) + fabsf(e.st)) + 1e-3f;
                    bs.use_x = fabsf(e.ct) >= fabsf(e.st);
                    bs.inv = 1.0f / (bs.use_x ? e.ct : e.st);
                    bs.rlo = e.rlo; bs.rhi = e.rhi; bs.pad = 0;
                    s_sp[sidx] = bs;
                }
                if (r <= e.rhi) {
                    const int rec = sidx * C::LPS + k;
                    const float kx = (float)r * e.ct;                 // src/tron.cu:514-515
                    const float ky = (float)r * e.st;
                    const int bx = (int)floorf(kx) - CW + 1;
                    const int by = (int)floorf(ky) - CW + 1;
                    s_b[rec] = ((uint32_t)bx & 0xffffu) | ((uint32_t)by << 16);
#pragma unroll
                    for (int i = 0; i < C::NW; ++i) {
                        s_wx[rec * C::NW + i] = kb_weight<KB>(kx - (float)(bx + i), kb);   // src/tron.cu:516
                        s_wy[rec * C::NW + i] = kb_weight<KB>(ky - (float)(by + i), kb);
                    }
                    const int ridx = (r * p.nro) / n;                 // src/tron.cu:517 (truncating)
                    const int ro = ridx + p.nro / 2;
                    float sdc = 1.0f;
                    if (p.apply_dcf)                                  // src/tron.cu:412
                        sdc = p.dcf_a * fabsf((float)ro - (float)(p.nro / 2)) + p.dcf_b;
                    const size_t sbase = ((size_t)p.nro * e.pe + ro) * p.nchan + c0;
#pragma unroll
                    for (int c = 0; c < CPB; ++c) {
                        float2 d = make_float2(0.f, 0.f);
                        if (c < ncb) {
                            d = load_sample<HALF>(in_bytes, sbase + c);
                            d.x *= sdc; d.y *= sdc;                   // src/tron.cu:414
                        }
                        s_d[rec * CPB + c] = d;
                    }
                }
            }
            __syncthreads();
            // ---- 3. gather: one thread per Cartesian point ----------------------------
            for (int s = 0; s < nsp; ++s) {
                const BatchSpoke bs = s_sp[s];
                const float q = Yf * bs.ct - Xf * bs.st;
                if (fabsf(q) < bs.reach) {
                    const float P = bs.use_x ? Xf : Yf;
                    const float ta = (P - (float)CW - eps) * bs.inv;
                    const float tb = (P + (float)CW + eps) * bs.inv;
                    const int ca = max((int)ceilf(fminf(ta, tb)), bs.rlo);
                    const int cb = min((int)floorf(fmaxf(ta, tb)), bs.rhi);
                    // reference order: aligned radii ascending, then anti-aligned ascending
                    // (src/tron.cu:512,521); r = 0 is met twice when Rlo == 0, as there.
#pragma unroll
                    for (int pass = 0; pass < 2; ++pass) {
                        const int a = pass == 0 ? max(ca, Rlo) : max(ca, -Rhi);
                        const int b = pass == 0 ? min(cb, Rhi) : min(cb, -Rlo);
                        for (int r = a; r <= b; ++r) {
                            const int rec = s * C::LPS + (r - bs.rlo);
                            const uint32_t bb = s_b[rec];
                            const int i = X - (int)(short)(bb & 0xffffu);
                            const int j = Y - (int)(short)(bb >> 16);
                            if ((unsigned)i < (unsigned)C::NW && (unsigned)j < (unsigned)C::NW) {
                                const float wgt = s_wx[rec * C::NW + i] * s_wy[rec * C::NW + j];
                                if (wgt > 0.f) {                      // src/tron.cu:518
#pragma unroll
                                    for (int c = 0; c < CPB; ++c) {
                                        const float2 d = s_d[rec * CPB + c];
                                        if (KB == TRON_KB_EXACT) {
                                            acc[c].x += d.x * wgt;    // src/tron.cu:519, unfused
                                            acc[c].y += d.y * wgt;
                                        } else {
                                            acc[c].x = fmaf(d.x, wgt, acc[c].x);
                                            acc[c].y = fmaf(d.y, wgt, acc[c].y);
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    if (inside) {
        int row = Y + h, col = X + h;
        if (p.out_shift) {      // both fftshifts of src/tron.cu:631 folded into the store index
            row = Y < 0 ? Y + n : Y;
            col = X < 0 ? X + n : X;
        }
        float2 *out = p.udata + (size_t)z * p.out_z + ((size_t)row * n + col) * p.out_p;
#pragma unroll
        for (int c = 0; c < CPB; ++c)
            if (c < ncb) {
                float2 v;
                v.x = acc[c].x * p.scale;                              // src/tron.cu:532-534
                v.y = acc[c].y * p.scale;
                out[(size_t)(c0 + c) * p.out_c] = v;
            }
    }
}

template <int CPB, int CW>
static hipError_t launch_grid_kb(const GridParams &p, int kb_mode, int half_in, hipStream_t s)
{
    const int chunks = (p.nchan - p.coil0 + CPB - 1) / CPB;
    dim3 grid((unsigned)((size_t)p.ntiles * p.nslices), (unsigned)chunks);
    const size_t lds = grid_lds_bytes(CPB, CW);
#define TRON_LAUNCH(KBM, HF)                                                                   \
    do {                                                                                       \
        auto kern = grid_tile_kernel<CPB, CW, KBM, HF>;                                        \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),              \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e_ != hipSuccess) return e_;                                                       \
        hipLaunchKernelGGL(kern, grid, dim3(kGridThreads), lds, s, p);                         \
    } while (0)
    if (kb_mode == TRON_KB_EXACT) { if (half_in) TRON_LAUNCH(TRON_KB_EXACT, true); else TRON_LAUNCH(TRON_KB_EXACT, false); }
    else                          { if (half_in) TRON_LAUNCH(TRON_KB_FAST, true);  else TRON_LAUNCH(TRON_KB_FAST, false); }
#undef TRON_LAUNCH
    return hipGetLastError();
}

template <int CW>
static hipError_t launch_grid_cw(const GridParams &p, int kb_mode, int half_in, hipStream_t s)
{
    const int nc = p.nchan - p.coil0;
    if (nc >= 8 && nc % 8 == 0) return launch_grid_kb<8, CW>(p, kb_mode, half_in, s);
    if (nc >= 4) return launch_grid_kb<4, CW>(p, kb_mode, half_in, s);
    if (nc >= 2) return launch_grid_kb<2, CW>(p, kb_mode, half_in, s);
    return launch_grid_kb<1, CW>(p, kb_mode, half_in, s);
}

hipError_t launch_grid(const GridParams &p, int kb_mode, int half_in, hipStream_t s)
{
    const int cw = (int)ceilf(p.W);
    switch (cw) {
        case 1: return launch_grid_cw<1>(p, kb_mode, half_in, s);
        case 2: return launch_grid_cw<2>(p, kb_mode, half_in, s);
        case 3: return launch_grid_cw<3>(p, kb_mode, half_in, s);
        case 4: return launch_grid_cw<4>(p, kb_mode, half_in, s);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------- adjoint tail

// = fftshift + crop + deapodkernel (+ coilcombinesos): src/tron.cu:633-635, 764.  Reads only
// the centre nx^2 of the FFT output; the shift is an index rotation, the deapodisation a table.
__global__ void __launch_bounds__(256) post_kernel(const PostParams p)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = blockIdx.y;
    if (id >= p.nx * p.nx) return;
    const int n = p.nxos;
    const int w = (n - p.nx) / 2;                                     // src/tron.cu:422
    const int row = id / p.nx, col = id % p.nx;
    const int mr = (row + w - n / 2 + n) % n;                         // undo fftshift(FORWARD), src/tron.cu:164-172
    const int mc = (col + w - n / 2 + n) % n;
    const float inv = p.inv_deapod[id];                               // src/tron.cu:398-400
    const float2 *src = p.fft + ((size_t)z * p.nchan) * n * n + (size_t)mr * n + mc;
    if (p.combine && p.nchan > 1) {
        float val = 0.f;
        for (int c = 0; c < p.nchan; ++c) {
            float2 v = src[(size_t)c * n * n];
            v.x *= inv; v.y *= inv;
            val += v.x * v.x + v.y * v.y;                             // src/tron.cu:262
        }
        p.out[(size_t)z * p.nx * p.nx + id] = make_float2(sqrtf(val), 0.f);
    } else if (p.combine) {
        float2 v = src[0];
        v.x *= inv; v.y *= inv;
        p.out[(size_t)z * p.nx * p.nx + id] = v;                      // src/tron.cu:266
    } else {
        float2 *dst = p.out + ((size_t)z * p.nx * p.nx + id) * p.nchan;
        for (int c = 0; c < p.nchan; ++c) {
            float2 v = src[(size_t)c * n * n];
            v.x *= inv; v.y *= inv;
            dst[c] = v;
        }
    }
}

hipError_t launch_post(const PostParams &p, hipStream_t s)
{
    dim3 grid((p.nx * p.nx + 255) / 256, p.nslices);
    hipLaunchKernelGGL(post_kernel, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- forward head

// = pad + deapodkernel(n = nxos, sigma = 1) + fftshift(FORWARD): src/tron.cu:642-644.
__global__ void __launch_bounds__(256) pre_kernel(const PreParams p)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    const int n = p.nxos;
    if (id >= n * n) return;
    const int w = n > p.nx ? (n - p.nx) / 2 : 0;                      // src/tron.cu:439
    const int xdst = id / n, ydst = id % n;
    const bool in = (xdst - w > 0) && (xdst - w < p.nx) && (ydst - w > 0) && (ydst - w < p.nx);   // src/tron.cu:449-450
    const float inv = p.inv_deapod[id];
    const int sr = (xdst + n / 2) % n, sc = (ydst + n / 2) % n;       // src/tron.cu:164-172
    const float2 *src = p.img + (size_t)k * p.nchan * p.nx * p.nx;
    float2 *dst = p.fft + ((size_t)k * p.nchan) * n * n + (size_t)sr * n + sc;
    for (int c = 0; c < p.nchan; ++c) {
        float2 v = make_float2(0.f, 0.f);
        if (in) v = src[((size_t)(xdst - w) * p.nx + (ydst - w)) * p.nchan + c];
        v.x *= inv; v.y *= inv;
        dst[(size_t)c * n * n] = v;
    }
}

hipError_t launch_pre(const PreParams &p, hipStream_t s)
{
    dim3 grid((p.nxos * p.nxos + 255) / 256, p.nimg);
    hipLaunchKernelGGL(pre_kernel, grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- degridding

// = degridradial2d, src/tron.cu:540-577: one thread per k-space sample, accumulation order
// (xu outer, yu inner) and arithmetic as in the reference.
template <int CPB, int KB>
__global__ void __launch_bounds__(256) degrid_kernel(const DegridParams p)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (id >= p.nro * p.npe) return;
    const int n = p.n;
    const int pe = id / p.nro;
    const int ro = id % p.nro;
    KbCoef kb;
    kb.W = p.W; kb.beta = p.beta; kb.terms = p.kb_terms; kb.poly = p.kb_poly;
    const float W = p.W;
    const float R = (float)ro / (float)p.nro - 0.5f;                  // src/tron.cu:554
    const float2 cs = p.trig[pe];
    float X = cs.y, Y = cs.x;                                         // X = sin, Y = cos (src/tron.cu:559)
    X = (float)n * R * X + (float)((n + 1) / 2);                      // src/tron.cu:560-561
    Y = (float)n * R * Y + (float)((n + 1) / 2);
    const float2 *src = p.udata + (size_t)k * p.in_z;
    float2 *dst = p.nudata + ((size_t)k * p.nro * p.npe + id) * p.nrep;
    for (int c0 = 0; c0 < p.nrep; c0 += CPB) {
        float2 acc[CPB];
#pragma unroll
        for (int c = 0; c < CPB; ++c) acc[c] = make_float2(0.f, 0.f);
        for (int xu = (int)ceilf(X - W); (float)xu <= (X + W); ++xu) {
            const float wgtx = kb_weight<KB>((float)xu - X, kb);
            for (int yu = (int)ceilf(Y - W); (float)yu <= (Y + W); ++yu) {
                const float wgt = wgtx * kb_weight<KB>((float)yu - Y, kb);
                int i = (xu + n) % n;                                 // src/tron.cu:569-570
                int j = (yu + n) % n;
                if (p.in_shift) { i = (i + n / 2) % n; j = (j + n / 2) % n; }   // fftshift(INVERSE) of :646 folded in
                const float2 *u = src + ((size_t)i * n + j) * p.in_p;
#pragma unroll
                for (int c = 0; c < CPB; ++c)
                    if (c0 + c < p.nrep) {
                        const float2 v = u[(size_t)(c0 + c) * p.in_c];
                        acc[c].x += v.x * wgt;                        // src/tron.cu:573, unfused
                        acc[c].y += v.y * wgt;
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < CPB; ++c)
            if (c0 + c < p.nrep) dst[c0 + c] = acc[c];
    }
}

hipError_t launch_degrid(const DegridParams &p, int kb_mode, hipStream_t s)
{
    dim3 grid((p.nro * p.npe + 255) / 256, p.nimg);
    if (p.nrep >= 4) {
        if (kb_mode == TRON_KB_EXACT) hipLaunchKernelGGL((degrid_kernel<4, TRON_KB_EXACT>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((degrid_kernel<4, TRON_KB_FAST>), grid, dim3(256), 0, s, p);
    } else {
        if (kb_mode == TRON_KB_EXACT) hipLaunchKernelGGL((degrid_kernel<1, TRON_KB_EXACT>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((degrid_kernel<1, TRON_KB_FAST>), grid, dim3(256), 0, s, p);
    }
    return hipGetLastError();
}

}  // namespace tron
