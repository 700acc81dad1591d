// Coil combination (src/tron.cu:222-302) and the vector kernels of the CGNR path (src/tron.cu:658-720): Caxpy (:658-663), the norms / dot products the reference takes
// from cuBLAS (:695-696,705,708), the density weights of precompensate (:405-416) inside the weighted dot product, and
// coilcombinesos (:255-268) on plain coil images.  Everything is batched over the slices of a launch (one set of CG
// scalars per slice, kept on the device) and runs on the plan's stream: no host round trip inside an iteration.
#include <string.h>

#include <algorithm>

#include "tron_device.h"

namespace tron {

constexpr int kCgThreads = 256;
constexpr int kCgBlocks = 64;          // partial sums per slice; summed in index order by cg_finish_kernel (deterministic)

__device__ __forceinline__ double block_sum(double v, double *sm)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < kCgThreads / 64; ++w) t += sm[w];
    return t;                            // valid in thread 0
}

// x *= scale (in place), partial[z][b] = sum |x|^2 of block b's share of slice z.  grid = (kCgBlocks, nslices)
__global__ void __launch_bounds__(kCgThreads) cg_scale_norm2_kernel(float2 *x, size_t n, float scale, double *partial)
{
    __shared__ double sm[kCgThreads / 64];
    float2 *xs = x + (size_t)blockIdx.y * n;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * kCgThreads + threadIdx.x; i < n; i += (size_t)kCgBlocks * kCgThreads) {
        float2 v = xs[i];
        v.x *= scale; v.y *= scale;
        xs[i] = v;
        acc += (double)v.x * v.x + (double)v.y * v.y;
    }
    const double t = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * kCgBlocks + blockIdx.x] = t;
}

// Sum of a slice's partial sums in a FIXED order (lane l takes entries l, l + 64, ...; then a shuffle tree): deterministic
// run to run, whichever workgroups produced the entries and in whatever order.  All 64 lanes of the calling wave take part.
__device__ __forceinline__ double ordered_sum(const double *part, int nparts)
{
    const int lane = threadIdx.x & 63;
    double t = 0.0;
    for (int b = lane; b < nparts; b += 64) t += part[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    return t;
}

// What a finished norm means for the iteration (Knopp, Kunis, Potts 2007, Alg. 1 as wired at src/tron.cu:686-712; F1):
// mode 0: num[z] = t; 1: alpha[z] = num[z] / t (t = <W v, v>); 2: beta[z] = t / num[z], then num[z] = t (t = |ztilde|^2)
__device__ __forceinline__ void cg_apply(int mode, double t, double *num, float *coef, int z)
{
    if (mode == 0) num[z] = t;
    else if (mode == 1) coef[z] = (float)(num[z] / t);
    else { coef[z] = (float)(t / num[z]); num[z] = t; }
}

// partial[z][b] = sum over block b's share of Re <W v, v>, W = a*|ro - nro/2| + b per sample (src/tron.cu:408-414):
// v[nchan*(ro + nro*pe) + c]; the weighted value is rounded to float first, as precompensate stores it (:414).  cg_finish_kernel sums the partials in a fixed order and turns them
// into the step size.  (A last-workgroup-done ending inside this kernel -- counter add after a __threadfence() -- was built and
// measured: the agent-scope release writes the XCD's L2 back, dirty with the 211 MB the degridding kernel has just stored, once
// per workgroup: 96 us at 64 workgroups per slice, 411 us at 512, against 6 us for the separate one-wave-per-slice finish.)
__global__ void __launch_bounds__(kCgThreads) cg_wnorm2_kernel(const float2 *v, unsigned nelem, int nchan, int nro, float dcf_a, float dcf_b,
                                                              double *partial)
{
    __shared__ double sm[kCgThreads / 64];
    const int z = blockIdx.y;
    const float2 *vs = v + (size_t)z * nelem;
    double acc = 0.0;
    if ((nchan & 1) == 0) {
        // two values (16 bytes) per thread and step: both belong to one sample, lanes read consecutive 16-byte pieces
        const float4 *v4 = reinterpret_cast<const float4 *>(vs);
        const unsigned stride = gridDim.x * kCgThreads;
        double acc2 = 0.0;
        auto term = [&](unsigned q, const float4 e) {
            const int ro = (int)(((2u * q) / (unsigned)nchan) % (unsigned)nro);
            const float sdc = dcf_a * fabsf((float)ro - (float)(nro / 2)) + dcf_b;
            const float ax = e.x * sdc, ay = e.y * sdc, bx = e.z * sdc, by = e.w * sdc;
            return ((double)ax * e.x + (double)ay * e.y) + ((double)bx * e.z + (double)by * e.w);
        };
        unsigned q = blockIdx.x * kCgThreads + threadIdx.x;
        for (; q + stride < nelem / 2; q += 2 * stride) {               // two loads in flight, two accumulation chains
            const float4 e0 = v4[q], e1 = v4[q + stride];
            acc += term(q, e0);
            acc2 += term(q + stride, e1);
        }
        if (q < nelem / 2) acc += term(q, v4[q]);
        acc += acc2;
    } else {
        for (unsigned i = blockIdx.x * kCgThreads + threadIdx.x; i < nelem; i += gridDim.x * kCgThreads) {
            const int ro = (int)((i / (unsigned)nchan) % (unsigned)nro);
            const float sdc = dcf_a * fabsf((float)ro - (float)(nro / 2)) + dcf_b;
            const float2 e = vs[i];
            const float ux = e.x * sdc, uy = e.y * sdc;
            acc += (double)ux * e.x + (double)uy * e.y;
        }
    }
    const double t = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[(size_t)z * gridDim.x + blockIdx.x] = t;
}

// The ending of every reduction (this file's norm kernels, the fused FFT tail's |ztilde|^2; nparts per slice): one wave per slice.  grid = nslices, block = 64
__global__ void __launch_bounds__(64) cg_finish_kernel(const double *partial, int nparts, double *num, float *coef, int mode)
{
    const int z = blockIdx.x;
    const double t = ordered_sum(partial + (size_t)z * nparts, nparts);
    if (threadIdx.x == 0) cg_apply(mode, t, num, coef, z);
}

// r = y (src/tron.cu:685): slice z's window of the spoke stream, n values from y + z * hop -- windows overlap, so not a 2-D copy
__global__ void __launch_bounds__(kCgThreads) cg_windows_kernel(float2 *r, const float2 *y, size_t n, size_t hop)
{
    float2 *rs = r + (size_t)blockIdx.y * n;
    const float2 *ys = y + (size_t)blockIdx.y * hop;
    for (size_t i = (size_t)blockIdx.x * kCgThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kCgThreads) rs[i] = ys[i];
}

// Caxpy, src/tron.cu:658-663, per slice: y = y + sign*coef[z]*x (unfused, like the reference's operator chain)
__global__ void __launch_bounds__(kCgThreads) cg_axpy_kernel(float2 *y, const float2 *x, const float *coef, float sign, size_t n)
{
    const float a = sign * coef[blockIdx.y];
    float4 *ys = reinterpret_cast<float4 *>(y + (size_t)blockIdx.y * n);
    const float4 *xs = reinterpret_cast<const float4 *>(x + (size_t)blockIdx.y * n);
    for (size_t i = (size_t)blockIdx.x * kCgThreads + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * kCgThreads) {
        const float4 q = xs[i];
        float4 r = ys[i];
        r.x = r.x + a * q.x; r.y = r.y + a * q.y; r.z = r.z + a * q.z; r.w = r.w + a * q.w;
        ys[i] = r;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        float2 *yl = y + (size_t)blockIdx.y * n + (n - 1);
        const float2 q = x[(size_t)blockIdx.y * n + (n - 1)];
        yl->x = yl->x + a * q.x; yl->y = yl->y + a * q.y;
    }
}

// The two image-space updates of an iteration in ONE pass over ptilde: x += alpha * ptilde (src/tron.cu:699, with the alpha of
// this iteration) and ptilde = ztilde + beta * ptilde (:710) -- ptilde is read once instead of twice.  last: only x is updated.
__global__ void __launch_bounds__(kCgThreads) cg_update_kernel(float2 *x, float2 *pt, const float2 *zt, const float *alpha, const float *beta,
                                                              size_t n, int last)
{
    const float a = alpha[blockIdx.y], b = last ? 0.f : beta[blockIdx.y];
    float2 *xs = x + (size_t)blockIdx.y * n, *ps = pt + (size_t)blockIdx.y * n;
    const float2 *zs = zt + (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * kCgThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kCgThreads) {
        const float2 pv = ps[i];
        float2 xv = xs[i];
        xv.x = xv.x + a * pv.x; xv.y = xv.y + a * pv.y;
        xs[i] = xv;
        if (!last) {
            const float2 q = zs[i];
            ps[i] = make_float2(q.x + b * pv.x, q.y + b * pv.y);
        }
    }
}

// coilcombinesos, src/tron.cu:255-268, on coil images [z][nchan*id + c], channel = c + nc*t: per repetition t,
// nc > 1 -> (sqrt(sum_c |.|^2), 0); nc == 1 -> copy.  out[z][nt*id + t] (.ra dims [1, nt, nx, ny, nz]).
// The reference passes nc and ignores nt (:764), which is only coherent for nt = 1; this is that kernel per repetition.
__global__ void __launch_bounds__(kCgThreads) sos_kernel(float2 *out, const float2 *coil, size_t npix, int nc, int nt)
{
    const size_t i = (size_t)blockIdx.x * kCgThreads + threadIdx.x;
    if (i >= npix * nt) return;
    const size_t id = i / nt;
    const int t = (int)(i % nt);
    const float2 *c = coil + ((size_t)blockIdx.y * npix + id) * ((size_t)nc * nt) + (size_t)nc * t;
    float2 *o = out + (size_t)blockIdx.y * npix * nt + i;
    if (nc == 1) { *o = c[0]; return; }
    float val = 0.f;
    for (int k = 0; k < nc; ++k) val += c[k].x * c[k].x + c[k].y * c[k].y;     // norm() of float2math.h
    *o = make_float2(sqrtf(val), 0.f);
}

__device__ __forceinline__ float2 cmulf(const float2 a, const float2 b)       // float2math.h:36-40
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// coilcombinewalsh + powit, src/tron.cu:222-253,270-302: per pixel the coil covariance over a (2*npatch+1)^2 patch
// (clipped at the image border), its dominant eigenvector by 5 power iterations from (1,...,1), image = sum_c conj(v_c) coil_c.
// Same operation order as the reference (unfused).  The matrix holds nc*nc entries (the reference: MAXCHAN^2 = 36).
constexpr int kWalshMaxCoils = 16;
__global__ void __launch_bounds__(64) walsh_kernel(float2 *out, const float2 *coil, int nimg, int nc, int nt, int npatch)
{
    const size_t npix = (size_t)nimg * nimg;
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= npix * nt) return;
    const size_t id = i / nt;
    const int t = (int)(i % nt);
    const int nchan = nc * nt;
    const float2 *ci = coil + (size_t)blockIdx.y * npix * nchan + (size_t)nc * t;
    float2 *o = out + (size_t)blockIdx.y * npix * nt + i;
    if (nc == 1) { *o = ci[(size_t)nchan * id]; return; }
    float2 A[kWalshMaxCoils * kWalshMaxCoils], x[kWalshMaxCoils], y[kWalshMaxCoils];
    const int px = (int)(id / nimg), py = (int)(id % nimg);                     // :279-280 (x = id / nimg)
    for (int k = 0; k < nc * nc; ++k) A[k] = make_float2(0.f, 0.f);
    for (int qx = max(0, px - npatch); qx <= min(nimg - 1, px + npatch); ++qx)
        for (int qy = max(0, py - npatch); qy <= min(nimg - 1, py + npatch); ++qy) {
            const float2 *q = ci + (size_t)nchan * ((size_t)qx * nimg + qy);
            for (int c2 = 0; c2 < nc; ++c2)
                for (int c1 = 0; c1 < nc; ++c1) {
                    const float2 m = cmulf(q[c1], make_float2(q[c2].x, -q[c2].y));
                    A[c1 * nc + c2].x += m.x; A[c1 * nc + c2].y += m.y;         // :289
                }
        }
    for (int k = 0; k < nc; ++k) x[k] = make_float2(1.f, 0.f);                  // powit, :226-227
    for (int it = 0; it < 5; ++it) {
        for (int j = 0; j < nc; ++j) {
            y[j] = make_float2(0.f, 0.f);
            for (int k = 0; k < nc; ++k) { const float2 m = cmulf(A[j * nc + k], x[k]); y[j].x += m.x; y[j].y += m.y; }
        }
        float norm_sq = 0.f;
        for (int k = 0; k < nc; ++k) norm_sq += y[k].x * y[k].x + y[k].y * y[k].y;
        norm_sq = sqrtf(norm_sq);
        const float inv = 1.0f / norm_sq;                                        // operator/ (float2, float), float2math.h:24-28
        for (int k = 0; k < nc; ++k) x[k] = make_float2(y[k].x * inv, y[k].y * inv);
    }
    float2 acc = make_float2(0.f, 0.f);
    for (int c = 0; c < nc; ++c) {
        const float2 m = cmulf(make_float2(x[c].x, -x[c].y), ci[(size_t)nchan * id + c]);   // :295
        acc.x += m.x; acc.y += m.y;
    }
    *o = acc;
}

// The same per 16x16 pixel tile, for NC = 2..8 coils and patches up to 9x9: the (16 + 2 npatch)^2 x NC coil values the tile's
// patches touch are staged ONCE in LDS, coil-planar (a wave's reads of one coil are 64 consecutive values: no bank conflicts);
// the covariance matrix is Hermitian term by term (q1 conj(q2) and q2 conj(q1) are exact conjugates in fp32), so a thread keeps
// its upper triangle -- NC (NC + 1) / 2 complex values -- and the two vectors of the power iteration in REGISTERS (the kernel
// above indexes private arrays with run-time coil counts: 2.3 KB of scratch per pixel, and re-reads each patch from HBM / L2).
// Same operations, same order, same unfused arithmetic per matrix entry as src/tron.cu:284-295.
template <int NC>
__global__ void __launch_bounds__(256) walsh_tile_kernel(float2 *out, const float2 *coil, int nimg, int nt, int npatch)
{
    constexpr int T = 16;
    extern __shared__ __align__(16) float2 s_c[];               // [coil][(T + 2 npatch)^2]
    const int side = T + 2 * npatch, plane = side * side;
    const int tiles = (nimg + T - 1) / T;
    const int tx = (int)(blockIdx.x / tiles) * T, ty = (int)(blockIdx.x % tiles) * T;   // tile origin (px = slow index, src/tron.cu:279-280)
    const int z = (int)blockIdx.y / nt, t = (int)blockIdx.y % nt;
    const int nchan = NC * nt;
    const size_t npix = (size_t)nimg * nimg;
    const float2 *ci = coil + (size_t)z * npix * nchan + (size_t)NC * t;
    for (int e = threadIdx.x; e < plane * NC; e += 256) {
        const int c = e % NC, pix = e / NC;                     // consecutive threads: the NC values of one pixel, then the next pixel of the row
        const int gx = tx - npatch + pix / side, gy = ty - npatch + pix % side;
        float2 v = make_float2(0.f, 0.f);
        if (gx >= 0 && gx < nimg && gy >= 0 && gy < nimg) v = ci[(size_t)nchan * ((size_t)gx * nimg + gy) + c];
        s_c[c * plane + pix] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x / T, ly = threadIdx.x % T;
    const int px = tx + lx, py = ty + ly;
    if (px >= nimg || py >= nimg) return;
    float2 A[NC * (NC + 1) / 2];                                // A[c1][c2], c1 <= c2, at c1 * NC - c1 (c1 - 1) / 2 + (c2 - c1)
#pragma unroll
    for (int k = 0; k < NC * (NC + 1) / 2; ++k) A[k] = make_float2(0.f, 0.f);
#pragma unroll 1
    for (int dx = -npatch; dx <= npatch; ++dx)
#pragma unroll 1
        for (int dy = -npatch; dy <= npatch; ++dy) {
            const int qx = px + dx, qy = py + dy;
            if (qx < 0 || qx >= nimg || qy < 0 || qy >= nimg) continue;     // the patch is clipped at the image border (:284-285)
            const int pix = (lx + npatch + dx) * side + (ly + npatch + dy);
            float2 q[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) q[c] = s_c[c * plane + pix];
#pragma unroll
            for (int c1 = 0; c1 < NC; ++c1)
#pragma unroll
                for (int c2 = c1; c2 < NC; ++c2) {
                    const float2 m = cmulf(q[c1], make_float2(q[c2].x, -q[c2].y));           // :289
                    float2 &a = A[c1 * NC - c1 * (c1 - 1) / 2 + (c2 - c1)];
                    a.x += m.x; a.y += m.y;
                }
        }
    auto entry = [&](int j, int k) {                            // A[j][k]; below the diagonal the conjugate of A[k][j]
        if (j <= k) return A[j * NC - j * (j - 1) / 2 + (k - j)];
        const float2 u = A[k * NC - k * (k - 1) / 2 + (j - k)];
        return make_float2(u.x, -u.y);
    };
    float2 x[NC], y[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) x[k] = make_float2(1.f, 0.f);                  // powit, :226-227
#pragma unroll 1
    for (int it = 0; it < 5; ++it) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            y[j] = make_float2(0.f, 0.f);
#pragma unroll
            for (int k = 0; k < NC; ++k) { const float2 m = cmulf(entry(j, k), x[k]); y[j].x += m.x; y[j].y += m.y; }
        }
        float norm_sq = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) norm_sq += y[k].x * y[k].x + y[k].y * y[k].y;
        norm_sq = sqrtf(norm_sq);
        const float inv = 1.0f / norm_sq;                                        // operator/ (float2, float), float2math.h:24-28
#pragma unroll
        for (int k = 0; k < NC; ++k) x[k] = make_float2(y[k].x * inv, y[k].y * inv);
    }
    float2 acc = make_float2(0.f, 0.f);
    const int pix0 = (lx + npatch) * side + (ly + npatch);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float2 m = cmulf(make_float2(x[c].x, -x[c].y), s_c[c * plane + pix0]);         // :295
        acc.x += m.x; acc.y += m.y;
    }
    out[(size_t)z * npix * nt + ((size_t)px * nimg + py) * nt + t] = acc;
}

template <int NC>
static void launch_walsh_tile(float2 *out, const float2 *coil, int nimg, int nt, int npatch, int nslices, hipStream_t s)
{
    const int tiles = (nimg + 15) / 16, side = 16 + 2 * npatch;
    hipLaunchKernelGGL((walsh_tile_kernel<NC>), dim3((unsigned)(tiles * tiles), (unsigned)(nslices * nt)), dim3(256),
                       (size_t)NC * side * side * sizeof(float2), s, out, coil, nimg, nt, npatch);
}

hipError_t launch_cg_scale_norm2(float2 *x, size_t n, int nslices, float scale, double *partial, hipStream_t s)
{
    hipLaunchKernelGGL(cg_scale_norm2_kernel, dim3(kCgBlocks, nslices), dim3(kCgThreads), 0, s, x, n, scale, partial);
    return hipGetLastError();
}
hipError_t launch_cg_wnorm2(const float2 *v, size_t n, int nslices, int nchan, int nro, float a, float b, double *partial, int parts_cap,
                            int *nparts, hipStream_t s)
{
    if (n >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    // enough workgroups to keep HBM busy when the slices are few: at least 8192 in all, kCgBlocks..parts_cap per slice
    const int nblk = std::max(kCgBlocks, std::min(parts_cap, (8192 + nslices - 1) / nslices));
    *nparts = nblk;
    hipLaunchKernelGGL(cg_wnorm2_kernel, dim3(nblk, nslices), dim3(kCgThreads), 0, s, v, (unsigned)n, nchan, nro, a, b, partial);
    return hipGetLastError();
}
hipError_t launch_cg_windows(float2 *r, const float2 *y, size_t n, size_t hop, int nslices, hipStream_t s)
{
    const unsigned nb = (unsigned)std::min<size_t>(512, (n + kCgThreads - 1) / kCgThreads);
    hipLaunchKernelGGL(cg_windows_kernel, dim3(std::max(nb, 1u), nslices), dim3(kCgThreads), 0, s, r, y, n, hop);
    return hipGetLastError();
}
hipError_t launch_cg_finish(const double *partial, int nparts, double *num, float *coef, int mode, int nslices, hipStream_t s)
{
    hipLaunchKernelGGL(cg_finish_kernel, dim3(nslices), dim3(64), 0, s, partial, nparts, num, coef, mode);
    return hipGetLastError();
}
hipError_t launch_cg_axpy(float2 *y, const float2 *x, const float *coef, float sign, size_t n, int nslices, hipStream_t s)
{
    const unsigned nb = (unsigned)std::min<size_t>(512, (n / 2 + kCgThreads - 1) / kCgThreads);
    hipLaunchKernelGGL(cg_axpy_kernel, dim3(std::max(nb, 1u), nslices), dim3(kCgThreads), 0, s, y, x, coef, sign, n);
    return hipGetLastError();
}
hipError_t launch_cg_update(float2 *x, float2 *pt, const float2 *zt, const float *alpha, const float *beta, size_t n, int nslices, int last, hipStream_t s)
{
    const unsigned nb = (unsigned)std::min<size_t>(512, (n + kCgThreads - 1) / kCgThreads);
    hipLaunchKernelGGL(cg_update_kernel, dim3(nb, nslices), dim3(kCgThreads), 0, s, x, pt, zt, alpha, beta, n, last);
    return hipGetLastError();
}
// mode 0: root-sum-of-squares, 1: Walsh (nc <= kWalshMaxCoils)
hipError_t launch_coil_combine(float2 *out, const float2 *coil, int nimg, int nc, int nt, int mode, int npatch, int nslices, hipStream_t s)
{
    const size_t work = (size_t)nimg * nimg * nt;
    if (mode == 1) {
        if (nc > kWalshMaxCoils) return hipErrorInvalidValue;
        if (nc >= 2 && nc <= 8 && npatch <= 4) {
            switch (nc) {
                case 2: launch_walsh_tile<2>(out, coil, nimg, nt, npatch, nslices, s); break;
                case 3: launch_walsh_tile<3>(out, coil, nimg, nt, npatch, nslices, s); break;
                case 4: launch_walsh_tile<4>(out, coil, nimg, nt, npatch, nslices, s); break;
                case 5: launch_walsh_tile<5>(out, coil, nimg, nt, npatch, nslices, s); break;
                case 6: launch_walsh_tile<6>(out, coil, nimg, nt, npatch, nslices, s); break;
                case 7: launch_walsh_tile<7>(out, coil, nimg, nt, npatch, nslices, s); break;
                default: launch_walsh_tile<8>(out, coil, nimg, nt, npatch, nslices, s); break;
            }
            return hipGetLastError();
        }
        hipLaunchKernelGGL(walsh_kernel, dim3((unsigned)((work + 63) / 64), nslices), dim3(64), 0, s, out, coil, nimg, nc, nt, npatch);
    } else {
        hipLaunchKernelGGL(sos_kernel, dim3((unsigned)((work + kCgThreads - 1) / kCgThreads), nslices), dim3(kCgThreads), 0, s,
                           out, coil, (size_t)nimg * nimg, nc, nt);
    }
    return hipGetLastError();
}

__global__ void warm_cgnr_tu() {}
hipError_t warm_cgnr()
{
    hipLaunchKernelGGL(warm_cgnr_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
