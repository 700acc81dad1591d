// Coil combination (src/tron.cu:222-302) and the vector kernels of the CGNR path (src/tron.cu:658-720): Caxpy (:658-663), the norms / dot products the reference takes
// from cuBLAS (:695-696,705,708), the density weights of precompensate (:405-416) inside the weighted dot product, and
// coilcombinesos (:255-268) on plain coil images.  Everything is batched over the slices of a launch (one set of CG
// scalars per slice, kept on the device) and runs on the plan's stream: no host round trip inside an iteration.
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

// partial[z][b] = sum over block b's share of Re <W v, v>, W = a*|ro - nro/2| + b per sample (src/tron.cu:408-414):
// v[nchan*(ro + nro*pe) + c]; the weighted value is rounded to float first, as precompensate stores it (:414).
__global__ void __launch_bounds__(kCgThreads) cg_wnorm2_kernel(const float2 *v, size_t n, int nchan, int nro, float dcf_a, float dcf_b, double *partial)
{
    __shared__ double sm[kCgThreads / 64];
    const float2 *vs = v + (size_t)blockIdx.y * n;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * kCgThreads + threadIdx.x; i < n; i += (size_t)kCgBlocks * kCgThreads) {
        const int ro = (int)((i / (size_t)nchan) % (size_t)nro);
        const float sdc = dcf_a * fabsf((float)ro - (float)(nro / 2)) + dcf_b;
        const float2 q = vs[i];
        const float ux = q.x * sdc, uy = q.y * sdc;
        acc += (double)ux * q.x + (double)uy * q.y;
    }
    const double t = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * kCgBlocks + blockIdx.x] = t;
}

// out[z] = sum_b partial[z][b], in index order.  mode 0: store; 1: alpha[z] = num[z] / out (out = vwv);
// 2: beta[z] = out / num[z], then num[z] = out (out = new |ztilde|^2).  grid = nslices, block = 64
__global__ void cg_finish_kernel(const double *partial, double *num, float *coef, int mode)
{
    if (threadIdx.x != 0) return;
    const int z = blockIdx.x;
    double t = 0.0;
    for (int b = 0; b < kCgBlocks; ++b) t += partial[(size_t)z * kCgBlocks + b];
    if (mode == 0) num[z] = t;
    else if (mode == 1) coef[z] = (float)(num[z] / t);
    else { coef[z] = (float)(t / num[z]); num[z] = t; }
}

// Caxpy, src/tron.cu:658-663, per slice: y = y + sign*coef[z]*x (unfused, like the reference's operator chain)
__global__ void __launch_bounds__(kCgThreads) cg_axpy_kernel(float2 *y, const float2 *x, const float *coef, float sign, size_t n)
{
    const float a = sign * coef[blockIdx.y];
    float2 *ys = y + (size_t)blockIdx.y * n;
    const float2 *xs = x + (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * kCgThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kCgThreads) {
        const float2 q = xs[i];
        float2 r = ys[i];
        r.x = r.x + a * q.x; r.y = r.y + a * q.y;
        ys[i] = r;
    }
}

// ptilde = ztilde + beta*ptilde (src/tron.cu:710)
__global__ void __launch_bounds__(kCgThreads) cg_xpby_kernel(float2 *pt, const float2 *zt, const float *coef, size_t n)
{
    const float b = coef[blockIdx.y];
    float2 *ps = pt + (size_t)blockIdx.y * n;
    const float2 *zs = zt + (size_t)blockIdx.y * n;
    for (size_t i = (size_t)blockIdx.x * kCgThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kCgThreads) {
        const float2 q = zs[i];
        float2 r = ps[i];
        r.x = q.x + b * r.x; r.y = q.y + b * r.y;
        ps[i] = r;
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

hipError_t launch_cg_scale_norm2(float2 *x, size_t n, int nslices, float scale, double *partial, hipStream_t s)
{
    hipLaunchKernelGGL(cg_scale_norm2_kernel, dim3(kCgBlocks, nslices), dim3(kCgThreads), 0, s, x, n, scale, partial);
    return hipGetLastError();
}
hipError_t launch_cg_wnorm2(const float2 *v, size_t n, int nslices, int nchan, int nro, float a, float b, double *partial, hipStream_t s)
{
    hipLaunchKernelGGL(cg_wnorm2_kernel, dim3(kCgBlocks, nslices), dim3(kCgThreads), 0, s, v, n, nchan, nro, a, b, partial);
    return hipGetLastError();
}
hipError_t launch_cg_finish(const double *partial, double *num, float *coef, int mode, int nslices, hipStream_t s)
{
    hipLaunchKernelGGL(cg_finish_kernel, dim3(nslices), dim3(64), 0, s, partial, num, coef, mode);
    return hipGetLastError();
}
hipError_t launch_cg_axpy(float2 *y, const float2 *x, const float *coef, float sign, size_t n, int nslices, hipStream_t s)
{
    const unsigned nb = (unsigned)std::min<size_t>(512, (n + kCgThreads - 1) / kCgThreads);
    hipLaunchKernelGGL(cg_axpy_kernel, dim3(nb, nslices), dim3(kCgThreads), 0, s, y, x, coef, sign, n);
    return hipGetLastError();
}
hipError_t launch_cg_xpby(float2 *pt, const float2 *zt, const float *coef, size_t n, int nslices, hipStream_t s)
{
    const unsigned nb = (unsigned)std::min<size_t>(512, (n + kCgThreads - 1) / kCgThreads);
    hipLaunchKernelGGL(cg_xpby_kernel, dim3(nb, nslices), dim3(kCgThreads), 0, s, pt, zt, coef, n);
    return hipGetLastError();
}
// mode 0: root-sum-of-squares, 1: Walsh (nc <= kWalshMaxCoils)
hipError_t launch_coil_combine(float2 *out, const float2 *coil, int nimg, int nc, int nt, int mode, int npatch, int nslices, hipStream_t s)
{
    const size_t work = (size_t)nimg * nimg * nt;
    if (mode == 1) {
        if (nc > kWalshMaxCoils) return hipErrorInvalidValue;
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
