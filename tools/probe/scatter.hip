// What a sample-driven gridding kernel could get out of the LDS atomics (round 5; the numbers behind DESIGN 4.1c):
// every lane is one k-space sample on a spoke through a 32x32 tile and updates the 4x4 grid points of its Kaiser-Bessel
// footprint in an LDS tile -- the access pattern of a scatter, not the strided walk of tools/probe/atom.hip -- with
//   MODE 0  ds_add_u32   x2 per point (re, im as 32-bit fixed point)
//   MODE 1  ds_add_u64   x1 per point (re << 32 + im in one 64-bit integer)
//   MODE 2  ds_add_f32   x2 per point
//   MODE 3  ds_add_u32   x2 per point, points of zero weight skipped by EXEC (a quarter of them)
//   MODE 4  no LDS update at all: the VALU work of the loop alone (positions, products, conversions)
// VALU work per point is what the real kernel would do (weight product, two multiplies, two conversions).
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/scatter.hip -o tools/probe/scatter_main
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int T = 32, PITCH = 40;           // tile + halo rows of 40 points

template <int MODE>
__global__ void __launch_bounds__(256) k(float *sink, int iters, float spacing)
{
    __shared__ unsigned long long tile64[PITCH * PITCH];        // 12.8 KB: (re, im) per point
    unsigned *tile = reinterpret_cast<unsigned *>(tile64);
    float *tilef = reinterpret_cast<float *>(tile64);
    for (int i = threadIdx.x; i < PITCH * PITCH; i += 256) tile64[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        // a spoke through the tile: direction from (it, wave), lane = radius
        const float ang = 0.37f * (float)(it * 4 + wave) + 0.11f * (float)blockIdx.x;
        const float c = __cosf(ang), s = __sinf(ang);
        const float u = (float)(lane - 32) * spacing;
        const float kx = 18.f + u * c, ky = 18.f + u * s;
        const float fx = floorf(kx - 2.f), fy = floorf(ky - 2.f);
        const int ix = (int)fx + 1, iy = (int)fy + 1;
        const float tx = kx - fx, ty = ky - fy;
        float wx[4], wy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wx[j] = fmaf(tx, fmaf(tx, 0.1f * (j + 1), -0.3f), 0.7f);
            wy[j] = fmaf(ty, fmaf(ty, 0.2f * (j + 1), -0.2f), 0.6f);
        }
        const float dre = 1000.f * c, dim = 1000.f * s;
        const bool inside = ix >= 0 && iy >= 0 && ix + 3 < PITCH && iy + 3 < PITCH;
        if (inside) {
            const int base = iy * PITCH + ix;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float w = wx[j] * wy[i];
                    const float vr = w * dre, vi = w * dim;
                    const int p = base + i * PITCH + j;
                    if (MODE == 0) {
                        atomicAdd(&tile[2 * p], (unsigned)__float2int_rn(vr));
                        atomicAdd(&tile[2 * p + 1], (unsigned)__float2int_rn(vi));
                    } else if (MODE == 1) {
                        const long long v = ((long long)__float2int_rn(vr) << 32) + (long long)__float2int_rn(vi);
                        atomicAdd(&tile64[p], (unsigned long long)v);
                    } else if (MODE == 2) {
                        atomicAdd(&tilef[2 * p], vr);
                        atomicAdd(&tilef[2 * p + 1], vi);
                    } else if (MODE == 3) {
                        if (((i * 4 + j + lane + it) & 3) != 0) {
                            atomicAdd(&tile[2 * p], (unsigned)__float2int_rn(vr));
                            atomicAdd(&tile[2 * p + 1], (unsigned)__float2int_rn(vi));
                        }
                    } else if (MODE == 5) {                                      // the atomics alone: no per-point arithmetic
                        atomicAdd(&tile64[p], (unsigned long long)(unsigned)(i * 4 + j + 1));
                    } else if (MODE == 6) {                                      // (racy) plain read-add-write of 8 bytes: the rate only
                        float2 *const q = reinterpret_cast<float2 *>(tile64) + p;
                        float2 t = *q;
                        t.x += vr; t.y += vi;
                        *q = t;
                        asm volatile("" ::: "memory");
                    } else {
                        acc += (float)__float2int_rn(vr) + (float)__float2int_rn(vi);
                    }
                }
        }
    }
    __syncthreads();
    float t = acc;
    for (int i = threadIdx.x; i < 2 * PITCH * PITCH; i += 256) t += (float)tile[i];
    if (t == 12345.678f) sink[0] = t;
}

int main()
{
    float *sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    auto run = [&](const char *name, auto kern, int wgs_per_cu, float spacing = 1.0f) {
        const int grid = 256 * wgs_per_cu;
        kern<<<grid, 256>>>(sink, 64, spacing); hipDeviceSynchronize();
        for (int r = 0; r < 3; ++r) kern<<<grid, 256>>>(sink, iters, spacing);
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) kern<<<grid, 256>>>(sink, iters, spacing);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double wave_iters = 5.0 * grid * 4 * (double)iters;           // wave iterations = 64 samples each
        const double sec = ms * 1e-3;
        printf("%-44s spacing %.2f %d wg/CU: %7.2f ms  %6.1f ns per 64 samples and CU (%5.0f clk at 2.1 GHz)  %7.1f G samples/s\n", name, spacing, wgs_per_cu, ms / 5,
               sec / (wave_iters / 256) * 1e9, sec / (wave_iters / 256) * 2.1e9, wave_iters * 64 / sec / 1e9);
    };
    for (float sp : {1.0f, 0.55f}) {
        run("ds_add_u32 x2 per point", k<0>, 4, sp);
        run("ds_add_u64 x1 per point", k<1>, 4, sp);
        run("ds_add_u32 x2, zero weights skipped (1/4)", k<3>, 4, sp);
        run("ds_add_u64 x1, no per-point arithmetic", k<5>, 4, sp);
        run("plain 8-byte read-add-write (racy)", k<6>, 4, sp);
        run("VALU only", k<4>, 4, sp);
    }
    run("ds_add_u64 x1 per point", k<1>, 2, 1.0f);
    run("ds_add_u64 x1 per point", k<1>, 5, 1.0f);
    run("ds_add_f32 x2 per point", k<2>, 4, 1.0f);
    return 0;
}
