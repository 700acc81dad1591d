// What the LDS-DMA path (global_load_lds_dwordx4) moves per clock and CU, by access shape and by how many waves issue it:
//   contiguous   lane l reads 16 bytes at base + 16 l      (1 KiB per instruction, 8 x 128-byte lines)
//   strided64    lane l reads 16 bytes at base + 64 l      (one 16-byte piece of 64 different 64-byte records: the arc
//                gridding kernel's planar copy, one coil pair of a spoke segment)
//   quad         lane l reads piece l & 3 of record l >> 2 (16 records x 64 bytes: a quad of lanes = one line)
// Source window per workgroup: 256 KiB (L2-resident), every instruction on lines no earlier one touched.
// Result (MI355X, profiles/round4_lds_dma_rates.txt): contiguous and quad 26 / 36 / 47 B per clock and CU with 1 / 2 / 4 workgroups
// issuing, the 64-byte stride 2.4-4.1: the path takes one new cache line every ~4 clocks per CU whatever the lanes ask of it, so a
// copy's time is its count of first-touched lines x the memory latency / the outstanding-miss depth, not its instruction count.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/ldsdma.hip -o tools/probe/ldsdma_main
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void lds_dma16_s(const void *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int SHAPE, bool DMA>
__global__ void __launch_bounds__(256) k(const unsigned char *src, float *sink, int iters, int per_wait)
{
    extern __shared__ __align__(16) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *base = src + (size_t)(blockIdx.x % 1024) * (256 << 10);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + wave * 8192;
    const unsigned lo = SHAPE == 0 ? 16u * lane : (SHAPE == 1 ? 64u * lane : 64u * (lane >> 2) + 16u * (lane & 3));
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        const unsigned off = (unsigned)(((it * 4 + wave) * 4096 + (it & 3) * 16) & ((256 << 10) - 4096 - 1)) & ~15u;
        if (DMA) {
            lds_dma16_s(base, off + lo, __builtin_amdgcn_readfirstlane((int)(lds0 + (it & 7) * 1024)));
            if ((it + 1) % per_wait == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            const float4 v = *reinterpret_cast<const float4 *>(base + off + lo);
            *reinterpret_cast<float4 *>(lds + wave * 8192 + (it & 7) * 1024 + 16 * lane) = v;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float4 t = *reinterpret_cast<const float4 *>(lds + wave * 8192 + 16 * lane);
    acc.x += t.x + t.y + t.z + t.w;
    if (acc.x == 12345.678f) sink[0] = acc.x;
}

int main()
{
    void *src; float *sink;
    hipMalloc(&src, (size_t)256 << 20); hipMalloc(&sink, 64);
    hipMemset(src, 1, (size_t)256 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    auto run = [&](const char *name, auto kern, int wgs_per_cu, int per_wait) {
        const int grid = 256 * wgs_per_cu;
        kern<<<grid, 256, 32768>>>((const unsigned char *)src, sink, iters, per_wait); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) kern<<<grid, 256, 32768>>>((const unsigned char *)src, sink, iters, per_wait);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double bytes = 5.0 * grid * 4 * (double)iters * 1024;
        printf("%-46s %d wg/CU, wait every %2d: %8.1f GB/s = %6.1f GB/s per CU (%5.1f B/clk at 2.0 GHz)\n", name, wgs_per_cu, per_wait,
               bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e9 / 256, bytes / (ms * 1e-3) / 256 / 2.0e9);
    };
    for (int w : {1, 2, 4}) {
        run("LDS-DMA contiguous (1 KiB per instruction)", k<0, true>, w, 8);
        run("LDS-DMA 64-byte stride (planar copy of records)", k<1, true>, w, 8);
        run("LDS-DMA quad = one 64-byte record", k<2, true>, w, 8);
    }
    run("LDS-DMA contiguous, wait every instruction", k<0, true>, 1, 1);
    run("LDS-DMA 64-byte stride, wait every 16", k<1, true>, 1, 16);
    return 0;
}
