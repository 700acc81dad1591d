// Streaming copy rate by access width per lane (8-byte vs 16-byte loads / stores), grid-stride and line-shaped
// (one wave = one 4 KiB line, as the FFT passes read): what the fft512 kernels' 8-byte accesses can reach at best.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/copywidth.hip -o tools/probe/copywidth_main
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void copy_k(T *__restrict__ dst, const T *__restrict__ src, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// one wave per 512-element float2 line, 8 loads of 8 B per lane (stride 64 elements), like fft512_rows_kernel
__global__ void __launch_bounds__(256) line8_k(float2 *__restrict__ dst, const float2 *__restrict__ src, size_t nlines)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t line = (size_t)blockIdx.x * 4 + wave; line < nlines; line += (size_t)gridDim.x * 4) {
        float2 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[line * 512 + q * 64 + lane];
#pragma unroll
        for (int q = 0; q < 8; ++q) dst[line * 512 + q * 64 + lane] = v[q];
    }
}
// the same line with 4 loads of 16 B per lane
__global__ void __launch_bounds__(256) line16_k(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t nlines)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t line = (size_t)blockIdx.x * 4 + wave; line < nlines; line += (size_t)gridDim.x * 4) {
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = src[line * 256 + q * 64 + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[line * 256 + q * 64 + lane] = v[q];
    }
}
int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f GB/s (read+write)\n", name, 10 * 2.0 * bytes / (ms * 1e-3) / 1e9);
    };
    for (int blocks : {2048, 8192}) {
        printf("grid-stride, %d blocks\n", blocks);
        time("  float (4 B per lane)", [&] { copy_k<float><<<blocks, 256>>>((float *)b, (const float *)a, bytes / 4); });
        time("  float2 (8 B per lane)", [&] { copy_k<float2><<<blocks, 256>>>((float2 *)b, (const float2 *)a, bytes / 8); });
        time("  float4 (16 B per lane)", [&] { copy_k<float4><<<blocks, 256>>>((float4 *)b, (const float4 *)a, bytes / 16); });
    }
    const size_t nlines = bytes / 4096;
    for (int blocks : {4096, 16384, 65536}) {
        printf("4 KiB line per wave, %d blocks\n", blocks);
        time("  8 x 8 B per lane", [&] { line8_k<<<blocks, 256>>>((float2 *)b, (const float2 *)a, nlines); });
        time("  4 x 16 B per lane", [&] { line16_k<<<blocks, 256>>>((float4 *)b, (const float4 *)a, nlines); });
    }
    return 0;
}
