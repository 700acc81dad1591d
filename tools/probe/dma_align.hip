// Does global_load_lds_dwordx4 take source addresses that are only 8-byte aligned (24-byte records: complex-half, six coils)?
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/dma_align.hip -o tools/probe/dma_align_main
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void lds_dma16_s(const void *sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__global__ void k(const unsigned char *src, unsigned *out, int stride, int shift)
{
    __shared__ __align__(16) unsigned buf[64 * 4];
    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)buf;
    lds_dma16_s(src, (unsigned)(threadIdx.x * stride + shift), dst);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = buf[threadIdx.x * 4 + i];
}
int main()
{
    const int n = 64 * 64 + 64;
    std::vector<unsigned> h(n / 4 + 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x1000000u + (unsigned)i;
    unsigned char *d; unsigned *o; hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64 * 16);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int stride : {16, 24, 64}) for (int shift : {0, 4, 8, 12, 16}) {
        k<<<1, 64>>>(d, o, stride, shift);
        std::vector<unsigned> g(64 * 4);
        if (hipMemcpy(g.data(), o, 64 * 16, hipMemcpyDeviceToHost) != hipSuccess) { printf("stride %d shift %d: launch failed\n", stride, shift); return 1; }
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) if (g[l * 4 + i] != h[(l * stride + shift) / 4 + i]) ++bad;
        printf("stride %2d shift %2d: %s (%d words differ)\n", stride, shift, bad ? "WRONG" : "ok", bad);
    }
    return 0;
}
