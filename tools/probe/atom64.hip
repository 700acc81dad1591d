// Is ds_add_u64 exact when several lanes of ONE wave instruction hit the same 8-byte address and the low words carry?
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/atom64.hip -o tools/probe/atom64_main
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const unsigned long long *val, const int *slot, unsigned long long *out, int per_lane)
{
    __shared__ unsigned long long t[256];
    t[threadIdx.x] = 0ull;
    __syncthreads();
    for (int i = 0; i < per_lane; ++i) {
        const int idx = (blockIdx.x * per_lane + i) * 256 + threadIdx.x;
        __hip_atomic_fetch_add(&t[slot[idx]], val[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = t[threadIdx.x];
}
int main()
{
    const int blocks = 64, per_lane = 50, n = blocks * per_lane * 256;
    std::vector<unsigned long long> v(n), want(blocks * 256, 0ull), got(blocks * 256);
    std::vector<int> s(n);
    srand(7);
    for (int mode = 0; mode < 3; ++mode) {          // 0: distinct slots per instruction, 1: pairs share a slot, 2: 8 lanes share
        std::fill(want.begin(), want.end(), 0ull);
        for (int b = 0; b < blocks; ++b)
            for (int i = 0; i < per_lane; ++i)
                for (int l = 0; l < 256; ++l) {
                    const int idx = (b * per_lane + i) * 256 + l;
                    const long long re = (rand() % 2000001) - 1000000, im = (rand() % 2000001) - 1000000;
                    v[idx] = (unsigned long long)((re << 32) + im);
                    s[idx] = mode == 0 ? (l * 7 + i) % 256 : (mode == 1 ? ((l / 2) * 5 + i) % 256 : ((l / 8) * 3 + i) % 256);
                    want[b * 256 + s[idx]] += v[idx];
                }
        unsigned long long *dv, *dout; int *ds;
        hipMalloc(&dv, n * 8); hipMalloc(&ds, n * 4); hipMalloc(&dout, blocks * 256 * 8);
        hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(ds, s.data(), n * 4, hipMemcpyHostToDevice);
        k<<<blocks, 256>>>(dv, ds, dout, per_lane);
        hipMemcpy(got.data(), dout, blocks * 256 * 8, hipMemcpyDeviceToHost);
        int bad = 0; long long worst = 0;
        for (int i = 0; i < blocks * 256; ++i) if (got[i] != want[i]) { ++bad; long long d = (long long)(got[i] - want[i]); if (llabs(d) > llabs(worst)) worst = d; }
        printf("mode %d: %d of %d sums differ (largest difference %lld = %.3f x 2^32)\n", mode, bad, blocks * 256, worst, (double)worst / 4294967296.0);
        hipFree(dv); hipFree(ds); hipFree(dout);
    }
    return 0;
}
