// LDS atomic / RMW rate probe (tooling)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, int iters)
{
    __shared__ float tile[64 * 33];
    for (int i = threadIdx.x; i < 64 * 33; i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx = lane + wave * 131;
    unsigned* ut = (unsigned*)tile;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        idx += 7; if (idx >= 64 * 33) idx -= 64 * 33;
        if (MODE == 0) atomicAdd(&tile[idx], 1.0f);
        if (MODE == 1) atomicAdd(&ut[idx], 1u);
        if (MODE == 2) acc += __uint_as_float(atomicAdd(&ut[idx], 1u));
        if (MODE == 3) { float v = tile[idx]; tile[idx] = v + 1.0f; }
        if (MODE == 4) acc += tile[idx];
        if (MODE == 5) tile[idx] = acc + it;
        if (MODE == 6) atomicMax(&ut[idx], (unsigned)it);
        if (MODE == 7) atomicAdd(&((unsigned long long*)tile)[idx >> 1], (unsigned long long)(long long)(it - 1000));   // ds_add_u64
        if (MODE == 8) { atomicAdd(&ut[idx], 1u); atomicAdd(&ut[(idx + 1056) % (64 * 33)], 2u); }                       // two ds_add_u32
    }
    __syncthreads();
    float s = acc;
    for (int i = threadIdx.x; i < 64 * 33; i += blockDim.x) s += tile[i];
    if (s == 12345.f) out[0] = s;
}
template <int MODE> void run(const char* name, float* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2048, blocks = 1024;
    k<MODE><<<blocks, 256>>>(d, 16);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.3f ms  %.1f CU-cycles/wave-instr\n", name, ms, ms * 1e-3 * 2.4e9 * 256 / ((double)blocks * 4 * iters));
}
int main()
{
    float* d; hipMalloc(&d, 4);
    run<0>("ds_add_f32", d); run<1>("ds_add_u32", d); run<2>("ds_add_rtn_u32", d);
    run<3>("read+add+write", d); run<4>("ds_read_b32", d); run<5>("ds_write_b32", d); run<6>("ds_max_u32", d);
    run<7>("ds_add_u64", d); run<8>("2 x ds_add_u32", d);
    return 0;
}
