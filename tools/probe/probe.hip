// Environment probe (tooling, not product): checks that a gfx950 code object built here
// runs on the GPU box, that rocFFT plans work, what LDS float atomics cost, and
// that the library can share a HIP runtime with PyTorch in one process.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return -1; } } while (0)

__global__ void lds_atomic_kernel(float* out, int iters, int stride)
{
    __shared__ float tile[64 * 33];
    for (int i = threadIdx.x; i < 64 * 33; i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    int lane = threadIdx.x & 63;
    int wave = threadIdx.x >> 6;
    for (int it = 0; it < iters; ++it) {
        int idx = ((lane * stride + it * 7 + wave * 131) % (64 * 33));
        atomicAdd(&tile[idx], 1.0f);
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < 64 * 33; i += blockDim.x) s += tile[i];
    atomicAdd(out, s);
}

__global__ void copy_kernel(float4* __restrict__ dst, const float4* __restrict__ src, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

extern "C" int probe_run(void* user_dev_ptr, int verbose)
{
    int dev = -1;
    CK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    if (verbose) printf("device %d: %s arch %s CUs %d LDS/block %zu\n", dev, prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.sharedMemPerBlock);

    // touch a caller-provided device pointer (e.g. from torch) if given
    if (user_dev_ptr) {
        hipPointerAttribute_t attr;
        hipError_t e = hipPointerGetAttributes(&attr, user_dev_ptr);
        printf("user ptr attributes: %s\n", e == hipSuccess ? "known to this runtime" : hipGetErrorString(e));
        CK(hipMemset(user_dev_ptr, 0, 16));
    }

    float* d_out;
    CK(hipMalloc(&d_out, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int stride : {1, 2, 33, 17}) {
        CK(hipMemset(d_out, 0, 4));
        const int iters = 4096, blocks = 1024;
        lds_atomic_kernel<<<blocks, 256>>>(d_out, 16, stride);
        CK(hipEventRecord(e0));
        lds_atomic_kernel<<<blocks, 256>>>(d_out, iters, stride);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double wave_instr = (double)blocks * 4 * iters;
        // cycles per wave-instruction per CU at 2.4 GHz, 256 CUs
        double cyc = ms * 1e-3 * 2.4e9 * 256 / wave_instr;
        printf("lds atomicAdd f32 stride %2d: %.3f ms, %.2f CU-cycles per wave-instr\n", stride, ms, cyc);
    }

    // streaming copy ceiling
    {
        size_t bytes = (size_t)1 << 30;
        float4 *a, *b;
        CK(hipMalloc(&a, bytes));
        CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes));
        copy_kernel<<<2048, 256>>>(b, a, bytes / 16);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) copy_kernel<<<2048, 256>>>(b, a, bytes / 16);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("float4 copy: %.1f GB/s (read+write)\n", 5 * 2.0 * bytes / (ms * 1e-3) / 1e9);
        CK(hipFree(a));
        CK(hipFree(b));
    }

    // rocFFT 512x512 batched C2C inverse, in place
    {
        rocfft_setup();
        for (int batch : {1, 64, 512}) {
            size_t n = 512;
            size_t lengths[2] = {n, n};
            rocfft_plan plan = nullptr;
            rocfft_status st = rocfft_plan_create(&plan, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                                                  rocfft_precision_single, 2, lengths, batch, nullptr);
            if (st != rocfft_status_success) { fprintf(stderr, "rocfft_plan_create failed %d\n", st); return -2; }
            size_t wbytes = 0;
            rocfft_plan_get_work_buffer_size(plan, &wbytes);
            void* wbuf = nullptr;
            if (wbytes) CK(hipMalloc(&wbuf, wbytes));
            rocfft_execution_info info;
            rocfft_execution_info_create(&info);
            if (wbytes) rocfft_execution_info_set_work_buffer(info, wbuf, wbytes);
            float2* d;
            CK(hipMalloc(&d, n * n * batch * sizeof(float2)));
            std::vector<float2> h(n * n * batch);
            for (size_t i = 0; i < h.size(); ++i) h[i] = make_float2(0.f, 0.f);
            h[1] = make_float2(1.f, 0.f);  // delta at column 1 -> exp(+2 pi i m2 / n)
            CK(hipMemcpy(d, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice));
            void* bufs[1] = {d};
            rocfft_execute(plan, bufs, nullptr, info);
            CK(hipDeviceSynchronize());
            std::vector<float2> r(n * n);
            CK(hipMemcpy(r.data(), d, n * n * sizeof(float2), hipMemcpyDeviceToHost));
            double err = 0;
            for (size_t m1 = 0; m1 < n; m1 += 37)
                for (size_t m2 = 0; m2 < n; m2 += 11) {
                    double ph = 2 * M_PI * m2 / n;
                    err = fmax(err, hypot(r[m1 * n + m2].x - cos(ph), r[m1 * n + m2].y - sin(ph)));
                }
            CK(hipEventRecord(e0));
            const int reps = 10;
            for (int i = 0; i < reps; ++i) rocfft_execute(plan, bufs, nullptr, info);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("rocfft 512x512 inverse batch %d: work %zu B, max err %.2e, %.3f us per image\n", batch, wbytes, err, ms * 1e3 / reps / batch);
            rocfft_execution_info_destroy(info);
            rocfft_plan_destroy(plan);
            if (wbuf) CK(hipFree(wbuf));
            CK(hipFree(d));
        }
    }
    CK(hipFree(d_out));
    return 0;
}

#ifdef PROBE_MAIN
int main() { return probe_run(nullptr, 1); }
#endif
