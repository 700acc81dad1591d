// Which CUs a stream made with hipExtStreamCreateWithCUMask really runs on (MI355X: 8 XCDs x 32 CUs), by mask pattern.
// Every workgroup records (XCC_ID, HW_ID) of its first wave; the host counts the distinct (xcc, se, sh, cu) it saw per XCD.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/cumask.hip -o tools/probe/cumask_main
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void where(unsigned *out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;      // long enough that the launch fills every CU it may use
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 15u) << 16 | (hw & 0xffffu) | (x == 12345.f ? 1u << 31 : 0u);
}

int main()
{
    const int wgs = 4096;
    unsigned *d; hipMalloc(&d, wgs * 4);
    std::vector<unsigned> h(wgs);
    auto run = [&](const char *name, const uint32_t *mask) {
        hipStream_t s;
        if (mask) { if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: create failed\n", name); return; } }
        else hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        where<<<wgs, 256, 0, s>>>(d, 20000);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, wgs * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> cus[16];
        for (unsigned v : h) cus[(v >> 16) & 15].insert(v & 0xff00u);     // cu_id [11:8], sh_id [12], se_id [15:13]
        printf("%-28s CUs used per XCD:", name);
        int total = 0;
        for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); total += (int)cus[x].size(); }
        printf("  total %d\n", total);
        hipStreamDestroy(s);
    };
    run("no mask", nullptr);
    for (int split : {2, 3, 4, 6, 8}) {
        uint32_t a[8] = {0}, b[8] = {0};
        for (int cu = 0; cu < 256; ++cu) (cu % split == 0 ? a : b)[cu / 32] |= 1u << (cu % 32);
        char nm[64];
        snprintf(nm, sizeof nm, "every %d-th bit", split); run(nm, a);
        snprintf(nm, sizeof nm, "all but every %d-th bit", split); run(nm, b);
    }
    {
        uint32_t a[8] = {0}, b[8] = {0};
        for (int cu = 0; cu < 256; ++cu) (cu < 64 ? a : b)[cu / 32] |= 1u << (cu % 32);
        run("bits 0-63", a); run("bits 64-255", b);
    }
    {
        uint32_t a[8] = {0}, b[8] = {0};
        for (int cu = 0; cu < 256; ++cu) ((cu / 8) % 4 == 0 ? a : b)[cu / 32] |= 1u << (cu % 32);
        run("bits with (bit/8)%4==0", a); run("the rest", b);
    }
    return 0;
}
