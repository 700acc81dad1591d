// The angle-dependent tables of a plan, built ON THE DEVICE from the (cos, sin) table: every window's spokes sorted by line
// angle (the order the arc, scatter and centre kernels sum in) and the centre kernel's block windows.  arc_prep_kernel
// (tron_grid_arc.hip) then clips the sorted lists against the tiles.  Until round 5 the host did both (std::stable_sort per window:
// 26 ms for the 256 windows of the metric shape, eleven times the step they serve); now a plan -- and tron_plan_retarget, which
// rebuilds exactly these tables for a new skip_angles while the previous batch is still being gridded -- queues three launches.
//
// The angles themselves (src/tron.cu:509-511: PHI * float(pe + skip), wrapped, sincosf) stay with the host's libm: they are what
// the reference and the oracle compute, and the bit-exact kernels read them.  Everything here only ORDERS spokes and bounds
// searches; a line angle that differs from the host's atan2f in its last bit changes neither (ties keep acquisition order, and
// the windows carry 2e-3 rad of slack).
#include "tron_device.h"

namespace tron {

constexpr int kTrajThreads = 256;
constexpr int kTrajMaxNpe = kArcMaxWindow;                  // 4 096 spokes per window

// grid = windows, block = 256.  Rank of spoke k = how many spokes of the window come before it in (line angle, acquisition index)
// order: the stable sort of build_arc_tables (rounds 3-5, host) as a counting rank -- npe^2 comparisons of LDS broadcasts, 160 k for
// the metric's 402 spokes.  Windows of more than kArcMaxNpe spokes also get the lists of every pass (spokes [q sub, (q + 1) sub) in
// the same order), which is what arc_prep_kernel is run on.
__global__ void __launch_bounds__(kTrajThreads)
traj_sort_kernel(const TrajSortParams p)
{
    __shared__ float s_phi[kTrajMaxNpe];
    const int w = blockIdx.x, tid = threadIdx.x;
    const float2 *t = p.trig + (size_t)w * p.win_stride;
    const float pi = 3.14159265358979f;                       // (float)M_PI
    for (int k = tid; k < p.npe; k += kTrajThreads) {
        const float2 cs = t[k];
        float a = atan2f(cs.y, cs.x);                         // line angle in [0, pi): a spoke and its mirror image are one line
        if (a < 0.f) a += pi;
        if (a >= pi) a -= pi;
        if (a < 0.f) a = 0.f;
        s_phi[k] = a;
    }
    __syncthreads();
    for (int k = tid; k < p.npe; k += kTrajThreads) {
        const float a = s_phi[k];
        const int q = k / p.sub, lo = q * p.sub, hi = min(p.npe, lo + p.sub);
        int rank = 0, rank_q = 0;
        for (int j = 0; j < p.npe; ++j) {
            const float b = s_phi[j];
            const int before = (b < a || (b == a && j < k)) ? 1 : 0;
            rank += before;
            rank_q += (j >= lo && j < hi) ? before : 0;
        }
        const size_t o = (size_t)w * p.npe + rank;
        p.order[o] = (unsigned short)k;
        p.phi[o] = a;
        p.cs[o] = t[k];
        if (p.npass > 1) {
            const size_t oq = (size_t)q * p.nwin * p.sub + (size_t)w * (hi - lo) + rank_q;      // pass q: rows of its own length
            p.order_q[oq] = (unsigned short)k;
            p.phi_q[oq] = a;
            p.cs_q[oq] = t[k];
        }
    }
}

// thread = (block, window): the block's run of the window's sorted list, first entry | entries << 16 (circular) -- the two binary
// searches of build_centre_windows (rounds 4-5, host) over the device's line angles; the block's angular window [lo, hi] is the
// plan's (host, geometry only).
__global__ void __launch_bounds__(kTrajThreads)
traj_centre_windows_kernel(const TrajCentreParams p)
{
    const int g = blockIdx.x * kTrajThreads + threadIdx.x;
    const int w = blockIdx.y;
    if (g >= p.ngroups) return;
    const float4 gw = p.gwin[g];
    const int flags = (int)gw.z;
    const float *ph = p.phi + (size_t)w * p.npe;
    int first = 0, count = p.npe;
    if (!(flags & 1)) {
        int lo = 0, cnt = p.npe;
        while (cnt > 0) {                                     // lower bound: first entry not below the window's start
            const int step = cnt >> 1;
            if (ph[lo + step] < gw.x) { lo += step + 1; cnt -= step + 1; } else cnt = step;
        }
        const int na = lo;
        lo = 0; cnt = p.npe;
        while (cnt > 0) {                                     // upper bound: first entry beyond the window's end
            const int step = cnt >> 1;
            if (!(gw.y < ph[lo + step])) { lo += step + 1; cnt -= step + 1; } else cnt = step;
        }
        const int nb = lo;
        first = na;
        count = (flags & 2) ? p.npe - na + nb : nb - na;
        count = max(0, min(count, p.npe));
    }
    p.out[(size_t)w * p.ngroups + g] = (uint32_t)(first % max(p.npe, 1)) | ((uint32_t)count << 16);
}

hipError_t launch_traj_sort(const TrajSortParams &p, hipStream_t s)
{
    if (p.npe < 1 || p.npe > kTrajMaxNpe || p.sub < 1 || p.npass < 1 || p.nwin < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(traj_sort_kernel, dim3((unsigned)p.nwin), dim3(kTrajThreads), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_traj_centre_windows(const TrajCentreParams &p, hipStream_t s)
{
    if (p.ngroups < 1 || p.nwin < 1) return hipSuccess;
    hipLaunchKernelGGL(traj_centre_windows_kernel, dim3((unsigned)((p.ngroups + kTrajThreads - 1) / kTrajThreads), (unsigned)p.nwin), dim3(kTrajThreads), 0, s, p);
    return hipGetLastError();
}

__global__ void warm_traj_tu() {}

hipError_t warm_traj()   // see warm_kernels() in tron_kernels.hip
{
    hipLaunchKernelGGL(warm_traj_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
