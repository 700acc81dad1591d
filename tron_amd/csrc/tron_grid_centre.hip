// The k-space centre of the TRON_KB_FAST gridding path: samples |r| < inner_r0 (14), which the arc kernel (tron_grid_arc.hip)
// leaves out of the four tiles that meet at the origin.
//
// Same (sample, point) pairs and the same arithmetic per pair as the reference's gridradial2d (src/tron.cu:465-536): kx, ky
// :514-515 with the SIGNED radius r of both loops (:512 and :521), weights :516, band :498-502, density compensation :412-414,
// scale :532, r = 0 counted twice where the band starts at 0 (:512 and :521 both visit it).
//
// Near the origin every spoke passes every point: a 2x2 block there meets ~400 spokes where a block of the arc kernel meets 6, so
// "thread = block" (arc kernel) starves and "sort the samples by cell" (binned kernel, which did this job until round 3 at 0.30
// lanes active and 0.28 ms per 1 024 coil-slices for 5 % of the samples) spends its time sorting.  Here one WAVE owns one 2x2
// block of one slice and its lanes share out the block's VISITS (sample, block):
//   window  the block's run of the angle-sorted spoke list (the arc kernel's rule; worked out by the host at plan creation);
//   chunk   64 spokes of the run, one per lane: clip against the block's footprint in SIGNED r (so no run ever "wraps"; |r| <
//           inner_r0); an exclusive scan of the chord lengths numbers the chunk's visits, and every spoke writes its record and
//           its lane number under each of its visits into the wave's LDS;
//   weights 64 visits at a time, one per lane: owner -> spoke record -> (kx, ky) -> the arc kernel's weights (pair table in LDS,
//           band masks, density compensation, r = 0 doubled): the four point weights and the sample's address go to LDS;
//   sums    the same 64 visits, LPV lanes per visit (one coil pair each, 16 bytes: the LPV lanes of a visit read ONE 64-byte
//           line, 16 lines per wave instruction instead of 64 -- this work is bound by the texture addresser, not by arithmetic:
//           a version with one lane per visit and all coils per lane spent 450 addresser cycles per 64 visits and ran at the
//           binned kernel's 0.2-0.3 ms), all loads of the group in flight together, 8 packed FMAs per visit and lane;
//   end     the lanes of a coil pair are summed by shuffles and the block is ADDED to what the arc kernel stored: this kernel
//           runs behind it on the same stream.  No atomics, no partial tiles in HBM, the same sums in the same order every run.
#include "tron_device.h"
#include "tron_grid_store.h"

namespace tron {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kCenWaves = 4;                                   // waves per workgroup, each on its own (block, slice)

struct CenWaveLds {
    uint4 rec[64];                  // per spoke of the chunk: cos, sin, first visit of the chunk, (first radius + 32) << 16 | spoke
    float4 wq[64];                  // per visit of the group: the four point weights (band, density compensation and scale of r = 0 applied)
    unsigned off[64];               //                         byte offset of the sample's first coil
    unsigned char owner[768];       // per visit of the chunk: the lane whose spoke it lies on (a chord holds at most (1 + 2 W) sqrt(2) + 1 < 12 radii, W <= 3)
};

struct CenLds {
    float2 lut[3 * kArcLutEntries]; // Kaiser-Bessel pair table (build_kb_pair_lut)
    CenWaveLds w[kCenWaves];
};

// LPV lanes per visit, each with one coil pair (16 bytes of the sample; one coil: LPV = 1, 8 bytes): chunks of 2 LPV coils
template <int LPV, bool ONE, bool HALF>
__global__ void __launch_bounds__(64 * kCenWaves)
grid_centre_kernel(const GridParams p)
{
    static_assert(LPV == 1 || LPV == 2 || LPV == 4, "lanes per visit");
    static_assert(!ONE || LPV == 1, "one coil: one lane per visit");
    constexpr int NC = ONE ? 1 : 2;                             // coils per lane
    constexpr int VPS = 64 / LPV;                               // visits per sub-step
    __shared__ CenLds lds;
    const int lane = threadIdx.x & 63;
    CenWaveLds &L = lds.w[threadIdx.x >> 6];
    for (int i = threadIdx.x; i < 3 * kArcLutEntries; i += 64 * kCenWaves) lds.lut[i] = p.kb_lut[i];
    __syncthreads();                                            // (the only barrier: every wave takes part before it may leave)
    // work item -> (block, slice): slice z lives on XCD z % 8 (workgroups go round the 8 XCDs), and its blocks follow each other in
    // time there, the block nearest the origin (most visits) first: the slice's samples are fetched into that XCD's L2 once
    const int wg = blockIdx.x;
    const int xcd = wg & 7;
    const long long seq = (long long)(wg >> 3) * kCenWaves + (threadIdx.x >> 6);       // position in that XCD's sequence
    const int z = (int)(seq / p.cen_ngroups) * 8 + xcd;
    if (z >= p.nslices) return;                                 // (padding of the last workgroups)
    const int gi = (int)(seq % p.cen_ngroups);
    const int grp = p.cen_groups[gi];                               // block: (col | row << 8) of the origin-centred 32 x 32 square, nearest first
    const int cp = lane % LPV;                                  // this lane's coil pair of the chunk
    const int c0 = p.coil0 + blockIdx.y * (2 * LPV) + 2 * cp;   // ... its first coil
    const int ncl = ONE ? 1 : min(2, p.nchan - c0);             // coils this lane really has (<= 0: none)
    const int n = p.nxos, h = n / 2;
    const int X0 = 2 * (grp & 255) - 16, Y0 = 2 * (grp >> 8) - 16;
    const int rcap0 = p.inner_r0 - 1;

    unsigned bmask[4];                                          // band per point as a mask over |r| (|r| <= 13 here), src/tron.cu:498-502
    int bandhi = -1, bandlo = 1 << 20;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int X = X0 + (q & 1), Y = Y0 + (q >> 1);
        const uint32_t bnd = p.band[(size_t)(Y + h) * n + (X + h)];
        const int lo = (int)(bnd & 0xffffu), hi = min((int)(bnd >> 16), 31);
        bmask[q] = 0u;
        if (lo <= hi) {
            bmask[q] = (0xffffffffu >> (31 - (hi - lo))) << lo;
            bandhi = max(bandhi, hi);
            bandlo = min(bandlo, lo);
        }
    }
    const int rcap = min(rcap0, bandhi);
    if (bandlo > rcap) return;                                  // no sample |r| < inner_r0 reaches this block (wave-uniform)

    const int npe = p.npe;
    const size_t win = (size_t)z * p.arc_slice_stride;
    const unsigned short *order = p.cen_order + win * npe;
    const float2 *scs = p.cen_cs + win * npe;
    const unsigned wnd = p.cen_win[win * p.cen_ngroups + gi];    // first spoke | spokes << 16 (build_centre_windows, at plan creation)
    const int jstart = (int)(wnd & 0xffffu), cnt = (int)(wnd >> 16);

    const float X0f = (float)X0, Y0f = (float)Y0;
    const v2f p0v = {X0f, Y0f}, lscale2 = {p.lut_scale, p.lut_scale};
    const float We = p.W + 1e-3f;
    const float xlo = X0f - We, xhi = X0f + 1.0f + We, ylo = Y0f - We, yhi = Y0f + 1.0f + We;
    const float rcap_f = (float)rcap;
    const float dcf_a = p.apply_dcf ? p.dcf_a : 0.0f, dcf_b = p.apply_dcf ? p.dcf_b : 1.0f;
    const float2 *lut = lds.lut + p.lut_bias;                   // entry of table position 0
    const unsigned nchan8 = (unsigned)p.nchan * (HALF ? 4u : 8u);
    const unsigned char *in = reinterpret_cast<const unsigned char *>(p.nudata) + ((size_t)z * (size_t)p.in_slice_stride + c0) * (HALF ? 4 : 8);

    v2f acc[4][NC];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[q][c] = (v2f){0.f, 0.f};

    for (int t0 = 0; t0 < cnt; t0 += 64) {
        // ---- this lane's spoke of the chunk: clip, number its visits ----
        const int t = t0 + lane;
        int len = 0, ra = 0;
        unsigned pe = 0u;
        float2 cs = make_float2(1.f, 0.f);
        if (t < cnt) {
            int j = jstart + t;
            if (j >= npe) j -= npe;
            cs = scs[j];
            pe = order[j];
            const float ic = __builtin_amdgcn_rcpf(cs.x), is = __builtin_amdgcn_rcpf(cs.y);   // (1 / 0 = inf clips like a huge number; the box edges are never 0)
            const float xa = xlo * ic, xb = xhi * ic;
            const float ya = ylo * is, yb = yhi * is;
            const float lo = fmaxf(fmaxf(fminf(xa, xb), fminf(ya, yb)), -rcap_f);
            const float hi = fminf(fminf(fmaxf(xa, xb), fmaxf(ya, yb)), rcap_f);
            ra = (int)ceilf(lo);
            const int rb = (int)floorf(hi);
            len = rb >= ra ? min(rb - ra + 1, 12) : 0;                                       // (never cuts: see CenWaveLds::owner)
        }
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const int excl = incl - len;
        const int total = __builtin_amdgcn_readlane(incl, 63);
        L.rec[lane] = make_uint4(__float_as_uint(cs.x), __float_as_uint(cs.y), (unsigned)excl, ((unsigned)(ra + 32) << 16) | pe);
        for (int i = 0; i < len; ++i) L.owner[excl + i] = (unsigned char)lane;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: its own LDS writes are visible once they have completed)

        for (int v0 = 0; v0 < total; v0 += 64) {
            // ---- weights of visits v0 .. v0 + 63, one per lane ----
            const int v = v0 + lane;
            float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned off = 0u;
            if (v < total) {
                const uint4 rec = L.rec[L.owner[v]];
                const int ri = (int)((rec.w >> 16) & 63u) - 32 + (v - (int)rec.z);
                // sample of radius r on spoke pe: nudata[nchan * (nro * pe + r + nro / 2) + c]   src/tron.cu:517,519 (nro == nxos)
                off = (unsigned)(p.nro * (int)(rec.w & 0xffffu) + p.nro / 2 + ri) * nchan8;
                // (kx, ky) = r (cos, sin) and the distances to the block's first column / row, op for op src/tron.cu:514-516
                const float rf = (float)ri;
                const v2f kxy = (v2f){rf, rf} * (v2f){__uint_as_float(rec.x), __uint_as_float(rec.y)};
                const v2f tp = (kxy - p0v) * lscale2;
                const v2f tt = {__builtin_truncf(tp.x), __builtin_truncf(tp.y)};
                const v2f fv = tp - tt;
                const float2 *lx = lut + (int)tt.x, *ly = lut + (int)tt.y;
                const float2 x0c = lx[0], x1c = lx[kArcLutEntries], x2c = lx[2 * kArcLutEntries];
                const float2 y0c = ly[0], y1c = ly[kArcLutEntries], y2c = ly[2 * kArcLutEntries];
                const int ar = ri < 0 ? -ri : ri;
                // src/tron.cu:412 (|ro - nro/2| = |r|); r = 0 is visited by both loops of the reference where the band starts at 0
                const float sdc = fmaf(dcf_a, fabsf(rf), dcf_b) * (ri == 0 ? 2.0f : 1.0f);
                const v2f fxv = {fv.x, fv.x}, fyv = {fv.y, fv.y}, sdcv = {sdc, sdc};
                const v2f wx = __builtin_elementwise_fma(fxv, __builtin_elementwise_fma(fxv, (v2f){x2c.x, x2c.y}, (v2f){x1c.x, x1c.y}), (v2f){x0c.x, x0c.y});
                const v2f wy = __builtin_elementwise_fma(fyv, __builtin_elementwise_fma(fyv, (v2f){y2c.x, y2c.y}, (v2f){y1c.x, y1c.y}), (v2f){y0c.x, y0c.y}) * sdcv;
                const v2f w01 = wx * (v2f){wy.x, wy.x}, w23 = wx * (v2f){wy.y, wy.y};    // src/tron.cu:516
                float wq[4] = {w01.x, w01.y, w23.x, w23.y};
#pragma unroll
                for (int q = 0; q < 4; ++q)                                                // src/tron.cu:512,521: Rlo <= |r| <= Rhi
                    wq[q] = __uint_as_float(__float_as_uint(wq[q]) & (unsigned)__builtin_amdgcn_sbfe((int)bmask[q], ar, 1));
                w4 = make_float4(wq[0], wq[1], wq[2], wq[3]);
            }
            L.wq[lane] = w4;
            L.off[lane] = off;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

            // ---- sums: LPV lanes per visit, every load of the group first ----
            v4f dd[LPV];
            float4 ww[LPV];
#pragma unroll
            for (int s = 0; s < LPV; ++s) {
                const int vi = s * VPS + lane / LPV;
                ww[s] = L.wq[vi];
                const unsigned o = L.off[vi];
                dd[s] = (v4f){0.f, 0.f, 0.f, 0.f};
                if (ncl > 0 && v0 + vi < total) {
                    if constexpr (ONE) {
                        const float2 s0 = load_sample<HALF>(in + o, 0);
                        dd[s] = (v4f){s0.x, s0.y, 0.f, 0.f};
                    } else if constexpr (HALF) {
                        const float2 s0 = load_sample<true>(in + o, 0), s1 = load_sample<true>(in + o, 1);
                        dd[s] = (v4f){s0.x, s0.y, s1.x, s1.y};
                    } else {
                        dd[s] = *reinterpret_cast<const v4f *>(in + o);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < LPV; ++s) {
                const float wq[4] = {ww[s].x, ww[s].y, ww[s].z, ww[s].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[q][0].x = fmaf(dd[s].x, wq[q], acc[q][0].x);                      // src/tron.cu:519
                    acc[q][0].y = fmaf(dd[s].y, wq[q], acc[q][0].y);
                    if constexpr (!ONE) {
                        acc[q][1].x = fmaf(dd[s].z, wq[q], acc[q][1].x);
                        acc[q][1].y = fmaf(dd[s].w, wq[q], acc[q][1].y);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the group's weights have been read: the next group may overwrite them
        }
    }

    // ---- the lanes of a coil pair (lane % LPV) hold partial blocks: sum them, the first LPV lanes add theirs to the grid ----
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float vx = acc[q][c].x, vy = acc[q][c].y;
#pragma unroll
            for (int o = 32; o >= LPV; o >>= 1) {
                vx += __shfl_xor(vx, o);
                vy += __shfl_xor(vy, o);
            }
            acc[q][c] = (v2f){vx, vy};
        }
    if (lane < LPV) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < ncl) {
#pragma unroll
                for (int qy = 0; qy < 2; ++qy) {
                    float4 v;
                    v.x = acc[2 * qy][c].x * p.scale;                       // src/tron.cu:532-534
                    v.y = acc[2 * qy][c].y * p.scale;
                    v.z = acc[2 * qy + 1][c].x * p.scale;
                    v.w = acc[2 * qy + 1][c].y * p.scale;
                    store_point_pair<true>(p, z, c0 + c, X0, Y0 + qy, v);
                }
            }
    }
}

template <int LPV, bool ONE, bool HALF>
static hipError_t launch_centre_lpv(const GridParams &p, hipStream_t s)
{
    const int per_chunk = ONE ? 1 : 2 * LPV;
    const int chunks = (p.nchan - p.coil0 + per_chunk - 1) / per_chunk;
    const long long per_xcd = (long long)p.cen_ngroups * ((p.nslices + 7) / 8);           // (block, slice) pairs of one XCD: slice z lives on XCD z % 8
    dim3 grid((unsigned)(8 * ((per_xcd + kCenWaves - 1) / kCenWaves)), (unsigned)chunks);
    hipLaunchKernelGGL((grid_centre_kernel<LPV, ONE, HALF>), grid, dim3(64 * kCenWaves), 0, s, p);
    return hipGetLastError();
}

// Adds the samples |r| < p.inner_r0 to the grid the arc kernel has stored (same stream, behind it); the same plans as the arc kernel.
hipError_t launch_grid_centre(const GridParams &p, int half_in, hipStream_t s)
{
    if (p.out_p != 1 || p.inner_r0 <= 0 || p.inner_r0 > 16 || p.W > 3.0f || !p.cen_win || !p.cen_order || !p.cen_cs || !p.cen_groups || !p.kb_lut || p.nro != p.nxos || p.npe > 65535)
        return hipErrorInvalidValue;
    const int nc = p.nchan - p.coil0;
    if (nc == 1) return half_in ? launch_centre_lpv<1, true, true>(p, s) : launch_centre_lpv<1, true, false>(p, s);
    if (nc <= 2) return half_in ? launch_centre_lpv<1, false, true>(p, s) : launch_centre_lpv<1, false, false>(p, s);
    if (nc <= 4) return half_in ? launch_centre_lpv<2, false, true>(p, s) : launch_centre_lpv<2, false, false>(p, s);
    return half_in ? launch_centre_lpv<4, false, true>(p, s) : launch_centre_lpv<4, false, false>(p, s);
}

__global__ void warm_grid_centre_tu() {}

hipError_t warm_grid_centre()   // see warm_kernels() in tron_kernels.hip
{
    hipLaunchKernelGGL(warm_grid_centre_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
