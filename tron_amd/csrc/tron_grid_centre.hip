// The k-space centre of the TRON_KB_FAST gridding path: samples |r| < inner_r0 (14), which the arc kernel (tron_grid_arc.hip)
// leaves out of the four tiles that meet at the origin.
//
// Same (sample, point) pairs and the same arithmetic per pair as the reference's gridradial2d (src/tron.cu:465-536): kx, ky
// :514-515 with the SIGNED radius r of both loops (:512 and :521), weights :516, band :498-502, density compensation :412-414,
// scale :532, r = 0 counted twice where the band starts at 0 (:512 and :521 both visit it).
//
// Near the origin every spoke passes every point: a 2x2 block there meets ~400 spokes where a block of the arc kernel meets 6, so
// "thread = block" (arc kernel) starves and "sort the samples by cell" (binned kernel, which did this job until round 3 at 0.30
// lanes active and 0.28 ms per 1 024 coil-slices for 5 % of the samples) spends its time sorting.  Here one WAVE works on one 2x2
// block of one slice at a time (an ITEM) and its lanes share out the block's VISITS (sample, block):
//   items   the workgroups stay for the launch (as many as the chip holds) and every wave draws its items from a counter of its
//           XCD; the next draw is asked for when an item starts and read when it ends;
//           a busy block (every spoke passes the ones at the origin, 50 the ones at the rim) is up to four items in launches of few
//           slices, one per run of its window: each leaves its sums in a slot, the last to finish adds them up in part order;
//   window  the block's run of the angle-sorted spoke list (the arc kernel's rule; worked out by the host at plan creation);
//   chunk   64 spokes of the run, one per lane (the chunk after it already requested): clip against the block's footprint in SIGNED
//           r (so no run ever "wraps"; |r| < inner_r0); an exclusive scan of the chord lengths (DPP) numbers the chunk's visits, and
//           every spoke writes its record and its lane number under each of its visits into the wave's LDS;
//   weights 64 visits at a time, one per lane: owner -> spoke record -> (kx, ky) -> the arc kernel's weights (pair table in LDS,
//           band masks, density compensation, r = 0 doubled): the four point weights and the sample's address go to LDS;
//   sums    the same 64 visits, LPV lanes per visit (one coil pair each, 16 bytes: the LPV lanes of a visit read ONE 64-byte
//           line, 16 lines per wave instruction instead of 64 -- a version with one lane per visit and all coils per lane spent 450
//           addresser cycles per 64 visits and ran at the binned kernel's 0.2-0.3 ms), all loads of the group in flight together,
//           8 packed FMAs per visit and lane;
//   end     the lanes' partial blocks cross the lanes through the wave's LDS, one lane per value sums them in lane order and ADDS
//           the sum to what the arc kernel stored: this kernel runs behind it on the same stream.  No floating-point atomics, no
//           partial tiles in HBM, the same sums in the same order every run.
// The work is a chain of latencies per wave (phase clock, tools/cenprof.py: ~2 000 clocks per chunk, ~2 500-3 000 per group of
// 64 visits, 25 000-35 000 per item), so what counts is how many waves a CU holds and that none of them idles: everything
// wave-uniform is kept in scalar registers (a wave index read from threadIdx is not, unless told).  Round 4, per 128 slices of 8
// coils: one workgroup per four items 153 us (126 registers, 4 waves per SIMD); 84 registers, 5 waves 122; as above 117, and
// 55 instead of 68 for 32 slices.
#include <algorithm>

#include <mutex>

#include "tron_device.h"
#include "tron_host.h"

namespace tron {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kCenTicketWords = 8 * 16;                         // GridParams::cen_ticket: a counter per XCD, 64 bytes apart; then one per busy (slice, chunk, block)
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
// Phase clock (-DTRON_PHASE_CLOCK builds only, tools/cenprof.py): shader-clock cycles per wave and phase, summed over the launch
#ifdef TRON_PHASE_CLOCK
constexpr int kCenProfCopies = 256;
__device__ unsigned long long g_cen_prof[kCenProfCopies * 16];
#define CPROF_DECL unsigned cprof[16] = {}; unsigned long long cprof_t = __builtin_readcyclecounter(); const unsigned long long cprof_0 = cprof_t, cprof_r0 = __builtin_amdgcn_s_memrealtime()
#define CPROF_MARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); cprof[i] += (unsigned)(t_ - cprof_t); cprof_t = t_; } while (0)
#define CPROF_COUNT(i, v) do { cprof[i] += (unsigned)(v); } while (0)
#define CPROF_FLUSH do { cprof[5] = (unsigned)(__builtin_readcyclecounter() - cprof_0); cprof[12] = 1; cprof[10] = (unsigned)(__builtin_amdgcn_s_memrealtime() - cprof_r0); if (lane == 0) for (int i_ = 0; i_ < 16; ++i_) if (cprof[i_]) atomicAdd(&g_cen_prof[((blockIdx.x * 4 + wave) % kCenProfCopies) * 16 + i_], (unsigned long long)cprof[i_]); } while (0)
#else
#define CPROF_DECL
#define CPROF_MARK(i)
#define CPROF_COUNT(i, v)
#define CPROF_FLUSH
#endif

// inclusive prefix sum over the 64 lanes on the DPP path (row shifts, then the row broadcasts of gfx9): 6 VALU instructions where
// __shfl_up is a round trip through the LDS crossbar each
__device__ __forceinline__ int wave_incl_scan_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return v;
}

template <int LPV, bool ONE, bool HALF>
__global__ void __launch_bounds__(64 * kCenWaves)
grid_centre_kernel(const GridParams p)
{
    static_assert(LPV == 1 || LPV == 2 || LPV == 4, "lanes per visit");
    static_assert(!ONE || LPV == 1, "one coil: one lane per visit");
    constexpr int NC = ONE ? 1 : 2;                             // coils per lane
    constexpr int VPS = 64 / LPV;                               // visits per sub-step
    constexpr int NV = 8 * NC;                                  // sums per lane: 4 points x NC coils x (re, im)
    static_assert(8 * (64 + LPV) * sizeof(float) <= sizeof(CenWaveLds), "the block's sums cross the lanes through the wave's list area");
    __shared__ CenLds lds;
    CPROF_DECL;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (scalar: what follows from it -- block, slice, window -- stays out of the vector registers)
    CenWaveLds &L = lds.w[wave];
    for (int i = threadIdx.x; i < 3 * kArcLutEntries; i += 64 * kCenWaves) lds.lut[i] = p.kb_lut[i];
    __syncthreads();                                            // (the only barrier: once per workgroup, which then stays for the launch)
    // Work items (block, slice, coil chunk) are handed out per XCD through a ticket counter: slice z lives on XCD z % 8
    // (workgroups go round the 8 XCDs).  Block-major, the block nearest the origin (most visits: 400 spokes against 50 at the rim)
    // first, so that the launch ends on its cheapest items; the slices of a block run side by side, and neighbouring blocks, which
    // share most of their samples, follow each other.  (Measured per 128 slices of 8 coils: slice-major 190 us; the 32 nearest
    // blocks of every slice, then the rest, each pass slice-major, 132; block-major 117.  Draws of 2 / 4 / 8 items: 127 / 192 / 283.)
    const int xcd = blockIdx.x & 7;
    const int per_chunk = ONE ? 1 : 2 * LPV;
    const unsigned chunks = (unsigned)((p.nchan - p.coil0 + per_chunk - 1) / per_chunk);
    const unsigned ngroups = (unsigned)p.cen_ngroups;
    const unsigned zc = xcd < p.nslices ? (unsigned)((p.nslices - xcd + 7) / 8) * chunks : 0u;     // (slice, chunk) pairs of this XCD
    const unsigned magic_zc = p.cen_magic_zc[xcd < (p.nslices & 7) ? 0 : 1];
    const unsigned nitems = zc * ngroups;
    unsigned *const ticket = p.cen_ticket + 16 * xcd;
    const int cp = lane % LPV;                                  // this lane's coil pair of the chunk
    const int n = p.nxos, h = n / 2;
    const int npe = p.npe;
    const v2f lscale2 = {p.lut_scale, p.lut_scale};
    const float We = p.W + 1e-3f;
    const float dcf_a = p.apply_dcf ? p.dcf_a : 0.0f, dcf_b = p.apply_dcf ? p.dcf_b : 1.0f;
    const float2 *lut = lds.lut + p.lut_bias;                   // entry of table position 0
    const float lut_lo = -(float)p.lut_bias;                    // the table's first position (pieces that far out are zero)
    const unsigned nchan8 = (unsigned)p.nchan * (HALF ? 4u : 8u);
    const float rs_nro = (float)p.nro, rs_inv = 1.0f / (float)p.nxos;
    CPROF_MARK(0);                                              // table

    // a wave's first item is its own number on the XCD, the tickets go on from there: no wave waits for the counter before it starts
    const unsigned nwaves = (gridDim.x >> 3) * kCenWaves;
    unsigned item = (blockIdx.x >> 3) * kCenWaves + (unsigned)wave;
    unsigned tk = 0u;
    while (item < nitems) {
    if (lane == 0) tk = nwaves + atomicAdd(ticket, 1u);         // the next item's ticket: on its way while this item is worked on
    const int gi = (int)(zc > 1 ? __umulhi(item, magic_zc) : item);               // item = block * (slices of this XCD * chunks) + slice * chunks + chunk; scalar arithmetic
    const unsigned rest = item - (unsigned)gi * zc;
    const unsigned zi = chunks > 1 ? __umulhi(rest, p.cen_magic_chunks) : rest;
    const int z = (int)zi * 8 + xcd;
    const int cbase = p.coil0 + (int)(rest - zi * chunks) * per_chunk;                     // the chunk's first coil (scalar); this lane's: + 2 * cp
    const bool have = ONE || cbase + 2 * cp < p.nchan;          // this lane has a coil at all
    // block record (build at plan creation from the band table, src/tron.cu:498-502): (col | row << 8) of the origin-centred
    // 32 x 32 square, the largest |r| < inner_r0 that reaches it; the band of its four points as masks over |r|
    const uint4 g0 = p.cen_grec[2 * gi], g1 = p.cen_grec[2 * gi + 1];
    const int grp = (int)g0.x, rcap = (int)g0.y;
    const int part = (int)(g0.z & 255u), nparts = (int)((g0.z >> 8) & 255u), heavy = (int)(g0.z >> 16), greal = (int)g0.w;
    const unsigned bmask[4] = {g1.x, g1.y, g1.z, g1.w};
    const int X0 = 2 * (grp & 255) - 16, Y0 = 2 * (grp >> 8) - 16;

    // where this lane's share of the block lies in the grid (the first NV * LPV lanes end up with one sum each)
    auto grid_point = [&]() -> float * {
        if (lane >= NV * LPV) return nullptr;
        const int k = lane / LPV, q = k / (2 * NC), cbit = ONE ? 0 : (k >> 1) & 1;
        const int X = X0 + (q & 1), Y = Y0 + (q >> 1);
        const int c0 = cbase + (ONE ? 0 : 2 * cp);
        if (c0 + cbit >= p.nchan || X + h >= n || Y + h >= n) return nullptr;
        const int row = p.out_shift ? (Y < 0 ? Y + n : Y) : Y + h;         // both fftshifts of src/tron.cu:631 folded in (store_point_pair)
        const int col = p.out_shift ? (X < 0 ? X + n : X) : X + h;
        return reinterpret_cast<float *>(p.udata + (size_t)z * p.out_z + ((size_t)row * n + col) * p.out_p + (size_t)(c0 + cbit) * p.out_c) + (k & 1);
    };

    const size_t win = (size_t)z * p.arc_slice_stride;
    const unsigned short *order = p.cen_order + win * npe;
    const float2 *scs = p.cen_cs + win * npe;
    const unsigned wnd = p.cen_win[win * p.cen_nblocks + greal];  // first spoke | spokes << 16 (build_centre_windows, at plan creation)
    int jstart = (int)(wnd & 0xffffu), cnt = (int)(wnd >> 16);
    if (nparts > 1) {                                           // this item's share of the block's window: part `part` of `nparts` equal runs
        const int per = (cnt + nparts - 1) / nparts;            // (nparts <= 4: a handful of scalar instructions)
        jstart += part * per;
        if (jstart >= npe) jstart -= npe;
        cnt = max(0, min(per, cnt - part * per));
    }

    // (wave-uniform floats: a conversion or an addition is a vector instruction whatever its operands, so its result is moved to a
    // scalar register by hand; seven vector registers less over the item)
    auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
    const float X0f = uni((float)X0), Y0f = uni((float)Y0);
    const v2f p0v = {X0f, Y0f}, p1v = {X0f + 1.0f, Y0f + 1.0f};
    const float xlo = uni(X0f - We), xhi = uni(X0f + 1.0f + We), ylo = uni(Y0f - We), yhi = uni(Y0f + 1.0f + We);
    const float rcap_f = uni((float)rcap);
    const unsigned char *in = reinterpret_cast<const unsigned char *>(p.nudata) + (size_t)z * (size_t)p.in_slice_stride * (HALF ? 4 : 8);   // (scalar)
    const unsigned coff = (unsigned)(cbase + (ONE ? 0 : 2 * cp)) * (HALF ? 4u : 8u);      // this lane's first coil within a sample

    v2f acc[4][NC];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[q][c] = (v2f){0.f, 0.f};
    CPROF_MARK(11);                                             // item: ticket, block record, window

    // The visits of the window are taken 64 at a time (a GROUP: one visit per lane for the weights, LPV lanes per visit for the sums).
    // (Requesting a group's samples and summing the group before it meanwhile -- two register sets -- was measured: 135-175 us per 128
    // slices against 123; it costs a wave per SIMD, and the waves are what hides this kernel's latencies.)
    int t0 = -64, v0 = 0, total = 0;                            // chunk (64 spokes of the window), group, visits of the chunk
    // ---- next chunk: this lane's spoke, clipped; its visits numbered ----
    float2 cs_next = make_float2(1.f, 0.f);
    unsigned pe_next = 0u;
    auto fetch_spoke = [&](const int tc) {
        if (tc + lane < cnt) {
            int j = jstart + tc + lane;
            if (j >= npe) j -= npe;
            cs_next = scs[j];
            pe_next = order[j];
        }
    };
    fetch_spoke(0);
    auto clip_chunk = [&]() {
        const int t = t0 + lane;
        int len = 0, ra = 0;
        const float2 cs = cs_next;
        const unsigned pe = pe_next;
        fetch_spoke(t0 + 64);                                                                 // the chunk after this one: in flight over this chunk's groups
        if (t < cnt) {
            const float ic = __builtin_amdgcn_rcpf(cs.x), is = __builtin_amdgcn_rcpf(cs.y);   // (1 / 0 = inf clips like a huge number; the box edges are never 0)
            const float xa = xlo * ic, xb = xhi * ic;
            const float ya = ylo * is, yb = yhi * is;
            const float lo = fmaxf(fmaxf(fminf(xa, xb), fminf(ya, yb)), -rcap_f);
            const float hi = fminf(fminf(fmaxf(xa, xb), fmaxf(ya, yb)), rcap_f);
            ra = (int)ceilf(lo);
            const int rb = (int)floorf(hi);
            len = rb >= ra ? min(rb - ra + 1, 12) : 0;                                       // (never cuts: see CenWaveLds::owner)
        }
        const int incl = wave_incl_scan_add(len);
        const int excl = incl - len;
        total = __builtin_amdgcn_readlane(incl, 63);
        L.rec[lane] = make_uint4(__float_as_uint(cs.x), __float_as_uint(cs.y), (unsigned)excl, ((unsigned)(ra + 32) << 16) | pe);
        for (int i = 0; i < len; ++i) L.owner[excl + i] = (unsigned char)lane;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (one wave: its own LDS writes are visible once they have completed)
        CPROF_MARK(1); CPROF_COUNT(6, 1); CPROF_COUNT(9, total);
    };
    // the next group with visits in it (false: the window is done); wave-uniform
    auto next_group = [&]() -> bool {
        v0 += 64;
        while (v0 >= total) {
            t0 += 64;
            if (t0 >= cnt) return false;
            clip_chunk();
            v0 = 0;
        }
        return true;
    };
    // ---- weights of visits v0 .. v0 + 63, one per lane; then, LPV lanes per visit, the samples are requested ----
    auto issue = [&](v4f (&dd)[LPV]) {
        const int v = v0 + lane;
        float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned off = 0u;
        if (v < total) {
            const uint4 rec = L.rec[L.owner[v]];
            const int ri = (int)((rec.w >> 16) & 63u) - 32 + (v - (int)rec.z);
            // sample of radius r on spoke pe: nudata[nchan * (nro * pe + (r nro) / nxos + nro / 2) + c]   src/tron.cu:517,519 (truncating
            // towards zero; r itself when nro == nxos)
            const float rf = (float)ri;
            const float sf = arc_sample_of(fabsf(rf), rs_nro, rs_inv);
            off = (unsigned)(p.nro * (int)(rec.w & 0xffffu) + p.nro / 2 + (ri < 0 ? -(int)sf : (int)sf)) * nchan8;
            // (kx, ky) = r (cos, sin) and the distances to the block's first column / row, op for op src/tron.cu:514-516
            const v2f kxy = (v2f){rf, rf} * (v2f){__uint_as_float(rec.x), __uint_as_float(rec.y)};
            // The distance to EACH of the block's two columns / rows by the reference's own subtraction, k - X (src/tron.cu:516), and a table
            // position of its own: the second one derived from the first (one lookup of the pair table, rounds 4-6) is k - X0 rounded
            // at the magnitude of a number up to 3 where the reference rounds k - (X0 + 1) at one up to 2 -- k = 0.99999988 (the r = 1
            // sample of a spoke 5e-4 rad off an axis: one slice in three has such a spoke), X0 = -2: k - X0 rounds to 3, "its" k - X1 = 2
            // lies outside the window, the reference's 1.99999988 inside, and the window jumps by 6e-4 of its peak there: 1.1e-5 relative
            // L2 on k-space whose energy sits at the centre (tests/test_gpu_headline.py, found in round 6).
            const v2f tpa = (kxy - p0v) * lscale2, tpb = (kxy - p1v) * lscale2;
            const v2f tta = {__builtin_truncf(tpa.x), __builtin_truncf(tpa.y)};
            const v2f ttb = {fmaxf(__builtin_truncf(tpb.x), lut_lo), fmaxf(__builtin_truncf(tpb.y), lut_lo)};     // (the far side of the second column: below the table, zero)
            const v2f fa = tpa - tta, fb = tpb - ttb;
            const float2 *lxa = lut + (int)tta.x, *lya = lut + (int)tta.y, *lxb = lut + (int)ttb.x, *lyb = lut + (int)ttb.y;
            const float2 x0c = {lxa[0].x, lxb[0].x}, x1c = {lxa[kArcLutEntries].x, lxb[kArcLutEntries].x}, x2c = {lxa[2 * kArcLutEntries].x, lxb[2 * kArcLutEntries].x};
            const float2 y0c = {lya[0].x, lyb[0].x}, y1c = {lya[kArcLutEntries].x, lyb[kArcLutEntries].x}, y2c = {lya[2 * kArcLutEntries].x, lyb[2 * kArcLutEntries].x};
            const int ar = ri < 0 ? -ri : ri;
            // src/tron.cu:412 (|ro - nro/2|); r = 0 is visited by both loops of the reference where the band starts at 0
            const float sdc = fmaf(dcf_a, sf, dcf_b) * (ri == 0 ? 2.0f : 1.0f);
            const v2f fxv = {fa.x, fb.x}, fyv = {fa.y, fb.y}, sdcv = {sdc, sdc};
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
        asm volatile("" ::: "memory");                          // (a wave's LDS operations complete in the order they were issued)
        unsigned o[LPV];
#pragma unroll
        for (int s = 0; s < LPV; ++s) o[s] = L.off[s * VPS + lane / LPV] + coff;
#pragma unroll
        for (int s = 0; s < LPV; ++s) {
            const int vi = s * VPS + lane / LPV;
            dd[s] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (have && v0 + vi < total) {
                if constexpr (ONE) {
                    const float2 s0 = load_sample<HALF>(in + o[s], 0);
                    dd[s] = (v4f){s0.x, s0.y, 0.f, 0.f};
                } else if constexpr (HALF) {
                    const float2 s0 = load_sample<true>(in + o[s], 0), s1 = load_sample<true>(in + o[s], 1);
                    dd[s] = (v4f){s0.x, s0.y, s1.x, s1.y};
                } else {
                    dd[s] = *reinterpret_cast<const v4f *>(in + o[s]);
                }
            }
        }
        asm volatile("" ::: "memory");
        CPROF_MARK(2); CPROF_COUNT(7, 1);
    };
    // ---- sums of a group whose samples were requested one group ago ----
    auto consume = [&](const v4f (&dd)[LPV]) {
#pragma unroll
        for (int s = 0; s < LPV; ++s) {
            const float4 ww = L.wq[s * VPS + lane / LPV];
            const float wq[4] = {ww.x, ww.y, ww.z, ww.w};
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
        CPROF_MARK(3);
    };
    {
        v4f dd[LPV];
        while (next_group()) { issue(dd); consume(dd); }
    }

    // ---- the lanes of a coil pair (lane % LPV) hold partial blocks: the sums cross the lanes through the wave's LDS (its lists are
    //      done with), eight values at a time, value k of lane l at [k][l] of rows 64 + LPV floats long, so that neither the writes
    //      nor the reads below meet on a bank; lane (k, coil pair) adds the 64 / LPV partial sums of value k in lane order and adds
    //      the result to the grid
    item = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);   // (the next ticket: asked for a whole item ago)
    {
        constexpr int ROW = 64 + LPV;
        float *sc = reinterpret_cast<float *>(&L);
        float sum = 0.f;
#pragma unroll
        for (int half = 0; half < NV / 8; ++half) {
            asm volatile("" ::: "memory");                      // (a wave's LDS operations complete in the order they were issued)
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) {
                const int k = 8 * half + k8;
                const v2f a = acc[k / (2 * NC)][(k >> 1) % NC];
                sc[k8 * ROW + lane] = (k & 1) ? a.y : a.x;
            }
            asm volatile("" ::: "memory");
            if (lane / (8 * LPV) == half) {
                const float *row = sc + ((lane / LPV) & 7) * ROW + cp;
#pragma unroll 1
                for (int j0 = 0; j0 < VPS; j0 += 16)              // (16 reads in flight at a time: all 64 of the one-coil-pair form cost 60 registers)
#pragma unroll
                    for (int j = j0; j < j0 + 16; ++j) sum += row[j * LPV];
            }
        }
        asm volatile("" ::: "memory");
        if (nparts > 1) {
            // A busy block is worked on as `nparts` items (runs of its window): each leaves its sums in a slot of its own, and
            // whichever finishes LAST adds all of them up in part order and adds the total to the grid -- the same bits whatever the
            // order the parts finished in.  (The parts of a block are drawn from one XCD's counter: one L2.)
            const size_t slot = ((size_t)z * chunks + (rest - zi * chunks)) * (size_t)p.cen_nheavy + (size_t)heavy;
            float *const ps = p.cen_parts + slot * (4 * 64);
            // (device-scope stores and loads, which pass the caches that are not coherent across the chip, and a wait for the stores'
            // acknowledgement before the counter is touched.  A release fence would do the same and write back the whole L2 besides,
            // the grid lines this kernel has just dirtied included: measured 872 us per launch instead of 104.)
            if (lane < NV * LPV) __hip_atomic_store(ps + part * 64 + lane, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned done = 0u;
            if (lane == 0) done = atomicAdd(p.cen_ticket + kCenTicketWords + slot, 1u);
            done = (unsigned)__builtin_amdgcn_readfirstlane((int)done);
            if (done + 1u == (unsigned)nparts) {
                if (lane < NV * LPV) {
                    float total = 0.f;
                    for (int q = 0; q < nparts; ++q) total += __hip_atomic_load(ps + q * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    float *const gpt = grid_point();
                    if (gpt) *gpt = fmaf(total, p.scale, *gpt);     // src/tron.cu:532-534
                }
            }
        } else if (lane < NV * LPV) {
            float *const gpt = grid_point();
            if (gpt) *gpt = fmaf(sum, p.scale, *gpt);               // src/tron.cu:532-534; what the arc kernel stored there + this block
        }
    }
    CPROF_MARK(4);
    CPROF_COUNT(8, 1);
    }   // items
    CPROF_FLUSH;
}

template <int LPV, bool ONE, bool HALF>
static hipError_t launch_centre_lpv(const GridParams &p, hipStream_t s)
{
    // as many workgroups as the chip holds at once (they stay and draw items), fewer when the launch has fewer items
    // (filled under a lock: the workers of tron_recon_radial2d_multi launch from their own threads)
    constexpr int kMaxDev = 64;
    static int wgs_per_xcd[kMaxDev] = {};                                                 // per device
    static std::mutex wgs_lock;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDev) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> guard(wgs_lock);
    if (wgs_per_xcd[dev] == 0) {
        int occ = 0, cus = 0;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, grid_centre_kernel<LPV, ONE, HALF>, 64 * kCenWaves, 0)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        wgs_per_xcd[dev] = std::max(1, occ) * std::max(1, cus / 8);
    }
    const int per_chunk = ONE ? 1 : 2 * LPV;
    const int chunks = (p.nchan - p.coil0 + per_chunk - 1) / per_chunk;
    const long long per_xcd = (long long)p.cen_ngroups * chunks * ((p.nslices + 7) / 8);   // items of the busiest XCD: slice z lives on XCD z % 8
    const long long wgs = std::min<long long>(wgs_per_xcd[dev], (per_xcd + kCenWaves - 1) / kCenWaves);
    if ((e = hipMemsetAsync(p.cen_ticket, 0, (kCenTicketWords + (size_t)p.nslices * chunks * p.cen_nheavy) * sizeof(unsigned), s)) != hipSuccess) return e;
    GridParams q = p;
    for (int i = 0; i < 2; ++i) {                               // XCDs below nslices % 8 hold one slice more than the others
        const unsigned zc = (unsigned)(p.nslices / 8 + 1 - i) * (unsigned)chunks;
        q.cen_magic_zc[i] = zc > 1 ? (unsigned)(0x100000000ull / zc) + 1u : 0u;               // x / d = mulhi(x, 2^32 / d + 1) for x d < 2^32
    }
    q.cen_magic_chunks = chunks > 1 ? (unsigned)(0x100000000ull / (unsigned)chunks) + 1u : 0u;     // (d = 1: see the kernel)
    hipLaunchKernelGGL((grid_centre_kernel<LPV, ONE, HALF>), dim3((unsigned)(8 * wgs)), dim3(64 * kCenWaves), 0, s, q);
    return hipGetLastError();
}

// Adds the samples |r| < p.inner_r0 to the grid the arc kernel has stored (same stream, behind it); the same plans as the arc kernel.
hipError_t launch_grid_centre(const GridParams &p, int half_in, hipStream_t s)
{
    if (p.out_p != 1 || p.inner_r0 <= 0 || p.inner_r0 > 16 || p.W > 3.0f || !p.cen_win || !p.cen_order || !p.cen_cs || !p.cen_grec || !p.cen_ticket || (p.cen_nheavy > 0 && !p.cen_parts) || !p.kb_lut || p.npe > 65535)
        return hipErrorInvalidValue;
    const int nc = p.nchan - p.coil0;
    if (nc == 1) return half_in ? launch_centre_lpv<1, true, true>(p, s) : launch_centre_lpv<1, true, false>(p, s);
    if (nc <= 2) return half_in ? launch_centre_lpv<1, false, true>(p, s) : launch_centre_lpv<1, false, false>(p, s);
    if (nc <= 4) return half_in ? launch_centre_lpv<2, false, true>(p, s) : launch_centre_lpv<2, false, false>(p, s);
    return half_in ? launch_centre_lpv<4, false, true>(p, s) : launch_centre_lpv<4, false, false>(p, s);
}

#ifdef TRON_PHASE_CLOCK
extern "C" __attribute__((visibility("default"))) int tron_debug_cen_profile(unsigned long long *out)   // reads and clears the phase clock (16 slots)
{
    static unsigned long long h[kCenProfCopies * 16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cen_prof), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < 16; ++i) out[i] = 0;
    for (int i = 0; i < kCenProfCopies * 16; ++i) { out[i % 16] += h[i]; h[i] = 0; }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_cen_prof), h, sizeof(h)) != hipSuccess;
}
#endif

__global__ void warm_grid_centre_tu() {}

hipError_t warm_grid_centre()   // see warm_kernels() in tron_kernels.hip
{
    hipLaunchKernelGGL(warm_grid_centre_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
