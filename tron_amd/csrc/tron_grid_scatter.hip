// Gridding kernel of the TRON_KB_FAST path for ONE channel per pass (any odd channel count, one by one) or TWO: every lane takes a
// SAMPLE and adds it to the 4x4 grid points of its Kaiser-Bessel footprint in a tile of 64-bit fixed-point sums in LDS (integer atomics).
//
// Why a second formulation.  grid_arc_kernel (thread = 2x2 grid points, the samples come to it) spends ~34 VALU instructions per
// (sample, block) visit of which only 4-8 depend on the coils, visits every sample from ~6 blocks, and runs its radius loop at 0.57
// of its lanes: at one coil it is VALU- and LDS-issue-bound at 0.12 of the HBM roofline (DESIGN.md 4.1c, round-5 phase clock).  The
// coil-independent work cannot be amortised over larger per-thread blocks (the band test of src/tron.cu:498-502 costs two
// instructions per POINT and visit, so 4x4 points per thread save nothing).  Turned round -- lane = sample -- every sample is
// handled once: positions and the eight window values once, then 16 point updates.  The updates are LDS atomics; float atomics run
// at 3 clocks per LANE on this chip (profiles/round1_lds_atomic_rates.txt), 64-bit INTEGER ones hide completely under the arithmetic
// (tools/probe/scatter.hip: 154 clocks per 64 samples and CU with ds_add_u64, 172 for the arithmetic alone) and are exact when
// lanes of one instruction meet on an address (tools/probe/atom64.hip), so the tile holds (re << 32) + im as two 32-bit fixed-point
// numbers per point and channel, scaled per (tile, slice) by a power of two:
//   scale   S = 2^e, the largest with  max(|d| dcf) * M * 4 K(0)^2 * S < 2^31, the maximum over the tile's own samples, M = the most spokes
//           whose line can pass any of the tile's 2x2 blocks (the arc kernel's window rule; arc_prep_kernel leaves it in the run header)
//           and 4 K(0)^2 a bound of the window products one spoke can add to one point: no sum can leave 32 bits.  What is lost is
//           2^-25..2^-23 of the tile's largest weighted sample per added term -- the level of fp32 rounding for data whose magnitude does
//           not vary by orders of magnitude INSIDE one tile (measured: parity_rel_l2_vs_oracle on the bench line; tests/test_gpu_scatter.py).
//   order   integer sums do not depend on the order of the additions: bit-identical results run to run by construction.
// Same (sample, point) pairs as the reference's gridradial2d (src/tron.cu:465-536): |kx - X| < W and |ky - Y| < W strictly (the
// arc kernel's pair table: exact support, build_kb_pair_lut), band Rlo <= |r| <= Rhi (:498-502, :512, :521) as one exact comparison
// checked against the host's band table (scatter_band_is_analytic), density compensation :412-414, scale :532, (kx, ky) = r (cos, sin)
// as two fp32 products (:514-515).
//
// Work per (tile, slice); a tile is 32 x 32 points and four waves, or 64 x 64 and eight (ScatGeom); tables: arc_prep_kernel with ONE batch
// per run (ArcPrepParams::flat), so that the record offsets number the tile's samples 0 .. total-1, plus one byte per record naming
// its member:
//   front    every wave takes an equal share of the records, 64 per iteration, R iterations per round: the lane's (spoke, radius) from
//            the member table (80 bytes per 64 records, copied to LDS by one LDS-DMA instruction a round ahead), its run entry from LDS,
//            its sample by a plain load straight from k-space -- all R loads of a wave in flight together -- into registers;
//   scale    the largest |d| dcf so far through an LDS maximum and a barrier -> S; sums rescaled if it grew by a power of two;
//   scatter  sample -> (kx, ky) -> first column / row of the footprint (the reference's own distance test decides a rounded boundary) -> two
//            pair-table positions per axis -> 4 + 4 window values; d * dcf * S * wx[j] once per column, then per point: * wy[i] (zero outside
//            the band), two conversions, one ds_add_u64 at an immediate offset;
//   store    thread = 2x2 points: (float)sum * 2^-e * scale, once, coil-planar, FFT-native order (as the arc kernel); the tile zeroed and
//            the next slice's run table written for the next slice: three barriers per slice.
// The samples |r| < inner_r0 (5 in this kernel's plans, tron_plan.cpp) are the centre kernel's (tron_grid_centre.hip), behind this one
// on the same stream.
#include <stdlib.h>

#include "tron_device.h"
#include "tron_grid_store.h"

namespace tron {

constexpr int kScatHalo = 4;                        // 2 W for W <= 2: first column of a footprint >= x0 - 2 W
constexpr int kScatMaxSpokes = 512;                 // = kArcMaxSpokes (arc_prep_kernel)

// Tile geometry: 32 x 32 points and four waves, or 64 x 64 and eight (round 5, late: a tile's samples are those within W of it, so a
// larger tile handles fewer samples twice -- (36 / 32)^2 = 1.27 against (68 / 64)^2 = 1.13 -- and the per-slice overheads of a
// workgroup are shared by four times the samples).
template <int TILE>
struct ScatGeom {
    static constexpr int kTile = TILE;
    static constexpr int kPitch = TILE + 2 * kScatHalo;            // points per row and rows of the tile of sums (40 / 72)
    static constexpr int kStride = kPitch + 2;                      // 8-byte words between its rows in LDS (40, 41, 42, 44: within 1.5 %)
    static constexpr int kThreads = TILE == 64 ? 512 : 256;
    static constexpr int kWaves = kThreads / 64;
    static constexpr int kRing = kPitch * kPitch - TILE * TILE;      // the halo ring's points
    static constexpr int kBlocksPerThread = (TILE / 2) * (TILE / 2) / kThreads;      // 2x2 blocks of the store: 1 / 2
};

template <int NC, int TILE>
struct ScatCfg {
    static constexpr int kScatWaves1 = 5, kScatR1 = 6;
    // 32-tiles: workgroups per CU (LDS: 21 / 31 units of 1280 bytes; registers: 96 / 128).  64-tiles: two workgroups of eight waves
    // per CU (LDS 47 units for one channel), i.e. four waves per SIMD.
    // (two channels: 32-tiles only in practice -- the plan does not make 64-tile tables for them: 100 KB of sums)
    static constexpr int WAVES = TILE == 64 ? (NC >= 2 ? 2 : 4) : (NC >= 2 ? 3 : kScatWaves1);
    // iterations (of 64 records per wave) whose samples wait in registers: one round up to 1 536 / 2 048 records per 32-tile
    static constexpr int R = TILE == 64 ? 8 : (NC >= 2 ? 8 : kScatR1);
};

constexpr int kScatLutS = 64;                       // pieces per grid unit of the pair table this kernel copies (kb_pair_lut_scale: 64 for every width it takes)

template <int NC, int TILE>
struct ScatLds {
    using G = ScatGeom<TILE>;
    // The two stretches of the Kaiser-Bessel pair table (build_kb_pair_lut) a sample can reach, planes c0 | c1 | c2: its first column
    // lies W-1 <= d < W from it (positions (W-1) s .. W s), the third d - 2 (positions (W-3) s .. (W-2) s): 2 x 65 entries of 400.
    float2 lutA[3][kScatLutS + 2];
    float2 lutB[3][kScatLutS + 2];
    uint4 run[kScatMaxSpokes];                                   // first sample | down << 31, ulo | len << 10 | (offset & 0x7fff) << 17, cos, sin
    uint32_t off[kScatMaxSpokes];                                // the entries' record offsets inside the run (whole: a 64-tile's run exceeds 15 bits)
    unsigned long long acc[NC][G::kPitch * G::kStride];          // (re << 32) + im, fixed point
    unsigned dmax_bits[2];                                        // largest |d| dcf of the rounds so far, by round parity
    unsigned pad[2];
    float psum[2][G::kWaves];                                    // every wave's sum of |d| dcf over its records of a round, by round parity (round 6)
    // a wave's next round of the member table (arc_prep_kernel: 80 bytes per group of 64 records), copied by LDS-DMA a round ahead
    unsigned char recb[G::kWaves][ScatCfg<NC, TILE>::R * 80];
};

typedef const __attribute__((address_space(3))) v2f *slds_f2p;

// Phase clock of tools/scatprof.py (-DTRON_PHASE_CLOCK builds only): shader-clock cycles per wave and phase, summed over all waves
// of all launches since the last read, plus loop counters; production builds carry none of it.
#ifdef TRON_PHASE_CLOCK
constexpr int kScatProfSlots = 16, kScatProfCopies = 4096;
__device__ unsigned long long g_scat_prof[kScatProfCopies * kScatProfSlots];
#define SPROF_DECL unsigned prof_acc[kScatProfSlots] = {}; unsigned long long prof_t = __builtin_readcyclecounter(); const unsigned long long prof_c0 = prof_t, prof_r0 = __builtin_amdgcn_s_memrealtime()
#define SPROF_MARK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[i] += (unsigned)(t_ - prof_t); prof_t = t_; } while (0)
#define SPROF_COUNT(i, v) do { prof_acc[i] += (unsigned)(v); } while (0)
#define SPROF_FLUSH do { prof_acc[10] = (unsigned)(__builtin_readcyclecounter() - prof_c0); prof_acc[11] = (unsigned)(__builtin_amdgcn_s_memrealtime() - prof_r0); if (lane == 0) { for (int i_ = 0; i_ < kScatProfSlots; ++i_) if (prof_acc[i_]) atomicAdd(&g_scat_prof[((blockIdx.x * 4 + wave) % kScatProfCopies) * kScatProfSlots + i_], (unsigned long long)prof_acc[i_]); } } while (0)
#else
#define SPROF_DECL
#define SPROF_MARK(i)
#define SPROF_COUNT(i, v)
#define SPROF_FLUSH
#endif

__device__ __forceinline__ int cvt_rpi(float x)      // floor(x + 0.5): one instruction (v_cvt_i32_f32 truncates, rndne + cvt are two)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

template <int NC, bool HALF, bool RS, int TILE>
__global__ void __launch_bounds__((ScatGeom<TILE>::kThreads), (ScatCfg<NC, TILE>::WAVES))
grid_scatter_kernel(const GridParams p)
{
    static_assert(NC == 1 || NC == 2, "one or two channels per pass");
    using G = ScatGeom<TILE>;
    constexpr int kScatTile = G::kTile, kScatPitch = G::kPitch, kScatStride = G::kStride, kScatThreads = G::kThreads, kWaves = G::kWaves;
    static_assert((sizeof(ScatLds<NC, TILE>) + 1279) / 1280 * (ScatCfg<NC, TILE>::WAVES * 4 / kWaves) <= 128, "the workgroups per CU that ScatCfg::WAVES asks for do not fit a CU's 128 LDS units of 1280 bytes");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    ScatLds<NC, TILE> &L = *reinterpret_cast<ScatLds<NC, TILE> *>(lds_raw);

    using C = ScatCfg<NC, TILE>;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int zper = p.arc_zper > 0 ? p.arc_zper : 1;
    const int ngroups = (p.nslices + zper - 1) / zper;
    const int zg = blockIdx.x % ngroups;
    const int tile = p.tile_order[blockIdx.x / ngroups] & 0xffff;
    if (tile >= p.ntiles) {
        if (tid == 0) atomicOr(p.errflag, 128u);
        return;
    }
    const int c0 = p.coil0 + blockIdx.y * NC;
    const int ncb = min(NC, p.nchan - c0);
    const int n = p.nxos, h = n / 2, rmax = n / 2 - 1;
    const int x0 = (tile % p.tiles_per_row) * kScatTile - h;      // tile origin, centred coordinates
    const int y0 = (tile / p.tiles_per_row) * kScatTile - h;
    if (p.skip_outside) {
        // nearest point of the tile to the k-space centre; beyond rmax + W every band is empty (src/tron.cu:498-502,512)
        const int ax = max(max(x0, -(x0 + kScatTile - 1)), 0), ay = max(max(y0, -(y0 + kScatTile - 1)), 0);
        const float lim = (float)rmax + p.W + 1.0f;
        if ((float)(ax * ax + ay * ay) > lim * lim) return;
    }

    // ---- once per workgroup: the two stretches of the window table ----
    const int ws = (int)(p.W * p.lut_scale + 0.5f);                   // W s, an integer (kb_pair_lut_scale)
    const int iA0 = ws - kScatLutS, iB0 = ws - 3 * kScatLutS;         // first entries: positions (W - 1) s and (W - 3) s
    for (int i = tid; i < 3 * (kScatLutS + 1); i += kScatThreads) {
        const int pl = i / (kScatLutS + 1), k = i - pl * (kScatLutS + 1);
        L.lutA[pl][k] = p.kb_lut[pl * kArcLutEntries + p.lut_bias + iA0 + k];
        L.lutB[pl][k] = p.kb_lut[pl * kArcLutEntries + p.lut_bias + iB0 + k];
    }
    const slds_f2p lutA = (slds_f2p)(__attribute__((address_space(3))) const void *)&L.lutA[0][0] - iA0;    // entry of table position 0 (virtual)
    const slds_f2p lutB = (slds_f2p)(__attribute__((address_space(3))) const void *)&L.lutB[0][0] - iB0;
    const float W = p.W, lscale = p.lut_scale, two_s = 2.0f * p.lut_scale;
    const float dcf_a = p.apply_dcf ? p.dcf_a : 0.0f, dcf_b = p.apply_dcf ? p.dcf_b : 1.0f;
    const float rs_nro = (float)p.nro, rs_inv = 1.0f / (float)p.nxos;
    const unsigned nchan_b = (unsigned)p.nchan * (HALF ? 4u : 8u);       // bytes per sample (all channels)
    const float fx0 = (float)(x0 - kScatHalo), fy0 = (float)(y0 - kScatHalo);

    // the store's layout: thread = 2x2 points (32-tiles: as in grid_arc_kernel; 64-tiles: two such blocks per thread)
    auto block_origin = [&](const int k, int &mx, int &my) {
        const int blk = tid + k * kScatThreads;
        mx = 2 * (blk % (kScatTile / 2));
        my = 2 * (blk / (kScatTile / 2));
    };
    auto out_offset = [&](const int mx, const int my, const int qy) -> unsigned {
        const int X0 = x0 + mx, Y = y0 + my + qy;
        const int row = p.out_shift ? (Y < 0 ? Y + n : Y) : Y + h;      // both fftshifts of src/tron.cu:631 folded in
        const int col = p.out_shift ? (X0 < 0 ? X0 + n : X0) : X0 + h;
        return (unsigned)(row * n + col) * 8u;
    };

    SPROF_DECL;
    // the run table of a slice (<= 512 entries, two per thread) is asked for one slice ahead and waits in registers
    constexpr int kEnt = kScatMaxSpokes / kScatThreads;           // entries per thread: 2 / 1
    uint4 pf_ent[kEnt];
    uint32_t pf_off[kEnt];
#pragma unroll
    for (int k = 0; k < kEnt; ++k) { pf_ent[k] = make_uint4(0u, 0u, 0u, 0u); pf_off[k] = 0u; }
    int4 hdr_next = make_int4(0, 1, 0, 0);
    int rbase_next = 0;
    auto fetch_table = [&](const int z, const int4 hh) {
        const size_t win = (size_t)z * p.arc_slice_stride;
        const uint4 *ent = p.arc_ent + win * p.arc_cap + hh.z;
        const uint32_t *eoff = p.arc_off + win * p.arc_cap + hh.z;
#pragma unroll
        for (int k = 0; k < kEnt; ++k)
            if (tid + k * kScatThreads < hh.x) { pf_ent[k] = ent[tid + k * kScatThreads]; pf_off[k] = eoff[tid + k * kScatThreads]; }
    };
    // the member table of one round of this wave (R groups of 80 bytes, contiguous: 40 lanes x 16 bytes at R = 8) -> LDS, no registers
    auto copy_groups = [&](const int z, const int rbase_z, const int total_z, const int r0_z) {
        const int quota_z = ((total_z + 64 * kWaves - 1) / (64 * kWaves)) << 6;
        const int g0 = (wave * quota_z >> 6) + r0_z;            // the wave's first group of that round
        if ((g0 << 6) >= total_z) return;                       // (the wave has no records there)
        const unsigned char *src = p.arc_rec + ((size_t)z * p.arc_slice_stride * p.arc_rec_cap + rbase_z) * 80;
        if (lane < (C::R * 80) / 16) lds_dma16_s(src, (unsigned)(g0 * 80 + lane * 16), lds_addr(&L.recb[wave][0]));
    };
    // the tile of sums is all zeros when a slice begins: its halo ring (everything but the 32 x 32 points) is dealt to the threads
    // here and cleared again by each slice's store, which also clears the points it has just read
    auto halo_index = [&](const int k) -> int {                 // k-th element of the ring, 0 <= k < 40^2 - 32^2
        constexpr int top = kScatHalo * kScatPitch;             // 4 full rows above, 4 below, 2 x 4 columns beside the 32 rows
        if (k < 2 * top) {
            const int kk = k < top ? k : k - top, row = kk / kScatPitch, col = kk - row * kScatPitch;
            return (k < top ? row : kScatPitch - kScatHalo + row) * kScatStride + col;
        }
        const int m = k - 2 * top, row = m >> 3, c8 = m & 7;
        return (kScatHalo + row) * kScatStride + (c8 < kScatHalo ? c8 : kScatTile + c8);
    };
    constexpr int kRing = G::kRing;
    auto table_to_lds = [&](const int ns_) {
#pragma unroll
        for (int k = 0; k < kEnt; ++k) {
            const int i = tid + k * kScatThreads;
            if (i < ns_) { L.run[i] = pf_ent[k]; L.off[i] = pf_off[k]; }
        }
    };
    if (zg * zper < p.nslices) {
        hdr_next = p.arc_hdr[(size_t)(zg * zper) * p.arc_slice_stride * p.ntiles + tile];
        rbase_next = p.arc_rbase[(size_t)(zg * zper) * p.arc_slice_stride * p.ntiles + tile];
        fetch_table(zg * zper, hdr_next);
        table_to_lds(hdr_next.x);
        copy_groups(zg * zper, rbase_next, hdr_next.w, 0);
    }
    {
        uint4 *const a4 = reinterpret_cast<uint4 *>(&L.acc[0][0]);
        constexpr int N4 = NC * kScatPitch * kScatStride / 2;
        for (int i = tid; i < N4; i += kScatThreads) a4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 2) L.dmax_bits[tid] = 0u;
    }

    for (int iz = 0; iz < zper; ++iz) {
        const int z = zg * zper + iz;
        if (z >= p.nslices) break;
        const int4 hdr = hdr_next;
        const int mwin = hdr.y, total = hdr.w, rbase = rbase_next;
        const bool more = iz + 1 < zper && z + 1 < p.nslices;
        if (more) {
            hdr_next = p.arc_hdr[(size_t)(z + 1) * p.arc_slice_stride * p.ntiles + tile];
            rbase_next = p.arc_rbase[(size_t)(z + 1) * p.arc_slice_stride * p.ntiles + tile];
        }
        const unsigned char *in = reinterpret_cast<const unsigned char *>(p.nudata) + ((size_t)z * (size_t)p.in_slice_stride + c0) * (HALF ? 4 : 8);

        SPROF_MARK(0);                                          // set-up (first slice: window table), table -> LDS
        __syncthreads();                                        // this slice's run table is in LDS, the sums are zero
        SPROF_MARK(1);

        // lane's sample of member entry e at offset k inside the segment -> byte offset of its first channel
        auto sample_off = [&](const uint4 e, const int k) -> unsigned {
            const unsigned first = e.x & 0x7fffffffu;
            unsigned so;
            if constexpr (RS) {                                  // `first` is the spoke's centre sample: radius u reads sample (u nro) / nxos (src/tron.cu:517)
                const unsigned t = (unsigned)(int)arc_sample_of((float)((int)(e.y & 1023u) + k), rs_nro, rs_inv);
                so = (e.x >> 31) ? first - t : first + t;
            } else {
                so = (e.x >> 31) ? first - (unsigned)k : first + (unsigned)k;
            }
            return so * nchan_b;
        };
        auto load_d = [&](const unsigned off, v2f (&d)[NC]) {
            if constexpr (HALF) {
                if constexpr (NC == 1) {
                    const __half2 hv = *reinterpret_cast<const __half2 *>(in + off);
                    const float2 f = __half22float2(hv);
                    d[0] = (v2f){f.x, f.y};
                } else {
                    const uint2 raw = *reinterpret_cast<const uint2 *>(in + off);
                    __half2 h0, h1;
                    __builtin_memcpy(&h0, &raw.x, 4);
                    __builtin_memcpy(&h1, &raw.y, 4);
                    const float2 f0 = __half22float2(h0), f1 = __half22float2(h1);
                    d[0] = (v2f){f0.x, f0.y};
                    d[1] = (v2f){f1.x, f1.y};
                }
            } else if constexpr (NC == 1) {
                const float2 f = *reinterpret_cast<const float2 *>(in + off);
                d[0] = (v2f){f.x, f.y};
            } else {
                const float4 f = *reinterpret_cast<const float4 *>(in + off);
                d[0] = (v2f){f.x, f.y};
                d[1] = (v2f){f.z, f.w};
            }
        };

        // This wave's quarter of the run's records, 64 per iteration, R iterations per ROUND: a round first finds every lane's
        // (spoke, radius) and asks for its sample -- all R loads of a wave in flight together, one memory latency per round -- then
        // agrees on the largest density-compensated |re|, |im| so far (-> the fixed-point scale; the sums are rescaled if it grew
        // by a power of two), then scatters from registers.  Most tiles are one round.
        constexpr int R = C::R;
        const int quota = ((total + 64 * kWaves - 1) / (64 * kWaves)) << 6;
        const int iters = quota >> 6;                           // per wave, the same for all of them
        const int pbeg = wave * quota, pend = min(total, pbeg + quota);
        SPROF_MARK(2);                                          // first member of the wave's quarter
        float run_max = 0.f;                                    // largest weighted sample of the rounds so far (workgroup-uniform)
        float run_sum = 0.f;                                    // ... and the sum of all of them
        int e2 = 0;
        bool have_scale = false;
        float S = 1.0f, invS = 1.0f;
        for (int r0 = 0; r0 < iters; r0 += R) {
            // ---- front: (spoke, radius) of every lane, samples requested ----
            v2f dreg[R][NC];
            unsigned meta[R];                                   // member | radius << 9 | valid << 31
            {
                // The lane's member: looked up, not searched -- arc_prep_kernel left, per group of 64 records, one byte per record (its member
                // minus the member that holds the group's first record) and that member; this round's groups were copied to LDS by LDS-DMA
                // while the round before (or the slice before) was scattered.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the copy has landed (it was issued a round ago)
                const unsigned char *const grp = &L.recb[wave][0];
#pragma unroll
                for (int q = 0; q < R; ++q) {
                    meta[q] = 0u;
                    const int pbase = pbeg + (r0 + q) * 64;
                    if (r0 + q < iters && pbase < pend) {
                        SPROF_COUNT(12, 1);
                        const int pos = pbase + lane;
                        const unsigned mem = (unsigned)*reinterpret_cast<const unsigned short *>(grp + q * 80 + 64) + (unsigned)grp[q * 80 + lane];
                        meta[q] = mem | ((unsigned)pos << 9);            // (the record's position for now)
                    }
                }
                SPROF_MARK(3);                                  // walk
                // the members' entries (independent LDS reads), then the samples (independent loads)
#pragma unroll
                for (int q = 0; q < R; ++q) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) dreg[q][c] = (v2f){0.f, 0.f};
                    const int pbase = pbeg + (r0 + q) * 64;
                    if (r0 + q < iters && pbase < pend) {
                        const int mem = (int)(meta[q] & 511u), pos = (int)(meta[q] >> 9);
                        const uint4 e = L.run[mem];
                        const int off = (int)L.off[mem], len = (int)((e.y >> 10) & 127u);
                        const int k = pos - off;
                        meta[q] = 0u;
                        if (pos < pend && k >= 0 && k < len) {
                            meta[q] = (unsigned)mem | ((unsigned)((int)(e.y & 1023u) + k) << 9) | 0x80000000u;
                            load_d(sample_off(e, k), dreg[q]);
                        }
                    }
                }
            }
            SPROF_MARK(9);                                      // entries, loads issued
            // ---- largest density-compensated |re|, |im| ----
            float mxv = 0.f, msum = 0.f;
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if (meta[q] >> 31) {
                    const float uf = (float)((meta[q] >> 9) & 1023u);
                    const float sdc = fabsf(fmaf(dcf_a, RS ? arc_sample_of(uf, rs_nro, rs_inv) : uf, dcf_b));
                    float mq = 0.f;
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if (c < ncb) mq = fmaxf(mq, fmaxf(fabsf(dreg[q][c].x), fabsf(dreg[q][c].y)) * sdc);
                    if (!(mq < 3.0e38f)) mq = 3.0e38f;          // inf / NaN in the data: garbage either way; keep the scale finite
                    mxv = fmaxf(mxv, mq);
                    msum += mq;                                 // (in q order: the same sum every run)
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mxv = fmaxf(mxv, __shfl_xor(mxv, o));
                msum += __shfl_xor(msum, o);                    // xor butterfly of a commutative operation: the same bits in every lane
            }
            const int slot = (r0 / R) & 1;                      // (two slots, never cleared inside a slice: a fast wave's next round cannot disturb this one)
            if (lane == 0) {
                atomicMax(&L.dmax_bits[slot], __float_as_uint(fmaxf(mxv, run_max)));
                L.psum[slot][wave] = fminf(msum, 3.0e38f);
            }
            SPROF_MARK(4);                                      // front: walk, loads, maximum
            __syncthreads();
            SPROF_MARK(5);
            const float new_max = __uint_as_float(L.dmax_bits[slot]);
            // S = 2^e, the largest with  bound * S < 2^31,  bound = what the |re| or |im| sum of ONE point can reach, the smaller of
            //   (a) M x the tile's largest |d| dcf x 1.75 K(0)^2: M = the most spokes whose line can pass one of the tile's 2x2 blocks (arc_prep_kernel:
            //       the run header), and one spoke adds at most its largest sample times the window products along a line (<= 1.65 K(0)^2);
            //   (b) K(0)^2 x the SUM of |d| dcf over every record of the tile so far: a sample adds at most K(0)^2 of itself to a point.
            // Until round 5 only (a) stood, and ONE large sample (a spike 300 x its neighbourhood in a 640-spoke window: 1.8e-5 relative L2 against
            // the oracle, tests/test_gpu_scatter.py) set the step of the whole tile; (b) charges a spike as one sample, and data that falls off
            // with the radius as the sum it is.  On flat data (a) is the smaller one, as before.  Deterministic: every wave's sum is taken in a
            // fixed order, the waves' sums are added in wave order by everybody.
            float rsum = 0.f;
#pragma unroll
            for (int ww = 0; ww < kWaves; ++ww) rsum += L.psum[slot][ww];
            run_sum = fminf(run_sum + rsum, 3.0e38f);
            {
                const float bound = fminf(new_max * (float)max(mwin, 1) * p.scat_wsum, run_sum * p.scat_wmax);
                int e_new = have_scale ? e2 : 0;
                if (bound > 0.f) {
                    const int ex = (int)((__float_as_uint(bound) >> 23) & 255u) - 127;      // bound < 2^(ex + 1)
                    e_new = min(max(30 - ex, -120), 120);
                }
                if (have_scale && run_max > 0.f && e_new < e2) {
                    // the scale shrinks by 2^k: what has been added so far is divided by it (rounded), all threads, then everyone goes on
                    const int ksh = min(e2 - e_new, 31);
                    const long long half_ulp = 1ll << (ksh - 1);
                    unsigned long long *const flat = &L.acc[0][0];
                    for (int i = tid; i < NC * kScatPitch * kScatStride; i += kScatThreads) {
                        const long long t = (long long)flat[i];
                        const int im = (int)(unsigned)(t & 0xffffffffll);
                        const int re = (int)((t - (long long)im) >> 32);
                        const long long im2 = ((long long)im + half_ulp) >> ksh, re2 = ((long long)re + half_ulp) >> ksh;
                        flat[i] = (unsigned long long)((re2 << 32) + im2);
                    }
                    __syncthreads();
                }
                // (the bound never falls from round to round -- both maxima and their sum only grow -- so the scale never has to grow; before the
                //  first non-zero sample every sum is zero and any scale will do)
                if (!(have_scale && run_max > 0.f) || e_new < e2) e2 = e_new;
                have_scale = true;
                run_max = new_max;
                S = __uint_as_float((unsigned)(e2 + 127) << 23);
                invS = __uint_as_float((unsigned)(127 - e2) << 23);
            }
            // ---- the NEXT slice's samples on their way into L2 (its run table has arrived in registers: every thread holds two entries and
            // touches the 128-byte lines of their segments); that slice's front then waits for L2, not for HBM ----
            // ---- the NEXT round's member table on its way (this slice's next round, or the first round of the workgroup's next slice) ----
            if (r0 + R < iters) copy_groups(z, rbase, total, r0 + R);
            else if (more) copy_groups(z + 1, rbase_next, hdr_next.w, 0);
            // ... and the next slice's run table (asked for HERE, behind the front's waits: the counters are in order, a request in front of
            // them would be waited for with the member table)
            if (more && r0 == 0) fetch_table(z + 1, hdr_next);
            // ---- back: scatter from registers ----
#pragma unroll
            for (int q = 0; q < R; ++q) {
                if (r0 + q >= iters) break;
                if (meta[q] >> 31) {
                    const float2 ecs = *reinterpret_cast<const float2 *>(&L.run[meta[q] & 511u].z);
                    const float uf = (float)((meta[q] >> 9) & 1023u);
                    const float cs_c = ecs.x, cs_s = ecs.y;
                    const float kx = uf * cs_c, ky = uf * cs_s;                            // src/tron.cu:514-515
                    // First column / row X with |k - X| < W: floor(k - W) + 1, the footprint = that and the next three.  (k - W is rounded
                    // at the magnitude of k: where that swallows a distance just below W -- k = -63.0000038, W = 2: k - W rounds to -65 --
                    // the column below is still inside by the reference's own test, fabsf(k - X) < W, whose difference is exact.)
                    float ixf = floorf(kx - W) + 1.0f, iyf = floorf(ky - W) + 1.0f;
                    if (kx - (ixf - 1.0f) < W) ixf -= 1.0f;
                    if (ky - (iyf - 1.0f) < W) iyf -= 1.0f;
                    // table positions of the distance from the first column (exact: the scale is a power of two), and of the third
                    const v2f t0 = (v2f){kx - ixf, ky - iyf} * (v2f){lscale, lscale};
                    const v2f t2 = t0 - (v2f){two_s, two_s};
                    const v2f tt0 = {__builtin_truncf(t0.x), __builtin_truncf(t0.y)}, tt2 = {__builtin_truncf(t2.x), __builtin_truncf(t2.y)};
                    const v2f f0 = t0 - tt0, f2 = t2 - tt2;
                    const slds_f2p lx0 = lutA + (int)tt0.x, ly0 = lutA + (int)tt0.y, lx2 = lutB + (int)tt2.x, ly2 = lutB + (int)tt2.y;
                    auto pair = [&](const slds_f2p qq, const float f) -> v2f {
                        const v2f a0 = qq[0], a1 = qq[kScatLutS + 2], a2 = qq[2 * (kScatLutS + 2)];
                        const v2f fv = {f, f};
                        return __builtin_elementwise_fma(fv, __builtin_elementwise_fma(fv, a2, a1), a0);
                    };
                    const v2f wxa = pair(lx0, f0.x), wxb = pair(lx2, f2.x), wya = pair(ly0, f0.y), wyb = pair(ly2, f2.y);
                    const float wx[4] = {wxa.x, wxa.y, wxb.x, wxb.y}, wy[4] = {wya.x, wya.y, wyb.x, wyb.y};
                    int bx = (int)(ixf - fx0), by = (int)(iyf - fy0);
                    bx = min(max(bx, 0), kScatPitch - 4);                                  // (never binds: the segments are clipped to tile + W)
                    by = min(max(by, 0), kScatPitch - 4);
                    const int base = by * kScatStride + bx;
                    // The band of src/tron.cu:498-502, 512, 521 -- ceil(R - W) <= u <= floor(R + W), R = hypotf(X, Y) -- as
                    // (u - W)^2 <= X^2 + Y^2 <= (u + W)^2: the same set for integer u (checked against the host's band table point by point before
                    // a plan takes this kernel, scatter_band_is_analytic), all three exact in fp32.  The four points next to the sample lie
                    // within sqrt(2) < W = 2 of it and pass always.
                    // As ONE comparison: |X^2 + Y^2 - C| <= D with C = (A + B) / 2, D = (B - A) / 2, A = (u - W)^2 (0 below W), B = (u + W)^2 --
                    // integers or multiples of 1/2 below 2^23: every term and difference exact.
                    const float um = fmaxf(uf - W, 0.0f), up = uf + W;
                    const float bandA = um * um, bandB = up * up;
                    const float bandC = 0.5f * (bandA + bandB), bandD = 0.5f * (bandB - bandA);
                    // Two points per instruction (round 6; a compare + select per point before: 4 instructions and 3 hazard cycles per outer point
                    // where this is 2.5): s = X^2 + Y^2 - C for a pair of columns, mask = clamp(D^2 + qd - s^2, 0, 1) as ONE packed fma with the
                    // clamp modifier.  W = 2 and u >= 3 (this kernel's plans): C = u^2 + 4, D = 4 u and s are integers, so a point inside its band
                    // has s^2 <= D^2 and the fma's exact value is >= 1, one outside has |s| >= D + 1 and the value is <= -2 D: a single rounding
                    // of the exact value cannot move either across (0, 1), whatever the magnitudes (D^2 + 1 < 2^24 up to nxos = 2048).
                    const float d2p1 = fmaf(bandD, bandD, 1.0f);
                    float xc[4], y2[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float xj = ixf + (float)j, yj = iyf + (float)j;
                        xc[j] = xj * xj - bandC;
                        y2[j] = yj * yj;
                    }
                    auto mask2 = [&](const float xa, const float xb, const float yy, const float w) -> v2f {      // (in band ? w : 0) for two columns of one row
                        const v2f sv = (v2f){xa, xb} + (v2f){yy, yy};
                        v2f m;
                        asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(m) : "v"(sv), "v"((v2f){d2p1, d2p1}));
                        return m * (v2f){w, w};
                    };
                    // density compensation (src/tron.cu:412: |ro - nro/2| = u, or u's sample) and the fixed-point scale, once per sample
                    const float sdc = fmaf(dcf_a, RS ? arc_sample_of(uf, rs_nro, rs_inv) : uf, dcf_b) * S;
                    v2f a[4][NC];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const v2f ds = dreg[q][c] * (v2f){sdc, sdc};
#pragma unroll
                        for (int j = 0; j < 4; ++j) a[j][c] = ds * (v2f){wx[j], wx[j]};
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        // row i's four weights: the inner columns of the inner rows lie within sqrt(2) < W of the sample and pass always
                        float wrow[4];
                        const v2f outer = mask2(xc[0], xc[3], y2[i], wy[i]);
                        wrow[0] = outer.x; wrow[3] = outer.y;
                        if (i == 1 || i == 2) {
                            wrow[1] = wy[i]; wrow[2] = wy[i];
                        } else {
                            const v2f inner = mask2(xc[1], xc[2], y2[i], wy[i]);
                            wrow[1] = inner.x; wrow[2] = inner.y;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float wyb2 = wrow[j];                                    // zero outside the point's band (adding 0 costs less than a branch)
#pragma unroll
                            for (int c = 0; c < NC; ++c)
                                if (c < ncb) {
                                    const v2f v = a[j][c] * (v2f){wyb2, wyb2};             // src/tron.cu:516, 519
                                    const int re = cvt_rpi(v.x), im = cvt_rpi(v.y);
                                    // (re << 32) + im as a signed 64-bit number: low word im, high word re - (im < 0)
                                    const unsigned lo = (unsigned)im, hi = (unsigned)re + (unsigned)(im >> 31);
                                    __hip_atomic_fetch_add(&L.acc[c][base + i * kScatStride + j], ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                }
                        }
                    }
                }
            }
            SPROF_MARK(6);                                      // scatter
        }
        // an empty run (a rim tile no spoke of this window crosses) has no round to issue the next slice's requests from: without
        // them the next slice of this workgroup would look its members up in a stale table (tests: few spokes per window, 64 slices)
        if (iters <= 0 && more) {
            copy_groups(z + 1, rbase_next, hdr_next.w, 0);
            fetch_table(z + 1, hdr_next);
        }
        __syncthreads();                                        // every sample of the slice has been added
        SPROF_MARK(7);

        // ---- store: thread = 2x2 points; the points read and the thread's share of the halo ring are zeroed for the next slice ----
        {
            const float os = invS * p.scale;                    // src/tron.cu:532-534
            unsigned char *zbase = reinterpret_cast<unsigned char *>(p.udata + (size_t)z * p.out_z + (size_t)c0 * p.out_c);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c < ncb) {
#pragma unroll
                    for (int kb = 0; kb < G::kBlocksPerThread; ++kb) {
                    int mx, my;
                    block_origin(kb, mx, my);
#pragma unroll
                    for (int qy = 0; qy < 2; ++qy) {
                        unsigned long long *const s2 = &L.acc[c][(my + qy + kScatHalo) * kScatStride + mx + kScatHalo];
                        float f[4];
#pragma unroll
                        for (int qx = 0; qx < 2; ++qx) {
                            const long long t = (long long)s2[qx];
                            s2[qx] = 0ull;
                            const int im = (int)(unsigned)(t & 0xffffffffll);
                            const int re = (int)((t - (long long)im) >> 32);
                            f[2 * qx] = (float)re * os;
                            f[2 * qx + 1] = (float)im * os;
                        }
                        float4 v = make_float4(f[0], f[1], f[2], f[3]);
                        float4 *const o = reinterpret_cast<float4 *>(zbase + (size_t)c * p.out_c * 8 + out_offset(mx, my, qy));
                        if (p.arc_accumulate) {                             // a later pass over more than kArcMaxNpe spokes per window
                            const float4 old = *o;
                            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
                        }
                        *o = v;
                    }
                    }
                }
                for (int k = tid; k < kRing; k += kScatThreads) L.acc[c][halo_index(k)] = 0ull;
            }
            if (tid < 2) L.dmax_bits[tid] = 0u;
            if (more) table_to_lds(hdr_next.x);                 // (every wave is past the scatter: the old table is done with)
        }
        SPROF_MARK(8);                                          // store, next slice's table
    }
    SPROF_FLUSH;
}

#ifdef TRON_PHASE_CLOCK
extern "C" __attribute__((visibility("default"))) int tron_debug_scat_profile(unsigned long long *out, int n)   // reads and clears the phase clock
{
    static unsigned long long h[kScatProfCopies * kScatProfSlots];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_scat_prof), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < n && i < kScatProfSlots; ++i) {
        out[i] = 0;
        for (int c = 0; c < kScatProfCopies; ++c) out[i] += h[c * kScatProfSlots + i];
    }
    for (size_t i = 0; i < sizeof(h) / sizeof(h[0]); ++i) h[i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_scat_prof), h, sizeof(h)) != hipSuccess;
}
#endif

template <int NC, bool HALF, bool RS, int TILE>
static hipError_t launch_scatter_tile(const GridParams &p, int first_plain, hipStream_t s)
{
    using G = ScatGeom<TILE>;
    const int tpr = p.nxos / TILE;
    GridParams q = p;
    q.tiles_per_row = tpr;
    q.ntiles = tpr * tpr;
    q.tile_order = p.tile_order + first_plain;
    q.arc_zper = p.arc_zper > 0 ? p.arc_zper : 1;
    const int ngroups = (p.nslices + q.arc_zper - 1) / q.arc_zper;
    const int chunks = (p.nchan - p.coil0 + NC - 1) / NC;
    dim3 grid((unsigned)((size_t)q.ntiles * ngroups), (unsigned)chunks);
    const size_t lds = sizeof(ScatLds<NC, TILE>);
    if (lds > 64 * 1024) {
        const hipError_t once = allow_dynamic_lds(reinterpret_cast<const void *>(grid_scatter_kernel<NC, HALF, RS, TILE>), (int)lds);
        if (once != hipSuccess) return once;
    }
    hipLaunchKernelGGL((grid_scatter_kernel<NC, HALF, RS, TILE>), grid, dim3(G::kThreads), lds, s, q);
    return hipGetLastError();
}

template <int NC, bool HALF, bool RS>
static hipError_t launch_scatter_rs(const GridParams &p, int first_plain, hipStream_t s)
{
    return p.scat_tile == 64 ? launch_scatter_tile<NC, HALF, RS, 64>(p, first_plain, s) : launch_scatter_tile<NC, HALF, RS, 32>(p, first_plain, s);
}

template <int NC>
static hipError_t launch_scatter_nc(const GridParams &p, int half_in, int first_plain, hipStream_t s)
{
    const bool rs = p.nro != p.nxos;
    if (half_in) return rs ? launch_scatter_rs<NC, true, true>(p, first_plain, s) : launch_scatter_rs<NC, true, false>(p, first_plain, s);
    return rs ? launch_scatter_rs<NC, false, true>(p, first_plain, s) : launch_scatter_rs<NC, false, false>(p, first_plain, s);
}

// The plans the arc kernel takes (grid_arc_supported) with one, two or an odd number of channels and a window of four points per axis; the caller has
// checked scatter_band_is_analytic for the plan's grid and width.
bool grid_scatter_supported(int nchan, int nxos, int nro, int npe, float W, int half_in)
{
    (void)half_in;                                              // fp32 and complex-half k-space alike: the samples are read by plain loads
    // W = 2 (the reference's default, src/tron.cu:69): four columns per footprint, of which the inner two lie within 1 of the sample, so that the
    // four inner points (within sqrt(2) < W) are always inside their band; the band test's squares are exact; the pair table has 64 pieces
    // per unit.  Other widths keep the arc kernel.
    // Channels: one, two, or any ODD count (one coil x nt repetitions; the reference itself takes one or an even number of coils, and the arc
    // kernel copies coil pairs): odd counts run one channel per pass, blockIdx.y = channel.
    return (nchan == 1 || nchan == 2 || (nchan & 1)) && W == 2.0f && kb_pair_lut_scale(W, kArcLutEntries) == kScatLutS
           && grid_arc_supported(nchan == 2 ? 2 : 1, nxos, nro, npe, W, 0) && (long long)nro * npe * nchan < (1ll << 28);
}

// p.tile_order[first_plain ...] must list the 32x32 tiles; run tables from arc_prep_kernel with ONE batch per run (ArcPrepParams::nrec >= 32767).
hipError_t launch_grid_scatter(const GridParams &p, int half_in, int first_plain, hipStream_t s)
{
    const int nc = p.nchan - p.coil0;
    if (p.out_p != 1 || p.inner_r0 < 3 || !p.arc_hdr || !p.arc_ent || !p.arc_off || !p.arc_rec || !p.arc_rbase || (p.scat_tile != 32 && p.scat_tile != 64) || (p.nxos / 2) % p.scat_tile != 0 || !p.kb_lut || p.lut_entries > kArcLutEntries || p.npe > kArcMaxNpe || !(p.scat_wsum > 0.f) || !(p.scat_wmax > 0.f) || (int)p.lut_scale != kScatLutS
        || !grid_scatter_supported(nc, p.nxos, p.nro, p.npe, p.W, half_in) || (reinterpret_cast<uintptr_t>(p.nudata) & (half_in ? 3 : 7)) != 0
        || (nc == 2 && (reinterpret_cast<uintptr_t>(p.nudata) & (half_in ? 7 : 15)) != 0))
        return hipErrorInvalidValue;
    return nc == 2 ? launch_scatter_nc<2>(p, half_in, first_plain, s) : launch_scatter_nc<1>(p, half_in, first_plain, s);
}

__global__ void warm_grid_scatter_tu() {}

hipError_t warm_grid_scatter()   // see warm_kernels() in tron_kernels.hip
{
    hipLaunchKernelGGL(warm_grid_scatter_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
