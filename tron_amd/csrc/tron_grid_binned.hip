// Gridding kernel of the TRON_KB_FAST path: sample records are BINNED by Cartesian cell inside the
// workgroup, so every point walks exactly the samples whose Kaiser-Bessel footprint covers it.
//
// Same arithmetic per (sample, point) pair as grid_tile_kernel / the reference's gridradial2d
// (src/tron.cu:465-536: weights :516, band :498-502,:512, r = 0 twice :512/:521, scale :532) but
// the pairs are found by a counting sort instead of a per-spoke search, and summed cell by cell
// rather than spoke by spoke -- so results agree with the reference to fp32 summation-order noise
// (~1e-7), not bit for bit.  That is why this kernel serves TRON_KB_FAST only; TRON_KB_EXACT keeps
// the order-preserving gather.
//
// One workgroup = 4 waves = one 32x32 tile; each thread owns 2x2 points.  Once per tile the spokes are clipped against
// tile + halo (thread = spoke) and compacted in acquisition order.  Then, per batch of <= NREC records:
//   plan    (one batch ahead) the longest run of spokes that fits, and which spoke holds each record: lanes along the
//           spokes, one LDS read + ballot + v_readlane per wave, no dependent LDS chains;
//   stage   lanes run along the spoke: 2x(2CW) weights once per sample (packed polynomials), the samples copied
//           global -> LDS by LDS-DMA (fp32, even coil counts) or through registers; the sample's base cell gets a packed
//           per-wave counter bumped with ONE integer LDS atomic (ds_add_rtn_u32: 6.6 cycles per wave instruction on
//           MI355X, vs 194 for ds_add_f32), which returns the sample's rank in its (wave, cell) bucket;
//   scan    exclusive prefix sum over the (32+2CW-1)^2 cells -> CSR row starts;
//   place   8-byte entries (pre-shifted LDS offsets of the record's weights and samples, its cell, |r|) are written to
//           their sorted slots: cell start + earlier waves' counts + rank (deterministic: each wave issues its atomics
//           in program order);
//   apply   a thread reads, for the 2CW+1 cell rows its 2x2 points can see, ONE concatenated entry range, and
//           accumulates weight pair x weight pair x sample for all coils in registers.
// Centre relief (GridParams::inner_r0): the samples next to the k-space centre, which every spoke passes, are gridded by
// workgroups of their own (the origin-centred inner tile, dealt over spoke ranges) and added onto the four centre tiles.
// No floating-point atomics anywhere; the output is written once, coil-planar, in FFT-native order.  DESIGN.md 4.1 has
// the measurements (phase clock, counters: the kernel is VALU-issue-bound).
#include <stdlib.h>

#include "tron_device.h"
#include "tron_grid_store.h"

namespace tron {

constexpr int kBinTile = 32;
constexpr int kBinThreads = 256;
constexpr int kBinMaxSpokes = 512;   // accepted spokes kept per clip round

template <int CPB, int CW>
struct BinCfg {
    static constexpr int NW = 2 * CW;
    static constexpr int NWP = NW + 2;                       // weights padded with a zero each side
    static constexpr int NCELL = kBinTile + 2 * CW - 1;      // base cells per dimension that can touch the tile
    static constexpr int CELLW = NCELL + 1;                  // + sentinel column
    static constexpr int NCELLS = NCELL * CELLW;
    static constexpr int CPT = (NCELLS + kBinThreads - 1) / kBinThreads;   // cells per thread in the scan
    // records per batch.  8 coils need 167 VGPRs = 3 waves per SIMD, so three 48 KiB workgroups per CU and the largest
    // batches that allows; with fewer coils the kernel fits 4 waves per SIMD and a FOURTH workgroup per CU (<= 40 KiB)
    // beats larger batches (measured on one box, gridding us per coil-slice, 36 KiB of records -> this:
    // 4 coils 3.51 -> 3.13, 2 coils 5.96 -> 5.45, 1 coil 11.1 -> 9.9; 8 coils at 24/28 KiB: 2.28 -> 2.45)
    static constexpr int REC_BYTES = 2 * NWP * 4 + CPB * 8 + 8 + 4 + 1;
    static constexpr int REC_KB = CPB >= 8 ? 36 : (CPB >= 6 ? 30 : (CPB >= 2 ? 24 : 28));   // 6 coils: 256 records too
    static constexpr int NREC_RAW = (REC_KB * 1024) / REC_BYTES;
    static constexpr int NREC = NREC_RAW >= 512 ? 512 : (NREC_RAW / 64) * 64;
    static constexpr int SLOT = 64;                          // longest spoke segment through tile + halo
};

template <int CPB, int CW>
struct BinLds {
    using C = BinCfg<CPB, CW>;
    int sp_pe[kBinMaxSpokes];
    int sp_seg[kBinMaxSpokes];            // rlo (low 16, signed) | len << 16
    int sp_start[kBinMaxSpokes + 1];      // exclusive scan of len
    unsigned hist[C::NCELLS];             // 4 x 8-bit per-wave counters per cell
    unsigned short start[C::NCELLS + 1];
    // per sorted slot, everything the apply loop needs besides the weights, pre-shifted into LDS byte offsets:
    //   .x = record's weight-row offset (id * 4 NWP) | 4 fxrel << 15 | 4 fyrel << 23 | (r == 0) << 31
    //   .y = |r| | record's sample offset (id * 16, or * 8 for one coil) << 16
    uint2 sorted[C::NREC];
    unsigned key[C::NREC];
    unsigned char rank[C::NREC];
    int wcnt[8];
    int clipw[2 * 4 * (kBinMaxSpokes / kBinThreads)];   // per (chunk, wave): accepted spokes, their records
    float wx[C::NREC * C::NWP];
    float wy[C::NREC * C::NWP];
    float2 d[C::NREC * CPB];
};

// How the k-space samples of a batch reach LDS:
//   kInRegs32 / kInRegs16  loaded into registers one batch ahead (fp32 / complex-half), scaled by the density
//                          compensation and written to LDS by the staging pass (any coil count, slice groups);
//   kInLdsDma              fp32, even coil counts: `global_load_lds_dwordx4` copies each sample's coil pairs straight
//                          into the [pair][record] image (lane-linear: record = lane), no registers, no ds_write pass;
//                          the density compensation is folded into the x weights instead.  The copies are issued when
//                          the batch opens and must have landed before its apply loop: they fly under staging, scan
//                          and placement, which is why the barriers in between are raw (LDS counter only) -- a
//                          __syncthreads() there would wait for them.
enum { kInRegs32 = 0, kInRegs16 = 1, kInLdsDma = 2 };

// Phase clock of tools/gridprof.py (-DTRON_PHASE_CLOCK builds only): shader-clock cycles per wave and phase, summed
// over all waves of all launches since the last read; production builds carry none of it.
#ifdef TRON_PHASE_CLOCK
constexpr int kProfSlots = 16, kProfCopies = 4096;               // copies: same-address atomics would dominate the kernel
__device__ unsigned long long g_bin_prof[kProfCopies * kProfSlots];
#define PROF_DECL unsigned prof_acc[kProfSlots] = {}; unsigned long long prof_t = __builtin_readcyclecounter()
#define PROF_MARK(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[i] += (unsigned)(t_ - prof_t); prof_t = t_; } while (0)
#define PROF_FLUSH do { if (lane == 0) { for (int i_ = 0; i_ < kProfSlots; ++i_) if (prof_acc[i_]) atomicAdd(&g_bin_prof[((blockIdx.x * 4 + wave) % kProfCopies) * kProfSlots + i_], (unsigned long long)prof_acc[i_]); } } while (0)
#else
#define PROF_DECL
#define PROF_MARK(i)
#define PROF_FLUSH
#endif

template <int CPB, int CW, int IN>
__global__ void __launch_bounds__(kBinThreads, 3)
grid_binned_kernel(const GridParams p)
{
    constexpr bool HALF = IN == kInRegs16;
    constexpr bool DMA = IN == kInLdsDma;
    static_assert(!DMA || CPB % 2 == 0, "LDS-DMA staging copies coil pairs");
    using C = BinCfg<CPB, CW>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    BinLds<CPB, CW> &L = *reinterpret_cast<BinLds<CPB, CW> *>(lds_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int z = blockIdx.x % p.nslices;
    // tile id | part << 16 | parts << 20 | slot << 24: a heavy tile (the k-space centre: every spoke crosses it) may be
    // dealt to `parts` workgroups over disjoint spoke ranges; grid_reduce_parts_kernel adds their tiles in part order
    const int entry = p.tile_order[blockIdx.x / p.nslices];
    const int tile = entry & 0xffff;
    const int part = (entry >> 16) & 15, nparts = max((entry >> 20) & 15, 1), slot = (entry >> 24) & 255;
    const int pe_lo = (int)(((long long)part * p.npe) / nparts), pe_hi = (int)(((long long)(part + 1) * p.npe) / nparts);
    // vslices > 1 (linear angles: every slice has the SAME trajectory, src/tron.cu:509 depends on pe only): `vslices`
    // consecutive slices ride in the coil dimension -- register channel c = slice (c / nchan) of the group, coil c % nchan
    // -- so clipping, weights and the sort are paid once per group instead of once per slice
    const int vs = DMA ? 1 : (p.vslices > 1 ? p.vslices : 1);   // (the LDS-DMA instantiation never carries slice groups)
    const int zbase = z * vs;
    const int c0 = vs > 1 ? 0 : p.coil0 + blockIdx.y * CPB;
    const int ncb = vs > 1 ? min(CPB, (p.nslices_total - zbase) * p.nchan) : min(CPB, p.nchan - c0);
    const int n = p.nxos;
    const int h = n / 2;
    const int rmax = n / 2 - 1;
    // inner_r0 > 0 (centre relief): the radial sampling density ~ npe / (pi r) makes the point blocks next to the k-space
    // centre the serial chain of the four tiles that meet there (their corner blocks see every spoke, a typical block
    // 1 in 80), so samples |r| < inner_r0 are taken OUT of those tiles and gridded by workgroups of their own: the
    // "inner tile" (id ntiles) is the 32x32 square centred on the origin, dealt over spoke ranges like a split tile, and
    // grid_reduce_parts_kernel adds its parts onto the centre tiles' corners.
    const bool inner = p.inner_r0 > 0 && tile == p.ntiles;
    if (tile < 0 || (tile >= p.ntiles && !inner)) {
        if (threadIdx.x == 0) atomicOr(p.errflag, 128u);
        return;
    }

    const int x0 = inner ? -kBinTile / 2 : (tile % p.tiles_per_row) * kBinTile - h;     // tile origin, centred coordinates
    const int y0 = inner ? -kBinTile / 2 : (tile / p.tiles_per_row) * kBinTile - h;
    const bool outer = p.inner_r0 > 0 && !inner && (x0 == 0 || x0 == -kBinTile) && (y0 == 0 || y0 == -kBinTile);
    if (p.skip_outside) {
        // nearest point of the tile to the k-space centre; beyond rmax + W every band is empty (src/tron.cu:498-502,512)
        const int ax = max(max(x0, -(x0 + kBinTile - 1)), 0), ay = max(max(y0, -(y0 + kBinTile - 1)), 0);
        const float lim = (float)rmax + p.W + 1.0f;
        if ((float)(ax * ax + ay * ay) > lim * lim) return;
    }
    const int mx = 2 * (lane & 15);                             // this thread's 2x2 points, tile-relative
    const int my = 8 * wave + 2 * (lane >> 4);
    const int X0 = x0 + mx, Y0 = y0 + my;
    int Rlo[4], Rhi[4];                                         // radial band per point, src/tron.cu:498-502
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int X = X0 + (q & 1), Y = Y0 + (q >> 1);
        Rlo[q] = 1 << 20; Rhi[q] = 1 << 20;                    // empty band: (unsigned)(ar - Rlo) > 0 for every sample
        if (X + h < n && Y + h < n) {
            const uint32_t bnd = p.band[(size_t)(Y + h) * n + (X + h)];
            const int lo = (int)(bnd & 0xffffu), hi = (int)(bnd >> 16);
            if (lo <= hi) { Rlo[q] = lo; Rhi[q] = hi; }         // corners beyond the last sample have lo > hi: empty
        }
    }
    // only tiles touching the k-space centre can meet r = 0 (counted twice by the reference, :512/:521)
    const bool has_centre = x0 <= CW && x0 + kBinTile > -CW && y0 <= CW && y0 + kBinTile > -CW;

    KbCoef kb;
    kb.W = p.W; kb.invW = 1.0f / p.W; kb.beta = p.beta;
#pragma unroll
    for (int t = 0; t < kKbPolyTerms; ++t) kb.poly[t] = p.kb_poly[t];

    float2 acc[4][CPB];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int c = 0; c < CPB; ++c) acc[q][c] = make_float2(0.f, 0.f);

    const unsigned char *in_bytes = reinterpret_cast<const unsigned char *>(p.nudata)
        + (size_t)zbase * (size_t)p.in_slice_stride * (HALF ? sizeof(__half2) : sizeof(float2));
    const float2 *trig = p.trig + (size_t)z * p.trig_slice_stride;

    const float eps = 0.01f;
    const float bx_lo = (float)x0 - p.W - eps, bx_hi = (float)(x0 + kBinTile - 1) + p.W + eps;
    const float by_lo = (float)y0 - p.W - eps, by_hi = (float)(y0 + kBinTile - 1) + p.W + eps;
    const int cx0 = x0 - CW, cy0 = y0 - CW;                     // base cell (0,0) of the histogram

    PROF_DECL;
    for (int round0 = pe_lo; round0 < pe_hi; round0 += kBinMaxSpokes) {
        // ---- clip: one thread per spoke (kBinMaxSpokes / 256 spokes each), accepted spokes compacted in acquisition
        //      order together with the exclusive scan of their segment lengths: one exchange of per-wave totals ------
        constexpr int NCH = kBinMaxSpokes / kBinThreads;
        static_assert(kBinMaxSpokes % kBinThreads == 0, "whole chunks of spokes");
        bool c_acc[NCH];
        int c_rlo[NCH], c_len[NCH], c_incl[NCH];
        unsigned long long c_m[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int pe = round0 + k * kBinThreads + tid;
            c_acc[k] = false;
            c_rlo[k] = 0;
            c_len[k] = 0;
            if (pe < pe_hi) {
                const float2 cs = trig[pe];
                const float ic = safe_rcp(cs.x), is = safe_rcp(cs.y);
                const float xa = bx_lo * ic, xb = bx_hi * ic;
                const float ya = by_lo * is, yb = by_hi * is;
                const float lo = fmaxf(fmaxf(fminf(xa, xb), fminf(ya, yb)), -(float)rmax);
                const float hi = fminf(fminf(fmaxf(xa, xb), fmaxf(ya, yb)), (float)rmax);
                if (lo <= hi) {
                    int rlo = (int)ceilf(lo);
                    int rhi = (int)floorf(hi);
                    if (inner) {                                // only the samples the centre tiles leave out
                        rlo = max(rlo, 1 - p.inner_r0);
                        rhi = min(rhi, p.inner_r0 - 1);
                    } else if (outer) {
                        // a quadrant tile holds one side of a spoke plus at most W sqrt(2) < inner_r0 beyond the origin
                        if (rhi >= p.inner_r0) {
                            if (rlo <= -p.inner_r0) atomicOr(p.errflag, 16u);
                            rlo = max(rlo, p.inner_r0);
                        } else if (rlo <= -p.inner_r0) {
                            rhi = min(rhi, -p.inner_r0);
                        } else {
                            rhi = rlo - 1;
                        }
                    }
                    if (rhi - rlo + 1 > C::SLOT) {              // cannot happen for a 32x32 tile with W <= 4
                        atomicOr(p.errflag, 2u);
                        rhi = rlo + C::SLOT - 1;
                    }
                    c_rlo[k] = rlo;
                    c_len[k] = max(rhi - rlo + 1, 0);
                    c_acc[k] = c_len[k] > 0;
                }
            }
            c_m[k] = __ballot(c_acc[k]);
            int v = c_len[k];                                   // inclusive scan of the lengths within the wave
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(v, o);
                if (lane >= o) v += t;
            }
            c_incl[k] = v;
            if (lane == 63) {
                L.clipw[(k * 4 + wave) * 2] = __popcll(c_m[k]);
                L.clipw[(k * 4 + wave) * 2 + 1] = v;
            }
        }
        if (tid == 0) L.sp_start[0] = 0;
        __syncthreads();
        int nacc = 0;                                           // accepted spokes of this round (uniform)
        {
            int bcnt[NCH], blen[NCH];
            int run_c = 0, run_l = 0;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w == wave) { bcnt[k] = run_c; blen[k] = run_l; }
                    run_c += L.clipw[(k * 4 + w) * 2];
                    run_l += L.clipw[(k * 4 + w) * 2 + 1];
                }
            nacc = run_c;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (c_acc[k]) {
                    const int slot = bcnt[k] + __popcll(c_m[k] & ((1ull << lane) - 1ull));
                    L.sp_pe[slot] = round0 + k * kBinThreads + tid;
                    L.sp_seg[slot] = (c_rlo[k] & 0xffff) | (c_len[k] << 16);
                    L.sp_start[slot + 1] = blen[k] + c_incl[k];
                }
        }
        __syncthreads();

        // ---- batches: the longest run of spokes whose records fit in NREC; the k-space samples of batch
        //      b+1 are fetched into registers while batch b is scanned, placed and applied ----------
        constexpr int RPT = (C::NREC + kBinThreads - 1) / kBinThreads;      // records per thread per batch
        int pf_pe[RPT], pf_r[RPT];
        float2 pf_cs[RPT];                                      // the spoke's (cos, sin): fetched with the batch, not in the staging pass
        float2 pf_d[DMA ? 1 : RPT][DMA ? 1 : CPB];
        // The next batch = the longest run of accepted spokes s0 .. s1-1 whose records fit in NREC, and for each of this
        // thread's records the spoke that holds it.  Every wave does this on its own, lanes along the spokes: ONE LDS read
        // fetches 64 segment ends, a ballot finds how many fit, and the ends are handed round with v_readlane -- no serial
        // scan for s1, no binary search per record (each step of either was a dependent LDS round trip).
        int pf_cnt = 0, pf_base = 0;                            // records of the prefetched batch, records before it
        auto prefetch = [&](int s0, int base) -> int {
            int spk[RPT];
#pragma unroll
            for (int j = 0; j < RPT; ++j) spk[j] = s0;
            int s1 = s0, cnt = 0;
            for (int sb = s0;; sb += 64) {
                const int idx = sb + 1 + lane;                  // e = where spoke idx starts = where idx - 1 ends
                const int e = L.sp_start[min(idx, nacc)] - base;
                const unsigned long long m = __ballot(idx <= nacc && e <= C::NREC);   // monotone along the lanes; spoke s0 always fits (SLOT <= NREC)
                const int nfit = m == ~0ull ? 64 : __builtin_ctzll(~m);
                for (int k = 0; k < nfit; ++k) {
                    const int ek = __builtin_amdgcn_readlane(e, k);
#pragma unroll
                    for (int j = 0; j < RPT; ++j)
                        if (tid + j * kBinThreads >= ek) spk[j] = sb + 1 + k;
                }
                if (nfit > 0) {
                    cnt = __builtin_amdgcn_readlane(e, nfit - 1);
                    s1 = sb + nfit;
                }
                if (nfit < 64) break;
            }
            pf_base = base;
            pf_cnt = cnt;
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int rec = tid + j * kBinThreads;
                pf_pe[j] = -1;
                if (rec < cnt) {
                    const int lo = spk[j];                      // sp_start[lo] <= base + rec < sp_start[lo + 1]
                    const int target = base + rec;
                    const int pe = L.sp_pe[lo];
                    const int r = (int)(short)(L.sp_seg[lo] & 0xffff) + (target - L.sp_start[lo]);
                    pf_pe[j] = pe;
                    pf_r[j] = r;
                    pf_cs[j] = trig[pe];
                    if constexpr (!DMA) {
                        const int ro = (p.nro == n ? r : (r * p.nro) / n) + p.nro / 2;   // src/tron.cu:517,519 (truncating division)
                        const size_t sbase = ((size_t)p.nro * pe + ro) * p.nchan + c0;
                        if (vs > 1) {
                            // channel c = (slice c / nchan of the group, coil c % nchan): one load per channel, each coalesced along the spoke
    #pragma unroll
                            for (int c = 0; c < CPB; ++c)
                                pf_d[j][c] = c < ncb ? load_sample<HALF>(in_bytes, (size_t)(c / p.nchan) * (size_t)p.in_slice_stride + sbase + (c % p.nchan))
                                                     : make_float2(0.f, 0.f);
                        } else if (!HALF && CPB % 2 == 0 && (ncb & 1) == 0 && (p.nchan & 1) == 0 && (c0 & 1) == 0) {
                            // the coils of one sample are contiguous: 16-byte loads (coils beyond ncb are zero padding)
                            const float4 *src4 = reinterpret_cast<const float4 *>(reinterpret_cast<const float2 *>(in_bytes) + sbase);
    #pragma unroll
                            for (int c = 0; c < CPB / 2; ++c) {
                                const float4 v = 2 * c < ncb ? src4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
                                pf_d[j][2 * c] = make_float2(v.x, v.y);
                                pf_d[j][2 * c + 1] = make_float2(v.z, v.w);
                            }
                        } else if (HALF && CPB % 4 == 0 && ncb == CPB && (p.nchan & 3) == 0 && (c0 & 3) == 0) {
                            // complex-half storage: four coils of one sample per 16-byte load
                            const uint4 *src4 = reinterpret_cast<const uint4 *>(reinterpret_cast<const __half2 *>(in_bytes) + sbase);
    #pragma unroll
                            for (int c = 0; c < CPB / 4; ++c) {
                                const uint4 v = src4[c];
                                const unsigned w[4] = {v.x, v.y, v.z, v.w};
    #pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    __half2 hh;
                                    __builtin_memcpy(&hh, &w[k], 4);
                                    pf_d[j][4 * c + k] = __half22float2(hh);
                                }
                            }
                        } else {
    #pragma unroll
                            for (int c = 0; c < CPB; ++c)
                                pf_d[j][c] = c < ncb ? load_sample<HALF>(in_bytes, sbase + c) : make_float2(0.f, 0.f);
                        }
                    }
                }
            }
            return s1;
        };

        int sp0 = 0, sp1 = 0;
        PROF_MARK(0);                                           // setup + clip + segment scan
        if (nacc > 0) sp1 = prefetch(0, 0);
        PROF_MARK(1);                                           // first prefetch
        while (sp0 < nacc) {
            const int nrec = pf_cnt;

            float pf_kx[RPT], pf_ky[RPT];
#pragma unroll
            for (int j = 0; j < RPT; ++j) {                           // (first use of the prefetched registers: every ordinary load has landed)
                pf_kx[j] = (float)pf_r[j] * pf_cs[j].x;               // src/tron.cu:514-515
                pf_ky[j] = (float)pf_r[j] * pf_cs[j].y;
            }
            if constexpr (DMA) {
                // this batch's samples, global -> LDS image [pair][record]: the previous apply loop (the only reader of
                // L.d) ended at the barrier that closed the last iteration
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    const int rec = tid + j * kBinThreads;
                    if (rec < nrec) {
                        const int r = pf_r[j];
                        const int ro = (p.nro == n ? r : (r * p.nro) / n) + p.nro / 2;   // src/tron.cu:517,519 (truncating division)
                        const float2 *src = reinterpret_cast<const float2 *>(in_bytes) + ((size_t)p.nro * pf_pe[j] + ro) * p.nchan + c0;
                        const unsigned dst0 = lds_addr(reinterpret_cast<const float4 *>(L.d) + j * kBinThreads + wave * 64);
#pragma unroll
                        for (int c = 0; c < CPB / 2; ++c)
                            if (2 * c < ncb) lds_dma16(src + 2 * c, dst0 + (unsigned)(c * C::NREC * sizeof(float4)));
                    }
                }
            }
            for (int c = tid; c < C::NCELLS; c += kBinThreads) L.hist[c] = 0u;
            PROF_MARK(2);                                       // first use of the prefetch, DMA issue, histogram clear
            lds_barrier();
            PROF_MARK(3);                                       // barrier

            // ---- A. stage + count: records are dealt out flat, 64 consecutive records per wave pass ----
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int rec = tid + j * kBinThreads;
                if (rec >= nrec) continue;
                const int r = pf_r[j];
                const float kx = pf_kx[j], ky = pf_ky[j];
                const int fx = (int)floorf(kx), fy = (int)floorf(ky);
                const int bx = fx - CW + 1, by = fy - CW + 1;
                float *wxr = L.wx + rec * C::NWP, *wyr = L.wy + rec * C::NWP;
                wxr[0] = 0.f; wxr[C::NWP - 1] = 0.f;
                wyr[0] = 0.f; wyr[C::NWP - 1] = 0.f;
                float sdc;
                {   // the 2 x NW weights as NW interleaved packed polynomials (x and y share an instruction); each
                    // component is the fmaf chain of kb_weight<TRON_KB_FAST>                    src/tron.cu:516
                    v2f dxy[C::NW], sxy[C::NW], wxy[C::NW];
                    const v2f one = {1.0f, 1.0f};
#pragma unroll
                    for (int i = 0; i < C::NW; ++i) {
                        dxy[i] = (v2f){kx - (float)(bx + i), ky - (float)(by + i)};
                        const v2f rr = dxy[i] * kb.invW;
                        sxy[i] = __builtin_elementwise_fma(-rr, rr, one);
                        wxy[i] = (v2f){kb.poly[kKbPolyTerms - kb_terms(CW)], kb.poly[kKbPolyTerms - kb_terms(CW)]};
                    }
#pragma unroll
                    for (int t = kKbPolyTerms - kb_terms(CW) + 1; t < kKbPolyTerms; ++t) {
                        const v2f cf = {kb.poly[t], kb.poly[t]};
#pragma unroll
                        for (int i = 0; i < C::NW; ++i) wxy[i] = __builtin_elementwise_fma(wxy[i], sxy[i], cf);
                    }
                    const int ro = (p.nro == n ? r : (r * p.nro) / n) + p.nro / 2;
                    sdc = 1.0f;
                    if (p.apply_dcf) sdc = p.dcf_a * fabsf((float)ro - (float)(p.nro / 2)) + p.dcf_b;   // src/tron.cu:412
                    const float sx = DMA ? sdc : 1.0f;               // LDS-DMA: the samples are in LDS unscaled, the x weights carry the DCF
#pragma unroll
                    for (int i = 0; i < C::NW; ++i) {
                        wxr[1 + i] = fabsf(dxy[i].x) < kb.W ? wxy[i].x * sx : 0.0f;
                        wyr[1 + i] = fabsf(dxy[i].y) < kb.W ? wxy[i].y : 0.0f;
                    }
                }
                // samples go to LDS coil-pair-major ([pair][record], 16 bytes each): conflict-free stores
                if constexpr (DMA) {
                } else if (CPB % 2 == 0) {
                    float4 *dst4 = reinterpret_cast<float4 *>(L.d);
#pragma unroll
                    for (int c = 0; c < CPB / 2; ++c) {
                        float4 v;
                        v.x = pf_d[j][2 * c].x * sdc; v.y = pf_d[j][2 * c].y * sdc;        // src/tron.cu:414
                        v.z = pf_d[j][2 * c + 1].x * sdc; v.w = pf_d[j][2 * c + 1].y * sdc;
                        dst4[c * C::NREC + rec] = v;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CPB; ++c)
                        L.d[c * C::NREC + rec] = make_float2(pf_d[j][c].x * sdc, pf_d[j][c].y * sdc);
                }
                const int fxrel = fx - cx0, fyrel = fy - cy0;
                const bool valid = (unsigned)fxrel < (unsigned)C::NCELL && (unsigned)fyrel < (unsigned)C::NCELL;
                const int ar = r < 0 ? -r : r;
                const unsigned key = (unsigned)(fxrel & 63) | ((unsigned)(fyrel & 63) << 6) | ((unsigned)ar << 12)
                                     | ((unsigned)wave << 26) | (valid ? 1u << 28 : 0u) | (r == 0 ? 1u << 29 : 0u);
                L.key[rec] = key;
                if (valid) {
                    const unsigned old = atomicAdd(&L.hist[fyrel * C::CELLW + fxrel], 1u << (8 * wave));
                    const unsigned rk = (old >> (8 * wave)) & 0xffu;
                    if (rk == 0xffu) atomicOr(p.errflag, 4u);
                    L.rank[rec] = (unsigned char)rk;
                }
            }
            PROF_MARK(4);                                       // stage
            // next batch: bounds now, samples in flight while this batch is scanned / placed / applied
            const int nsp0 = sp1;
            int nsp1 = nsp0;
            if (nsp0 < nacc) nsp1 = prefetch(nsp0, pf_base + nrec);
            PROF_MARK(5);                                       // next batch: bounds + prefetch
            lds_barrier();
            PROF_MARK(6);                                       // barrier

            // ---- B. exclusive scan over cells ------------------------------------------------
            {
                int cnt[C::CPT];
                int tsum = 0;
#pragma unroll
                for (int k = 0; k < C::CPT; ++k) {
                    const int c = tid * C::CPT + k;
                    unsigned hv = c < C::NCELLS ? L.hist[c] : 0u;
                    cnt[k] = (int)((hv & 0xff) + ((hv >> 8) & 0xff) + ((hv >> 16) & 0xff) + (hv >> 24));
                    tsum += cnt[k];
                }
                int v = tsum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(v, o);
                    if (lane >= o) v += t;
                }
                if (lane == 63) L.wcnt[wave] = v;
                lds_barrier();
                int run = v - tsum;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (w < wave) run += L.wcnt[w];
#pragma unroll
                for (int k = 0; k < C::CPT; ++k) {
                    const int c = tid * C::CPT + k;
                    if (c <= C::NCELLS) L.start[c] = (unsigned short)run;
                    run += cnt[k];
                }
            }
            PROF_MARK(7);                                       // scan (one barrier inside)
            lds_barrier();
            PROF_MARK(8);                                       // barrier

            // ---- C. place record ids in cell order ---------------------------------------------
            for (int rec = tid; rec < nrec; rec += kBinThreads) {
                const unsigned key = L.key[rec];
                if (key & (1u << 28)) {
                    const int cell = (int)((key >> 6) & 63) * C::CELLW + (int)(key & 63);
                    const unsigned hv = L.hist[cell];
                    const int w = (key >> 26) & 3;
                    const unsigned below = hv & ((1u << (8 * w)) - 1u);
                    const int wb = (int)((below & 0xff) + ((below >> 8) & 0xff) + ((below >> 16) & 0xff));
                    const int pos = L.start[cell] + wb + L.rank[rec];
                    if (pos >= C::NREC) { atomicOr(p.errflag, 8u); continue; }
                    L.sorted[pos] = make_uint2((unsigned)(rec * C::NWP * 4) | ((key & 63u) << 17) | (((key >> 6) & 63u) << 25) | (((key >> 29) & 1u) << 31),
                                               ((key >> 12) & 0x3fffu) | ((unsigned)(rec * (CPB % 2 == 0 ? 16 : 8)) << 16));
                }
            }
            PROF_MARK(9);                                       // place
            if constexpr (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the batch's samples are in LDS
            PROF_MARK(10);                                      // wait for the DMA
            lds_barrier();
            PROF_MARK(11);                                      // barrier

            // ---- D. apply: each thread walks the 2CW+1 cell rows its 2x2 points can see as ONE loop
            //         (row ranges concatenated), so a wave runs max-over-lanes(total), not sum of row maxima
            {
                constexpr int NR = 2 * CW + 1;
                int delta[NR], cum[NR + 1];
                cum[0] = 0;
#pragma unroll
                for (int dy = 0; dy < NR; ++dy) {
                    const int rowbase = (my + dy) * C::CELLW + mx;
                    const int kbeg = L.start[rowbase];
                    const int kend = L.start[rowbase + 2 * CW + 1];
                    delta[dy] = kbeg - cum[dy];                        // k = i + delta[row]
                    cum[dy + 1] = cum[dy] + (kend - kbeg);
                }
                const int total = cum[NR];
                // where visit i sits in the sorted list: i + delta[its row] = i + delta[0] + the gaps of the rows i has passed
                // (cum is non-decreasing, so the tests nest); byte offsets into L.sorted
                int gap8[NR];
#pragma unroll
                for (int dy = 1; dy < NR; ++dy) gap8[dy] = (delta[dy] - delta[dy - 1]) * 8;
                const unsigned sorted0 = lds_addr(L.sorted) + (unsigned)(delta[0] * 8);
                auto entry_of = [&](const int i) {
                    int off = 0;
#pragma unroll
                    for (int dy = 1; dy < NR; ++dy) off += i >= cum[dy] ? gap8[dy] : 0;
                    const __attribute__((address_space(3))) unsigned *q =
                        (const __attribute__((address_space(3))) unsigned *)(size_t)(sorted0 + (unsigned)off + (unsigned)(i * 8));
                    return make_uint2(q[0], q[1]);
                };
                // byte addresses of the padded weights of this thread's first column / row inside a record's weight row:
                // ip = mx + 2CW - fxrel, jp = 2CW - (fyrel - my)
                // (LDS byte addresses as integers: base + thread offset in ONE register, so a visit adds and subtracts only)
                typedef const __attribute__((address_space(3))) float *lds_fp;
                const unsigned wx0 = lds_addr(L.wx) + (unsigned)((mx + 2 * CW) * 4);
                const unsigned wy0 = lds_addr(L.wy) + (unsigned)((my + 2 * CW) * 4);
                // software pipeline: the sorted entry of visit i+1 is fetched while visit i is processed
                uint2 ent_next = make_uint2(0u, 0u);
                if (0 < total) ent_next = entry_of(0);
                for (int i = 0; i < total; ++i) {
                    const uint2 ent = ent_next;
                    if (i + 1 < total) ent_next = entry_of(i + 1);
                    // the samples are asked for FIRST, with the weights: one LDS round trip per visit, not two
                    float4 dd[CPB / 2 > 0 ? CPB / 2 : 1];
                    const char *drec = reinterpret_cast<const char *>(L.d) + (ent.y >> 16);
                    if (CPB % 2 == 0) {
#pragma unroll
                        for (int c = 0; c < CPB / 2; ++c) dd[c] = *reinterpret_cast<const float4 *>(drec + (size_t)c * C::NREC * sizeof(float4));
                    }
                    const unsigned woff = ent.x & 0x7fffu;
                    const lds_fp wxr = (lds_fp)(size_t)(wx0 + woff - ((ent.x >> 15) & 0xffu));
                    const lds_fp wyr = (lds_fp)(size_t)(wy0 + woff - ((ent.x >> 23) & 0xffu));
                    const float wxa = wxr[0], wxb = wxr[1];
                    const float wya = wyr[0], wyb = wyr[1];
                    const int ar = (int)(ent.y & 0xffffu);
                    float wq[4];
                    wq[0] = wxa * wya; wq[1] = wxb * wya; wq[2] = wxa * wyb; wq[3] = wxb * wyb;   // src/tron.cu:516
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((unsigned)(ar - Rlo[q]) > (unsigned)(Rhi[q] - Rlo[q])) wq[q] = 0.f;    // src/tron.cu:512,521
                    if (has_centre && (int)ent.x < 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (Rlo[q] == 0) wq[q] += wq[q];                                      // r = 0 sits in both loops
                    }
                    if (CPB % 2 == 0) {
#pragma unroll
                        for (int c = 0; c < CPB / 2; ++c) {
                            const float4 d = dd[c];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc[q][2 * c].x = fmaf(d.x, wq[q], acc[q][2 * c].x);              // src/tron.cu:519
                                acc[q][2 * c].y = fmaf(d.y, wq[q], acc[q][2 * c].y);
                                acc[q][2 * c + 1].x = fmaf(d.z, wq[q], acc[q][2 * c + 1].x);
                                acc[q][2 * c + 1].y = fmaf(d.w, wq[q], acc[q][2 * c + 1].y);
                            }
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < CPB; ++c) {
                            const float2 d = *reinterpret_cast<const float2 *>(drec + (size_t)c * C::NREC * sizeof(float2));
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc[q][c].x = fmaf(d.x, wq[q], acc[q][c].x);
                                acc[q][c].y = fmaf(d.y, wq[q], acc[q][c].y);
                            }
                        }
                    }
                }
            }
            PROF_MARK(12);                                      // apply
            lds_barrier();
            PROF_MARK(13);                                      // barrier
            sp0 = nsp0;
            sp1 = nsp1;
        }
    }

    PROF_MARK(14);
    if (nparts > 1 || inner) {
        // partial tile of this spoke range, tile-local [coil][row][col], already scaled; summed by grid_reduce_parts_kernel
        // (slice groups: the group's vs * nchan channels take the place of the coils)
        const int pch = vs > 1 ? vs * p.nchan : p.nchan;
        float2 *part_base = p.partial + ((((size_t)z * p.nsplit_slots + slot) * p.max_parts + part) * pch) * (kBinTile * kBinTile);
#pragma unroll
        for (int qy = 0; qy < 2; ++qy)
#pragma unroll
            for (int c = 0; c < CPB; ++c)
                if (c < ncb) {
                    float4 v;
                    v.x = acc[2 * qy][c].x * p.scale; v.y = acc[2 * qy][c].y * p.scale;
                    v.z = acc[2 * qy + 1][c].x * p.scale; v.w = acc[2 * qy + 1][c].y * p.scale;
                    *reinterpret_cast<float4 *>(part_base + (size_t)(c0 + c) * (kBinTile * kBinTile) + (my + qy) * kBinTile + mx) = v;
                }
        PROF_FLUSH;
        return;
    }
#pragma unroll
    for (int qy = 0; qy < 2; ++qy)
#pragma unroll
        for (int c = 0; c < CPB; ++c)
            if (c < ncb) {
                float4 v;
                v.x = acc[2 * qy][c].x * p.scale;                       // src/tron.cu:532-534
                v.y = acc[2 * qy][c].y * p.scale;
                v.z = acc[2 * qy + 1][c].x * p.scale;
                v.w = acc[2 * qy + 1][c].y * p.scale;
                if (vs > 1) store_point_pair(p, zbase + c / p.nchan, c % p.nchan, X0, Y0 + qy, v);
                else store_point_pair(p, z, c0 + c, X0, Y0 + qy, v);
            }
    PROF_MARK(15);                                              // output store
    PROF_FLUSH;
}

// Adds the partial tiles of a split tile in part order (fixed order: results do not depend on scheduling) and stores
// the sum exactly as an unsplit workgroup would.  grid = (split slots x slices [or slice groups], channels), block = 256.
__global__ void __launch_bounds__(kBinThreads)
grid_reduce_parts_kernel(const GridParams p)
{
    const int z = blockIdx.x % p.nslices;
    const int slot = blockIdx.x / p.nslices;
    const int entry = p.split_slots[slot];                      // tile id | parts << 20
    const int tile = entry & 0xffff, nparts = (entry >> 20) & 15;
    // slice groups (vslices > 1): channel ch of group z is coil ch % nchan of slice z * vslices + ch / nchan
    const int vs = p.vslices > 1 ? p.vslices : 1;
    const int pch = vs > 1 ? vs * p.nchan : p.nchan;
    const int ch = vs > 1 ? (int)blockIdx.y : p.coil0 + (int)blockIdx.y;
    const int zs = vs > 1 ? z * vs + ch / p.nchan : z;
    const int c = vs > 1 ? ch % p.nchan : ch;
    if (vs > 1 && zs >= p.nslices_total) return;
    const int n = p.nxos, h = n / 2;
    const bool inner = p.inner_r0 > 0 && tile == p.ntiles;      // origin-centred tile: added onto the centre tiles' stores
    const int x0 = inner ? -kBinTile / 2 : (tile % p.tiles_per_row) * kBinTile - h;
    const int y0 = inner ? -kBinTile / 2 : (tile / p.tiles_per_row) * kBinTile - h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mx = 2 * (lane & 15), my = 8 * wave + 2 * (lane >> 4);
    const float2 *base = p.partial + (((size_t)z * p.nsplit_slots + slot) * p.max_parts * pch + ch) * (kBinTile * kBinTile);
#pragma unroll
    for (int qy = 0; qy < 2; ++qy) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        // eight parts at a time with all their loads in flight (a load-add chain per part made this tiny kernel take
        // one HBM round trip per part); summed in part order all the same
        for (int g0 = 0; g0 < nparts; g0 += 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = g0 + k < nparts
                           ? *reinterpret_cast<const float4 *>(base + (size_t)(g0 + k) * pch * (kBinTile * kBinTile) + (my + qy) * kBinTile + mx)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (g0 + k < nparts) { sum.x += v[k].x; sum.y += v[k].y; sum.z += v[k].z; sum.w += v[k].w; }
        }
        if (inner) store_point_pair<true>(p, zs, c, x0 + mx, y0 + my + qy, sum);
        else store_point_pair(p, zs, c, x0 + mx, y0 + my + qy, sum);
    }
}

template <int CPB, int CW>
static hipError_t launch_binned_cpb(const GridParams &p, int half_in, hipStream_t s)
{
    const int tpr = (p.nxos + kBinTile - 1) / kBinTile;
    GridParams q = p;
    q.tiles_per_row = tpr;
    q.ntiles = tpr * tpr;
    const int chunks = p.vslices > 1 ? 1 : (p.nchan - p.coil0 + CPB - 1) / CPB;
    const int entries = p.tile_entries > 0 ? p.tile_entries : q.ntiles;
    dim3 grid((unsigned)((size_t)entries * q.nslices), (unsigned)chunks);
    size_t lds = sizeof(BinLds<CPB, CW>);
    // sample path: complex-half and odd coil counts / slice groups / unaligned streams go through registers; fp32 with
    // even coil counts (16-byte coil pairs) is copied global -> LDS directly
    int in_mode = half_in ? kInRegs16 : kInRegs32;
    if (!half_in && CPB % 2 == 0 && p.vslices <= 1 && (p.nchan & 1) == 0 && (p.coil0 & 1) == 0
        && (reinterpret_cast<uintptr_t>(p.nudata) & 15) == 0)
        in_mode = kInLdsDma;
    if (lds > 64 * 1024) {   // above the default dynamic-LDS limit: raise it once per instantiation and device
        const void *fns[3] = {reinterpret_cast<const void *>(grid_binned_kernel<CPB, CW, kInRegs32>),
                              reinterpret_cast<const void *>(grid_binned_kernel<CPB, CW, kInRegs16>),
                              reinterpret_cast<const void *>(grid_binned_kernel<CPB, CW, (CPB % 2 == 0 ? kInLdsDma : kInRegs32)>)};
        for (const void *f : fns) {
            const hipError_t e1 = allow_dynamic_lds(f, (int)sizeof(BinLds<CPB, CW>));
            if (e1 != hipSuccess) return e1;
        }
    }
    if (in_mode == kInRegs16)
        hipLaunchKernelGGL((grid_binned_kernel<CPB, CW, kInRegs16>), grid, dim3(kBinThreads), lds, s, q);
    else if (in_mode == kInLdsDma)
        hipLaunchKernelGGL((grid_binned_kernel<CPB, CW, (CPB % 2 == 0 ? kInLdsDma : kInRegs32)>), grid, dim3(kBinThreads), lds, s, q);
    else
        hipLaunchKernelGGL((grid_binned_kernel<CPB, CW, kInRegs32>), grid, dim3(kBinThreads), lds, s, q);
    if (q.nsplit_slots > 0)
        hipLaunchKernelGGL(grid_reduce_parts_kernel, dim3((unsigned)((size_t)q.nsplit_slots * q.nslices),
                                                          (unsigned)(p.vslices > 1 ? p.vslices * p.nchan : p.nchan - p.coil0)),
                           dim3(kBinThreads), 0, s, q);
    return hipGetLastError();
}

template <int CW>
static hipError_t launch_binned_cw(const GridParams &p, int half_in, hipStream_t s)
{
    if (p.vslices > 1) {                       // slices in the coil dimension: vslices * nchan channels per pass
        const int nv = p.vslices * p.nchan;
        if (nv > 8 || (p.nsplit_slots > 0 && p.inner_r0 <= 0) || p.coil0 != 0 || p.trig_slice_stride != 0) return hipErrorInvalidValue;
        if (nv > 4) return launch_binned_cpb<8, CW>(p, half_in, s);
        if (nv > 2) return launch_binned_cpb<4, CW>(p, half_in, s);
        return launch_binned_cpb<2, CW>(p, half_in, s);
    }
    const int nc = p.nchan - p.coil0;
    // one padded 8-coil pass beats two 4-coil passes: the per-sample work (weights, sort) is paid per pass
    // (6 coils: 3.3 vs 4.9 us per coil-slice; 12 coils: 4.1 vs 4.4); passes of 6 coils where they pad less than passes of 8
    // (5, 6, 10, 12 coils: the whole-body data set has 6) -- fp32 input only, complex-half loads come in fours
    if (nc >= 5) {
        const int pad8 = (nc + 7) / 8 * 8, pad6 = (nc + 5) / 6 * 6;
        if (pad6 < pad8 && !half_in) return launch_binned_cpb<6, CW>(p, half_in, s);
        return launch_binned_cpb<8, CW>(p, half_in, s);
    }
    if (nc >= 4) return launch_binned_cpb<4, CW>(p, half_in, s);
    if (nc >= 2) return launch_binned_cpb<2, CW>(p, half_in, s);
    return launch_binned_cpb<1, CW>(p, half_in, s);
}

// p.tile_order must list the 32x32 tiles (see build_tile_order(nxos, 32, ...)).
hipError_t launch_grid_binned(const GridParams &p, int half_in, hipStream_t s)
{
    const int cw = (int)ceilf(p.W);
    switch (cw) {
        case 1: return launch_binned_cw<1>(p, half_in, s);
        case 2: return launch_binned_cw<2>(p, half_in, s);
        case 3: return launch_binned_cw<3>(p, half_in, s);
        default: return hipErrorInvalidValue;
    }
}


#ifdef TRON_PHASE_CLOCK
extern "C" __attribute__((visibility("default"))) int tron_debug_grid_profile(unsigned long long *out, int n)   // reads and clears the phase clock
{
    static unsigned long long h[kProfCopies * kProfSlots];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_bin_prof), sizeof(h)) != hipSuccess) return 1;
    for (int i = 0; i < n && i < kProfSlots; ++i) {
        out[i] = 0;
        for (int c = 0; c < kProfCopies; ++c) out[i] += h[c * kProfSlots + i];
    }
    for (size_t i = 0; i < sizeof(h) / sizeof(h[0]); ++i) h[i] = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bin_prof), h, sizeof(h)) != hipSuccess;
}
#endif

__global__ void warm_grid_binned_tu() {}

hipError_t warm_grid_binned()   // see warm_kernels() in tron_kernels.hip
{
    hipLaunchKernelGGL(warm_grid_binned_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
