// Streaming tiled degridding = degridradial2d of the reference (src/tron.cu:540-577) for launches of many images.
//
// degrid_tile_kernel (tron_degrid_tile.hip) loads one tile, clips the spokes, runs the sample loop and retires, three
// workgroups per CU by LDS: the tile load and the clipping of one overlap the sample loops of the others only by chance.
// Here ONE workgroup of 768 threads owns a CU (three waves per SIMD, as there) and walks a run of images of its tile and
// coil chunk: the next image's tile + halo is copied global -> LDS by the DMA path (global_load_lds_dwordx4, no
// registers) into the second of two tile buffers while the current one is sampled, and the spoke lists are built once
// per run when every image has the same angles.  How many images a run holds depends on the tile: the centre tiles hold
// the most samples (density ~ 1/r) and take short runs so that no workgroup outlasts the launch (DegridParams::group_end).
// Measured and not kept (8 coils x 64 images, 1.68-1.78 us per coil image as built; DESIGN.md sections 4.4 and 8 have the
// numbers): waves entering the sample loop a fraction of a pass apart; 512 and 1 024 threads; the records of ALL images of
// a run as one stream of 64-record blocks over three tile buffers, dealt a block per wave and step or handed out through an
// LDS counter with no barrier at all; the kept records dealt evenly over the waves; the wait for the next copy moved between
// a wave's first gather and its first store; three tile buffers with counted vmcnt waits.  Elimination builds: copies,
// clipping and barriers alone 0.74 us, with the gather 1.37, with the stores instead 1.44, everything 1.70 -- the copies and
// the stores add up as if they shared one path, the gather (LDS ~50 % busy, bank conflicts 2.25-fold measured and 2.7-fold
// simulated for ANY row pitch: 32 samples along a spoke always meet a short vector of the bank lattice) overlaps them in part.
// Round 5 (profiles/round5_forward_dealing_ab.log, round5_forward_unkept_records.log): the kept records are dealt by bank class
// (below: 1.2 LDS cycles per read where the sorted order of round 3 met 1.8; degridding -4 to -5 % in interleaved runs on two boxes).
// Measured and not kept: three kept passes instead of two; an order of the UNKEPT records other than along their spokes (any fixed
// permutation inside a 64-record block: 3.0-3.3 cycles per read against 2.7, simulated); the long lists round the centre dealt to
// several workgroups in equal shares, every record kept and every run 16 images (1.62 against 1.59 us: each share loads the whole
// tile again, +29 % tile loads); ONE workgroup for both coil chunks of a (tile, image) at eight coils -- chunk 0's sums of the kept records
// waiting in registers (149 VGPRs), whole 64-byte sample records stored by four lanes each, lists / sort / preparation once for both --
// bit-identical and 3 % SLOWER (1.58 against 1.53 us: half the workgroups, and the L2 was merging the 32-byte halves anyway).
// The records beyond the kept passes -- 21 % of the bench's samples, in the 32 tiles round the
// centre -- cost 0.40 of 1.62 us (a build that skips them); the drain of an image's stores behind vmcnt(0) costs nothing (a build
// without the wait).
//
// The tile is held as the input planes lie in memory ([col][row] for the fused forward FFT, which stores the grid
// transposed): the sample loop strides accordingly.  The halo is rounded up to even widths so that a 16-byte piece
// (two points) never straddles the periodic wrap (src/tron.cu:569-570) or the fftshift of :646, both folded into the
// source index.  Coordinates, weights and accumulation order are dg_sample_loop's, i.e. the reference's.
#include <stdlib.h>

#include <algorithm>

#include "tron_degrid_sample.h"

namespace tron {

constexpr int kDsThreads = 768;     // three waves per SIMD (512 and 1024 threads measured the same within noise)
constexpr int kDsMaxSpokes = 512;   // spokes clipped per round
constexpr int kDsMaxBlocks = 512;   // 64-record blocks indexed by the inverse map
constexpr bool kDsSortLong = true;   // a list longer than the kept passes (the tiles round the centre): its first kDsKeep passes are dealt the same way
constexpr int kDsKeepPasses = 2;
constexpr int kDsCoils = 4;         // coils per workgroup (32-byte pieces of the coil-interleaved output lines)

template <int CW>
struct DsLds : DgLists<kDsMaxSpokes, kDsMaxBlocks, kDsThreads / 64> {
    static constexpr int HALO = (CW + 1) & ~1;                // points before the tile: ceil(W), rounded up to even (16-byte pieces)
    static constexpr int TS = (kDgTile + HALO + CW + 2) & ~1; // ... and ceil(W) + 1 after it (the fast weights' last slot), rounded likewise
    // row pitch in points: = 6 or 12 mod 32, so that no short step (a rows, b points) lands on the same pair of the 64
    // banks (a * PITCH + b = 0 mod 32 has no solution shorter than 5): spokes of any angle gather with few conflicts
    static constexpr int PITCH = TS <= 38 ? 38 : 44;
    static constexpr int PLANE = TS * PITCH;
    static constexpr int BUF = PLANE * kDsCoils;              // points per tile buffer
    alignas(16) float2 tile[2 * BUF + 16];                    // two buffers of [coil][a][b], b along memory; zeroed pad
    float4 stage[(kDsThreads / 64) * 128];                    // per wave: the sums of a pass on their way to the lanes that store them (dg_gather_store, PAIRS)
};

template <int CW, bool TR, int KB>
__global__ void __launch_bounds__(kDsThreads) degrid_stream_kernel(const DegridParams p)
{
    using L_t = DsLds<CW>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    L_t &L = *reinterpret_cast<L_t *>(lds_raw);
    constexpr int HALO = L_t::HALO, TS = L_t::TS, PITCH = L_t::PITCH, PLANE = L_t::PLANE, BUF = L_t::BUF;
    constexpr int SX = TR ? 1 : PITCH, SY = TR ? PITCH : 1;          // LDS strides of a step along the sine / cosine axis

    const int tid = threadIdx.x, wave = tid >> 6;
    const int n = p.n;                                          // columns (cosine axis, "Y" of the reference)
    const int nr = p.nrows > 0 ? p.nrows : n;                   // rows (sine axis, "X")
    const int tpr = n / kDgTile, tprr = nr / kDgTile;
    // Workgroup id -> (tile, image run, coil chunk).  The coil chunks of one (tile, run) write interleaved 32-byte pieces
    // of the same output lines (samples are coil-interleaved, src/tron.cu:550): they are placed 8 ids apart, i.e. on the
    // same XCD and in step with each other, so its L2 merges the pieces before they reach HBM.
    const int chunks = (p.nrep + kDsCoils - 1) / kDsCoils;
    const int grp = blockIdx.x / (8 * chunks), within = blockIdx.x % (8 * chunks);
    int rem = grp * 8 + (within & 7);                           // (tile, run) index: classes of run length 1, 2, 4, 8, 16
    int tpos = 0, run = 1, k0 = -1;
    {
        const int ntiles = tpr * tprr;
        int t0 = 0;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int t1 = c < 4 ? min(p.group_end[c], ntiles) : ntiles;
            const int g = min(1 << c, p.group_max);
            const int runs = (p.nimg + g - 1) / g;
            const int cnt = max(t1 - t0, 0) * runs;
            if (k0 < 0) {
                if (rem < cnt) { tpos = t0 + rem / runs; run = g; k0 = (rem % runs) * g; }
                else rem -= cnt;
            }
            t0 = max(t0, t1);
        }
    }
    if (k0 < 0) return;
    const int k1 = min(p.nimg, k0 + run);
    const int tile = p.tile_order ? p.tile_order[tpos] : tpos;
    const int c0 = (within >> 3) * kDsCoils;
    const int ncb = min(kDsCoils, p.nrep - c0);
    const int tx0 = (tile / tpr) * kDgTile;                     // first row (sine axis, "X" of the reference)
    const int ty0 = (tile % tpr) * kDgTile;                     // first column (cosine axis, "Y")

    {   // every sample lies within n/2 of the grid centre (src/tron.cu:554-561: |R| <= 1/2): a tile whose nearest cell is
        // farther away owns none
        const float hc = (float)((n + 1) / 2);
        const float dx = fmaxf(fmaxf((float)tx0 - hc, hc - (float)(tx0 + kDgTile)), 0.f);
        const float dy = fmaxf(fmaxf((float)ty0 - hc, hc - (float)(ty0 + kDgTile)), 0.f);
        const float lim = 0.5f * (float)n + 1.5f;
        if (nr == n && dx * dx + dy * dy > lim * lim) return;
    }
    KbCoef kb;
    kb.W = p.W; kb.invW = 1.0f / p.W; kb.beta = p.beta;
#pragma unroll
    for (int t = 0; t < kKbPolyTerms; ++t) kb.poly[t] = p.kb_poly[t];

    // ---- the tile's 16-byte pieces: source offsets once per workgroup (periodic wrap of src/tron.cu:569-570 and the
    //      fftshift(INVERSE) of :646 folded in), piece q -> LDS bytes [16 q, 16 q + 16) of a buffer
    constexpr int PPC = PLANE / 2;                              // pieces per coil plane (those past a row's TS points: padding, not fetched)
    constexpr int NIT = (PPC * kDsCoils + kDsThreads - 1) / kDsThreads;
    unsigned voff[NIT];
    bool vact[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int q = it * kDsThreads + tid;
        const int c = q / PPC, e = q - c * PPC;
        const int a = e / (PITCH / 2), b = 2 * (e - a * (PITCH / 2));
        int i = tx0 - HALO + (TR ? b : a), j = ty0 - HALO + (TR ? a : b);
        i += i < 0 ? nr : 0; i -= i >= nr ? nr : 0;              // -nr <= i < 2 nr: one step each way
        j += j < 0 ? n : 0; j -= j >= n ? n : 0;
        if (p.in_shift) {
            i += nr / 2; if (i >= nr) i -= nr;
            j += n / 2; if (j >= n) j -= n;
        }
        if (TR) { i += p.in_rot; if (i >= nr) i -= nr; } else { j += p.in_rot; if (j >= n) j -= n; }
        voff[it] = (unsigned)(((size_t)c * p.in_c + (TR ? (size_t)j * nr + i : (size_t)i * n + j)) * sizeof(float2));
        vact[it] = c < ncb && b < TS;
    }
    const unsigned tile_lds = lds_addr(L.tile);
    auto fetch = [&](const int k, const int buf) {
        const float2 *src = p.udata + (size_t)k * p.in_z + (size_t)c0 * p.in_c;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const unsigned dst = tile_lds + (unsigned)((buf * BUF) * sizeof(float2)) + (unsigned)((it * kDsThreads + wave * 64) * 16);
            if (vact[it]) lds_dma16_s(src, voff[it], (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
        }
    };
    fetch(k0, 0);
    if (tid < 16) L.tile[2 * BUF + tid] = make_float2(0.f, 0.f);

    const int nrounds = (p.npe + kDsMaxSpokes - 1) / kDsMaxSpokes;
    DgRound rd;
    rd.nacc = rd.nrec = 0; rd.mapped = true;
    // Same angles every image and one clipping round: the records are the same for every image of the run, and so is all a
    // sample needs from its record -- weights, footprint, destination (DgPrep).  Fast weights: the first kDsKeep passes'
    // worth is computed once and kept in registers over the run (a tile holds ~1 000 records on average, 1.3 passes of 768;
    // the counters showed VALU and LDS time adding up rather than overlapping, and two thirds of a pass's VALU
    // instructions are this preparation); later passes and the exact weights take the per-record loop.
    constexpr int kDsKeep = kDsKeepPasses;
    const bool same_records = nrounds == 1 && p.trig_img_stride == 0;
    DgPrep<2 * CW> kept[kDsKeep];
#pragma unroll
    for (int j = 0; j < kDsKeep; ++j) kept[j].own = false;
    if (same_records) {
        rd = dg_clip_round<kDsThreads, kDsMaxSpokes, kDsMaxBlocks>(p, L, k0, 0, tid, tx0, ty0, n, nr);
        if (KB == TRON_KB_FAST && (rd.nrec <= kDsKeep * kDsThreads || kDsSortLong)) {
            // All records fit the kept passes: they are dealt out BY THE BANKS THEIR FOOTPRINT STARTS ON.  Every LDS read of a record
            // sits at a fixed offset from the point its footprint starts at, and a ds_read_b64 serves a wave in two groups of 32 lanes,
            // one LDS cycle per distinct address on a group's busiest pair of banks (pair = 8-byte word mod 32): a group whose lanes
            // start on 32 different pairs gathers all its 16 x coils reads without a conflict.  So: counting sort by start point (round
            // 3: neighbouring lanes gather neighbouring points; 64 samples along a spoke met 2.7 cycles per read, the sorted order 1.8 --
            // tools/probe/degrid_deal_sim.py replays the bench's trajectory), then CLASS = start point mod 32, RANK = the record's place
            // among its class in sorted order, and row r of the deal = the records of rank r, lane = class.  Rows are 32 wide while
            // every class still has a record of that rank; rows at least kDsRowPad wide keep their holes (a whole group, no
            // conflict), narrower ones follow each other without gaps (1.2 cycles per read over the bench's tiles at 4 % more lane
            // slots).  Records of other tiles' samples drop out here instead of idling a lane every image.  Once per run.
            constexpr int EPT = (PLANE + kDsThreads - 1) / kDsThreads;      // points per thread in the scan
            constexpr int NSTR = (PLANE + 31) / 32;                         // points of one class
            constexpr int NQ = kDsThreads / 32, EQ = (NSTR + NQ - 1) / NQ;  // ... dealt to NQ threads, EQ each
            constexpr int kDsRowPad = 24, kDsMaxRows = 128;     // (33: no padded rows)
            unsigned *hist = reinterpret_cast<unsigned *>(L.stage);         // [PLANE] counts, then first positions
            unsigned short *perm = reinterpret_cast<unsigned short *>(hist + PLANE);   // [kDsKeep * kDsThreads] record ids by lane slot (0xffff: none)
            unsigned short *cpre = perm + kDsKeep * kDsThreads;             // [PLANE] records of the point's class at points before it
            unsigned short *part = cpre + ((PLANE + 1) & ~1);               // [NQ][32] a thread's share of its class
            unsigned *rowmask = reinterpret_cast<unsigned *>(part + NQ * 32);   // [kDsMaxRows] classes that have a record of rank r
            unsigned short *rowpos = reinterpret_cast<unsigned short *>(rowmask + kDsMaxRows);   // [kDsMaxRows] first lane slot of row r
            unsigned short *ccount = rowpos + kDsMaxRows;                   // [32] records per class
            int *deal = reinterpret_cast<int *>(ccount + 32);               // padded rows (-1: the plain sorted order), lane slots in all; [2..] the row scan's wave sums
            static_assert(kDsThreads % 32 == 0 && (PLANE * 4 + kDsKeep * kDsThreads * 2 + ((PLANE + 1) & ~1) * 2 + NQ * 64 + kDsMaxRows * 6 + 64 + 4 * (4 + 3 * (kDsMaxRows / 64))) <= (int)sizeof(L.stage),
                          "sort tables must fit the stage area");
            for (int i = tid; i < PLANE; i += kDsThreads) hist[i] = 0u;
            for (int i = tid; i < kDsKeep * kDsThreads; i += kDsThreads) perm[i] = 0xffffu;
            __syncthreads();
            int slot[kDsKeep], rank[kDsKeep];
#pragma unroll
            for (int j = 0; j < kDsKeep; ++j) {
                const int rec = tid + j * kDsThreads;
                slot[j] = -1; rank[j] = 0;
                if (rec < rd.nrec) {
                    int pe, ro;
                    float X, Y;
                    if (dg_coords<kDsMaxSpokes>(p, L, rd, rec, tid & 63, tx0, ty0, n, nr, pe, ro, X, Y)) {
                        slot[j] = ((int)ceilf(X - p.W) + HALO - tx0) * SX + ((int)ceilf(Y - p.W) + HALO - ty0) * SY;
                        rank[j] = (int)atomicAdd(&hist[slot[j]], 1u);
                    }
                }
            }
            __syncthreads();
            {   // exclusive scan of the counts
                unsigned cnt[EPT], tsum = 0u;
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const int i = tid * EPT + e;
                    cnt[e] = i < PLANE ? hist[i] : 0u;
                    tsum += cnt[e];
                }
                unsigned v = tsum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned t = __shfl_up(v, o);
                    if ((tid & 63) >= o) v += t;
                }
                if ((tid & 63) == 63) L.wcnt[wave] = (int)v;
                __syncthreads();
                unsigned run = v - tsum;
                for (int w = 0; w < wave; ++w) run += (unsigned)L.wcnt[w];
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const int i = tid * EPT + e;
                    if (i < PLANE) hist[i] = run;
                    run += cnt[e];
                }
            }
            int nown = 0;
            for (int w = 0; w < kDsThreads / 64; ++w) nown += L.wcnt[w];
            __syncthreads();
            {   // per class (start point mod 32): the records at earlier points -- thread (class, q) sums its EQ points, then adds its predecessors' sums
                const int c = tid & 31, q = tid >> 5;
                unsigned pc[EQ], psum = 0u;
#pragma unroll
                for (int e = 0; e < EQ; ++e) {
                    const int s_ = c + 32 * (q * EQ + e);
                    pc[e] = (q * EQ + e < NSTR && s_ < PLANE) ? (s_ + 1 < PLANE ? hist[s_ + 1] : (unsigned)nown) - hist[s_] : 0u;
                    psum += pc[e];
                }
                part[q * 32 + c] = (unsigned short)psum;
                __syncthreads();
                unsigned base = 0u;
                for (int qq = 0; qq < q; ++qq) base += part[qq * 32 + c];
#pragma unroll
                for (int e = 0; e < EQ; ++e) {
                    const int s_ = c + 32 * (q * EQ + e);
                    if (q * EQ + e < NSTR && s_ < PLANE) cpre[s_] = (unsigned short)base;
                    base += pc[e];
                }
                if (q == NQ - 1) ccount[c] = (unsigned short)base;
            }
            __syncthreads();
            {   // rows: thread = rank (the first kDsMaxRows threads, i.e. two waves)
                unsigned mask = 0u;
                int cmax = 0;
#pragma unroll 8
                for (int c = 0; c < 32; ++c) {
                    const int cc = ccount[c];
                    mask |= (cc > tid ? 1u : 0u) << c;
                    cmax = max(cmax, cc);
                }
                const bool row = tid < kDsMaxRows;
                const int width = row ? __popc(mask) : 0;
                // rows at least kDsRowPad wide keep their holes (widths never grow with the rank: those rows are a prefix); the others
                // follow without gaps: an exclusive scan of their widths -- and of ALL widths, should the holes overflow the kept passes
                const unsigned long long padded = __ballot(width >= kDsRowPad);
                if (row && (tid & 63) == 0) deal[2 + wave] = __popcll(padded);
                __syncthreads();
                int rfull = 0;
#pragma unroll
                for (int w = 0; w < kDsMaxRows / 64; ++w) rfull += deal[2 + w];
                const int wa = tid >= rfull ? width : 0;
                int va = wa, vb = width;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int ta = __shfl_up(va, o), tb = __shfl_up(vb, o);
                    if ((tid & 63) >= o) { va += ta; vb += tb; }
                }
                if (row && (tid & 63) == 63) { deal[4 + 2 * wave] = va; deal[5 + 2 * wave] = vb; }
                __syncthreads();
                int ba = 0, bb = 0, total_a = 0;
#pragma unroll
                for (int w = 0; w < kDsMaxRows / 64; ++w) {
                    if (w < wave) { ba += deal[4 + 2 * w]; bb += deal[5 + 2 * w]; }
                    total_a += deal[4 + 2 * w];
                }
                const bool holes_fit = rfull * 32 + total_a <= kDsKeep * kDsThreads;
                if (row) {
                    rowmask[tid] = mask;
                    rowpos[tid] = (unsigned short)(holes_fit ? rfull * 32 + ba + va - wa : bb + vb - width);
                }
                if (tid == 0) {
                    // a class of more than kDsMaxRows records (few spokes, all along one line of the bank lattice): round 3's sorted order
                    deal[0] = cmax > kDsMaxRows ? -1 : (holes_fit ? rfull : 0);
                    deal[1] = cmax > kDsMaxRows || !holes_fit ? nown : rfull * 32 + total_a;
                }
            }
            __syncthreads();
            const int rfull = deal[0], nslots = deal[1];
#pragma unroll
            for (int j = 0; j < kDsKeep; ++j)
                if (slot[j] >= 0) {
                    int pos = (int)hist[slot[j]] + rank[j];                   // place in sorted order
                    if (rfull >= 0) {
                        const int c = slot[j] & 31, rc = (int)cpre[slot[j]] + rank[j];
                        pos = rc < rfull ? rc * 32 + c : (int)rowpos[rc] + __popc(rowmask[rc] & ((1u << c) - 1u));
                    }
                    perm[pos] = (unsigned short)(tid + j * kDsThreads);
                }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kDsKeep; ++j) {
                const int idx = tid + j * kDsThreads;
                if (idx < nslots) {
                    const int rec = perm[idx];
                    if (rec != 0xffff) kept[j] = dg_prep<CW, kDsMaxSpokes, HALO, SX, SY>(p, kb, L, rd, rec, rec & 63, tx0, ty0, n, nr, c0);
                }
            }
            __syncthreads();                                                // the stage area goes back to the waves
        } else if (KB == TRON_KB_FAST) {
#pragma unroll
            for (int j = 0; j < kDsKeep; ++j) {
                const int rec = tid + j * kDsThreads;
                if (rec < rd.nrec) kept[j] = dg_prep<CW, kDsMaxSpokes, HALO, SX, SY>(p, kb, L, rd, rec, tid & 63, tx0, ty0, n, nr, c0);
            }
        }
    }
    const int first_rec = (same_records && KB == TRON_KB_FAST) ? kDsKeep * kDsThreads : 0;
    for (int k = k0; k < k1; ++k) {
        const int buf = (k - k0) & 1;
        if (!same_records) rd = dg_clip_round<kDsThreads, kDsMaxSpokes, kDsMaxBlocks>(p, L, k, 0, tid, tx0, ty0, n, nr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's pieces of image k have landed
        __syncthreads();                                        // ... everybody's; and nobody still samples the other buffer
        if (k + 1 < k1) fetch(k + 1, buf ^ 1);
        float2 *dst = p.nudata + (size_t)k * p.nro * p.npe * p.nrep;
        {
            const bool pairs = ncb == kDsCoils && (p.nrep & 1) == 0;      // whole 16-byte pieces
#pragma unroll
            for (int j = 0; j < kDsKeep; ++j) {
                if (pairs) {
                    // (the same in the per-record loop -- later passes, CGNR's per-image lists -- measured no gain)
                    if (__any(kept[j].own)) dg_gather_store<kDsCoils, CW, PLANE, SX, SY, true, true>(p, L, buf * BUF, kept[j], dst, ncb, L.stage + wave * 128);
                } else if (kept[j].own) {
                    dg_gather_store<kDsCoils, CW, PLANE, SX, SY, true>(p, L, buf * BUF, kept[j], dst, ncb);
                }
            }
        }
        for (int r = 0; r < nrounds; ++r) {
            if (r > 0) {
                __syncthreads();                                // the lists of the round before are no longer read
                rd = dg_clip_round<kDsThreads, kDsMaxSpokes, kDsMaxBlocks>(p, L, k, r * kDsMaxSpokes, tid, tx0, ty0, n, nr);
            }
            dg_sample_loop<kDsCoils, CW, KB, kDsThreads, kDsMaxSpokes, PLANE, HALO, SX, SY, true>(p, kb, L, buf * BUF, rd, tid, tx0, ty0, n, nr,
                                                                                         dst, c0, ncb, first_rec);
        }
        if (!same_records && k + 1 < k1) __syncthreads();       // before the lists are rebuilt
    }
}

template <int CW, bool TR>
static hipError_t launch_degrid_stream_cw(const DegridParams &p, int kb_mode, hipStream_t s)
{
    const int ntiles = (p.n / kDgTile) * ((p.nrows > 0 ? p.nrows : p.n) / kDgTile);
    const int chunks = (p.nrep + kDsCoils - 1) / kDsCoils;
    size_t nruns = 0;
    int t0 = 0;
    for (int c = 0; c < 5; ++c) {
        const int t1 = c < 4 ? std::min(p.group_end[c], ntiles) : ntiles;
        const int g = std::min(1 << c, p.group_max);
        nruns += (size_t)std::max(t1 - t0, 0) * ((p.nimg + g - 1) / g);
        t0 = std::max(t0, t1);
    }
    dim3 grid((unsigned)(((nruns + 7) / 8) * 8 * chunks));
    const size_t lds = sizeof(DsLds<CW>);
    static_assert(sizeof(DsLds<CW>) <= 160 * 1024, "two tile buffers and the spoke lists must fit the CU's LDS");
    if (kb_mode == TRON_KB_EXACT) {
        const hipError_t once = allow_dynamic_lds(reinterpret_cast<const void *>(degrid_stream_kernel<CW, TR, TRON_KB_EXACT>), (int)sizeof(DsLds<CW>));
        if (once != hipSuccess) return once;
        hipLaunchKernelGGL((degrid_stream_kernel<CW, TR, TRON_KB_EXACT>), grid, dim3(kDsThreads), lds, s, p);
    } else {
        const hipError_t once = allow_dynamic_lds(reinterpret_cast<const void *>(degrid_stream_kernel<CW, TR, TRON_KB_FAST>), (int)sizeof(DsLds<CW>));
        if (once != hipSuccess) return once;
        hipLaunchKernelGGL((degrid_stream_kernel<CW, TR, TRON_KB_FAST>), grid, dim3(kDsThreads), lds, s, p);
    }
    return hipGetLastError();
}

// Launches of many images on grids of whole tiles: see the head of this file.  Anything else: degrid_tile_kernel.
// (Angles that differ from image to image -- CGNR's sliding windows -- rebuild the spoke lists per image: at 16 images
// of 8 coils the streaming kernel then takes 247 us where the tile kernel takes 241.)
bool degrid_stream_supported(const DegridParams &p, int kb_mode)
{
    const int nr = p.nrows > 0 ? p.nrows : p.n;
    const int cw = (int)ceilf(p.W);
    return cw >= 1 && cw <= (kb_mode == TRON_KB_EXACT ? 4 : 3) && p.tile_order && p.in_p == 1 && p.n % kDgTile == 0 && nr % kDgTile == 0 && p.n >= 2 * kDgTile && nr >= 2 * kDgTile
           && (long long)p.nro * p.npe * p.nrep < (1ll << 31) && p.nrep >= kDsCoils && p.group_max >= 4 && (p.trig_img_stride == 0 || p.nimg >= 32) && p.nro <= 0x7fff
           && ((size_t)kDsCoils * p.in_c + (size_t)p.n * nr) * sizeof(float2) < (1ull << 32);
}

hipError_t launch_degrid_stream(const DegridParams &p, int kb_mode, hipStream_t s)
{
    const int cw = (int)ceilf(p.W);
    if (p.in_transposed) {
        switch (cw) {
            case 1: return launch_degrid_stream_cw<1, true>(p, kb_mode, s);
            case 2: return launch_degrid_stream_cw<2, true>(p, kb_mode, s);
            case 3: return launch_degrid_stream_cw<3, true>(p, kb_mode, s);
            case 4: return launch_degrid_stream_cw<4, true>(p, kb_mode, s);
        }
    } else {
        switch (cw) {
            case 1: return launch_degrid_stream_cw<1, false>(p, kb_mode, s);
            case 2: return launch_degrid_stream_cw<2, false>(p, kb_mode, s);
            case 3: return launch_degrid_stream_cw<3, false>(p, kb_mode, s);
            case 4: return launch_degrid_stream_cw<4, false>(p, kb_mode, s);
        }
    }
    return hipErrorInvalidValue;
}

__global__ void warm_degrid_stream_tu() {}

hipError_t warm_degrid_stream()   // see warm_kernels() in tron_kernels.hip: this translation unit had none until round 4
{
    hipLaunchKernelGGL(warm_degrid_stream_tu, dim3(1), dim3(64), 0, nullptr);
    return hipGetLastError();
}

}  // namespace tron
