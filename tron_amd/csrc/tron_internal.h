// Internal declarations shared by the HIP kernels and the host side of libtronhip.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>
#include <set>
#include <utility>

#include "tron_host.h"

namespace tron {

// More than 64 KiB of dynamic LDS is an opt-in per kernel AND per device (hipFuncAttributeMaxDynamicSharedMemorySize): set once for
// each (kernel, device) pair -- a `static` once per process would leave the other devices of tron_recon_radial2d_multi without it.
inline hipError_t allow_dynamic_lds(const void *fn, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, dev})) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({fn, dev});
    return e;
}

constexpr int kTile = 16;          // Cartesian tile edge owned by one workgroup (gridding)
constexpr int kGridThreads = 64;   // one wave per tile, each lane owns 2x2 points
constexpr int kGridRecords = 128;  // sample records staged in LDS per batch
constexpr int kKbPolyTerms = 16;   // slots of the fast Kaiser-Bessel polynomial (highest power first, zeros in front of a shorter one)
// Terms the polynomial needs for a peak-relative error below 1e-8 (fp32 noise is 6e-8): 8 up to W = 1 (1e-9), 10 up to W = 2
// (7e-9), 12 up to W = 3 (6e-9), 16 beyond.  The host fits kb_terms(ceil(W)) coefficients into the LAST slots, so a kernel that
// knows ceil(W) at compile time starts its Horner chain at slot kKbPolyTerms - kb_terms(CW), and one that does not runs all
// 16 slots over leading zeros to the same value.
constexpr int kb_terms(int cw) { return cw <= 1 ? 8 : (cw <= 2 ? 10 : (cw <= 3 ? 12 : 16)); }

// Parameters of one gridding launch (adjoint interpolation), see tron_kernels.hip.
struct GridParams {
    const void *nudata;      // k-space, float2 (or half2) [c + nchan*(ro + nro*spoke)]
    float2 *udata;           // Cartesian output
    const float2 *trig;      // (cos, sin) per spoke of the stream / window
    const uint32_t *band;    // Rlo | Rhi<<16 per grid point, centred raster order
    const int *tile_order;   // tile ids, most expensive first
    unsigned int *errflag;   // device word, set non-zero on internal overflow
    long long in_slice_stride;  // complex elements between the windows of consecutive slices
    int trig_slice_stride;      // table entries between consecutive slices (0: same angles every slice)
    int nxos, nro, npe, nchan;
    int tiles_per_row, ntiles, nslices;
    int coil0;               // first coil handled by blockIdx.y == 0 is coil0 (chunks of CPB follow)
    float W, beta, scale, dcf_a, dcf_b;
    int apply_dcf;
    long long out_z, out_c;  // output strides (complex elements) per slice and per coil
    int out_p;               // output stride per pixel
    int out_shift;           // 1: rows/cols stored in FFT-native order ((Y+n)%n), 0: centred (Y+n/2)
    float kb_poly[kKbPolyTerms];  // fast mode: highest power first, includes the 0.5/W factor
    int skip_outside;        // binned kernel: tiles wholly beyond radius nxos/2-1+W (always zero, src/tron.cu:498-502) are not stored;
                             // only set when the consumer (launch_fft512_adjoint) does not read them either
    // binned kernel, split tiles (small launches: the k-space-centre tiles are dealt to several workgroups each)
    int tile_entries;        // entries of tile_order (0: ntiles plain tile ids)
    int nsplit_slots;        // split tiles per slice (0: none); slot s is tile split_slots[s] & 0xffff with (>> 20) & 15 parts
    int max_parts;
    const int *split_slots;
    float2 *partial;         // [slice][slot][part][coil][32 x 32] partial tiles
    // binned kernel, linear angles: `vslices` consecutive slices share one pass (nslices = groups, nslices_total = slices)
    int vslices, nslices_total;
    int inner_r0;            // binned kernel, centre relief (0: off): samples |r| < inner_r0 are gridded by the origin-centred inner tile
                             // (entry tile id == ntiles, dealt over spoke ranges; slot 0 of split_slots), the four centre tiles take |r| >= inner_r0
    // arc kernel (tron_grid_arc.hip): per (window, tile) the run of crossing spokes, dealt into batches at plan time
    const int4 *arc_hdr;               // [window][tile] -> run length, batches, first entry, records
    const uint4 *arc_ent;              // [window][arc_cap] -> first sample index | down << 31, ulo | len << 10 | offset << 17, cos, sin
    const uint32_t *arc_win;           // [window][tile][256] -> per thread of the tile's workgroup its run of the tile's list: first entry | end << 16
    const uint32_t *arc_off;           // scatter kernel: [window][arc_cap] -> record offset of every entry inside its run
    const unsigned char *arc_rec;      //                 [window][arc_rec_cap][80] -> per group of 64 records: 64 member bytes + the group's first member (16 bits)
    const int *arc_rbase;              //                 [window][tile] -> first group of the tile's run in arc_rec
    int arc_rec_cap;                   //                 groups per window
    int scat_tile;                     // scatter kernel: tile edge its tables were made for (32 or 64)
    int arc_cap, arc_nrec;             // entries per window; records per batch the runs were dealt for
    int arc_accumulate;                // the arc kernel adds to the grid instead of storing (passes after the first, npe > kArcMaxNpe)
    int arc_slice_stride;              // windows between consecutive slices: 1 (golden angle) or 0 (every slice has the same angles)
    const float2 *kb_lut;              // [3][kArcLutEntries] Kaiser-Bessel pair table: position t = d lut_scale of the signed distance d from a block's
                                       // first column, entry trunc(t) + lut_bias: c0 + f (c1 + f c2) for that column (.x) and the next (.y); zero from |x| = W on
    int lut_entries, lut_bias;
    float lut_scale;
    int arc_zper;                      // consecutive slices one workgroup grids in turn (the table stays in LDS)
    float scat_wsum;                   // scatter kernel: bound of the window products one spoke adds to one grid point, 1.75 K(0)^2 (its fixed-point scale)
    float scat_wmax;                   //                 ... and of what one SAMPLE adds: K(0)^2 (with a margin of 1e-5)
    // centre kernel (tron_grid_centre.hip): the angle-sorted spoke lists of the arc kernel's plan, kept
    const unsigned short *cen_order;   // [window][npe] window-relative spoke index, ascending line angle (mod pi)
    const uint32_t *cen_win;           // [window][cen_ngroups] the block's run of that list: first entry | entries << 16 (circular)
    const float2 *cen_cs;              // [window][npe] (cos, sin) of that spoke
    const uint4 *cen_grec;             // [cen_ngroups][2] work units of the centre kernel, busiest first: a 2x2 block of the origin-centred 32 x 32 square that
                                       // a sample |r| < inner_r0 reaches, or one of up to four parts of a busy block's window:
                                       // (col | row << 8, largest such |r|, part | parts << 8 | busy-block index << 16, block index), (band of the four points as masks over |r|)
    int cen_nblocks, cen_nheavy;       // blocks (cen_win's row length), blocks worked on in parts
    float *cen_parts;                  // [slice][coil chunk][busy block][4][64] the parts' sums until the last one adds them up
    unsigned cen_magic_zc[2], cen_magic_chunks;    // set by the launcher: division by an XCD's (slice, chunk) pairs / by the coil chunks as a multiplication
    unsigned *cen_ticket;              // [8][16] work counters of the centre kernel, one per XCD, then [slice][coil chunk][busy block] parts done (zeroed by its launcher)
    int cen_ngroups;
};

struct PostParams {           // crop + deapodise + (optional) root-sum-of-squares, adjoint tail
    const float2 *fft;        // [slice][coil][nxos][nxos], FFT-native order
    float2 *out;              // combine: [slice][nx*nx]; else [slice][nchan*id + c]
    const float *inv_deapod;  // nx*nx
    int nx, nxos, nchan, nslices, combine;
};

struct PreParams {            // pad + deapodise + shift, forward head
    const float2 *img;        // [image][nchan*(row*nx+col) + c]
    float2 *fft;              // [image][coil][nxos][nxos] FFT-native order
    const float *inv_deapod;  // nxos*nxos
    int nx, nxos, nchan, nimg;
    int ny, nyos;             // rows of a non-square image / grid (0: square)
};

struct DegridParams {
    const float2 *udata;      // Cartesian input
    float2 *nudata;           // [image][nrep*(ro + nro*pe) + c]
    const float2 *trig;       // (cos, sin) per spoke
    int trig_img_stride;      // table entries between consecutive images (0: same angles every image)
    const int *tile_order;    // degrid_tile_kernel: 32x32 tiles, centre first (nullptr: raster order)
    long long in_z, in_c;     // input strides per image and per coil
    int in_p, in_shift;       // pixel stride; 1: input is the raw FFT output (second fftshift folded into indexing)
    int in_transposed;        // degrid_tile_kernel: input planes are stored [col][row] (fused forward FFT)
    int in_rot;               // ... with every line rotated: point i of a line lies at (i + in_rot) mod its length (the streaming kernel's halo:
                              // a tile's row segments then start on a 128-byte line, three lines each instead of four)
    int group_end[4];         // degrid_stream_kernel: tile_order positions [group_end[c-1], group_end[c]) take runs of 2^c images (c = 4: the rest)
    int group_max;            // ... capped by this (a quarter of the launch's images at most); < 4: degrid_tile_kernel only (8 images of 8 coils: 2.07 vs 1.79 us per coil image there)
    int n, nrep, nro, npe, nimg;
    int nrows;                // simple kernel only: rows of a non-square grid (0: n); n is then the column count
    float W, beta;
    float kb_poly[kKbPolyTerms];
};

// launchers (tron_kernels.hip); kb_mode: TRON_KB_EXACT / TRON_KB_FAST; half_in: nudata is half2
hipError_t launch_grid(const GridParams &p, int kb_mode, int half_in, hipStream_t s);
// TRON_KB_FAST only; p.tile_order must list 32x32 tiles (tron_grid_binned.hip)
hipError_t launch_grid_binned(const GridParams &p, int half_in, hipStream_t s);
constexpr int kBinnedTile = 32;
// TRON_KB_FAST, fp32 input, even coil counts, centre relief active (tron_grid_arc.hip): the plain tiles of p.tile_order
// (entries [first_plain, first_plain + ntiles)) are gridded by the arc kernel, which leaves the samples |r| < p.inner_r0 to launch_grid_centre
hipError_t launch_grid_arc(const GridParams &p, int half_in, int first_plain, hipStream_t s);
// the samples |r| < p.inner_r0 ADDED to the grid the arc kernel has stored (tron_grid_centre.hip): same stream, behind launch_grid_arc
hipError_t launch_grid_centre(const GridParams &p, int half_in, hipStream_t s);
hipError_t warm_grid_centre();
constexpr int kArcMaxNpe = 1024;       // spokes of one window the arc kernel's run tables hold (arc_prep_kernel: thread = spoke, four per thread)
constexpr int kArcPassNpe = 812;       // ... but a window of more than this many is gridded in PASSES of at most this many, the later ones adding to the grid: a centre
                                       // tile's run holds 0.63 of a pass's spokes (its quadrant of directions + 2 asin(W sqrt(2) / 14)) and has 512 entries.  (Until round 6
                                       // passes began at 1 025 spokes, and windows of 813 .. 1 024 overflowed their run tables: binned kernel.)
constexpr int kArcMaxWindow = 4096;    // most spokes of one window the arc / scatter kernels take (grid_arc_supported)
constexpr int kArcMaxPasses = 6;       // windows of more spokes are gridded in passes of <= kArcMaxNpe, the later ones adding to the grid
bool grid_arc_supported(int nchan, int nxos, int nro, int npe, float W, int half_in);
int grid_arc_nrec(int nchan, int half_in);
constexpr int kArcLutEntries = 400;    // Kaiser-Bessel pair-table entries held in LDS (build_kb_pair_lut: (2 W + 1) s + 4 of them, s a power of two, W > 1)
// plan-time pass of the arc kernel: clips every window's angle-sorted spokes against every tile and deals the runs into batches
struct ArcPrepParams {
    const unsigned short *order;       // [window][npe] window-relative spoke index, ascending line angle (mod pi)
    const float *phi;                  // [window][npe] that line angle in [0, pi)
    const float2 *cs;                  // [window][npe] (cos, sin) of that spoke
    int4 *hdr;                         // out, see GridParams::arc_hdr
    uint4 *ent;
    uint32_t *win;                     // out, see GridParams::arc_win
    const uint32_t *band;              // Rlo | Rhi << 16 per grid point (build_band_table)
    int *alloc;                        // [window] entries handed out so far (zeroed by the caller)
    unsigned int *errflag;
    int nxos, nro, npe, ntiles, inner_r0, nrec, cap;
    float W;
    int flat;                          // tables for grid_scatter_kernel: one batch per run (nrec >= 32767), hdr.y = the longest block window of the tile
    int tile;                          // flat tables: tile edge, 32 or 64 (0 = 32)
    uint32_t *off;                     // flat tables: [window][cap] every entry's record offset inside its run (the entry's own field holds its low 15 bits)
    unsigned char *rec;                // flat tables: [window][rec_cap][80] per GROUP of 64 records of a run: 64 x (member - the member of the group's first
                                       //              record), that member (16 bits), padding
    int *rbase;                        //              [window][tile] the run's first group in rec
    int *ralloc;                       //              [window] groups handed out so far (zeroed by the caller)
    int rec_cap;                       //              groups per window
};
hipError_t launch_arc_prep(const ArcPrepParams &p, int nwindows, hipStream_t s);
// The lists arc_prep_kernel reads, built on the device from the (cos, sin) table (tron_traj_dev.hip): every window's spokes in ascending
// line angle, ties in acquisition order; windows of more than kArcMaxNpe spokes also per pass of `sub` spokes.
struct TrajSortParams {
    const float2 *trig;                // (cos, sin) per spoke; window w = entries [w win_stride, w win_stride + npe)
    int win_stride, nwin, npe, npass, sub;
    unsigned short *order;             // out, [window][npe]: see ArcPrepParams
    float *phi;
    float2 *cs;
    unsigned short *order_q;           // out (npass > 1), [pass][window][spokes of the pass]
    float *phi_q;
    float2 *cs_q;
};
hipError_t launch_traj_sort(const TrajSortParams &p, hipStream_t s);
// ... and the centre kernel's block windows (GridParams::cen_win) from the sorted line angles
struct TrajCentreParams {
    const float *phi;                  // [window][npe] ascending line angles
    const float4 *gwin;                // [block] angular window of the block's run: (lo, hi, all | wrap << 1, 0), build_centre_group_windows
    uint32_t *out;                     // [window][block] first entry | entries << 16
    int nwin, npe, ngroups;
};
hipError_t launch_traj_centre_windows(const TrajCentreParams &p, hipStream_t s);
hipError_t warm_traj();
// one or two channels, W <= 2 (tron_grid_scatter.hip): lane = sample, 64-bit fixed-point sums in LDS; same tables (ArcPrepParams::flat), same call as launch_grid_arc
bool grid_scatter_supported(int nchan, int nxos, int nro, int npe, float W, int half_in);
hipError_t launch_grid_scatter(const GridParams &p, int half_in, int first_plain, hipStream_t s);
hipError_t warm_grid_scatter();
hipError_t warm_grid_arc();
hipError_t launch_post(const PostParams &p, hipStream_t s);
hipError_t launch_pre(const PreParams &p, hipStream_t s);
hipError_t launch_precompensate(float2 *nudata, int nchan, int nro, int npe, float a, float b, hipStream_t s);
hipError_t launch_degrid(const DegridParams &p, int kb_mode, hipStream_t s);
size_t grid_lds_bytes(int cpb, int cw);
hipError_t warm_kernels();       // force-load the code object of tron_kernels.hip
hipError_t launch_clock_probe(unsigned long long *d_out4, hipStream_t s);   // shader cycles, 100 MHz ticks of a short spin (tron_plan_shader_clock)
hipError_t warm_grid_binned();   // ... and of tron_grid_binned.hip
hipError_t warm_fft512();        // ... and of tron_fft512.hip
hipError_t warm_degrid_tile();   // ... and of tron_degrid_tile.hip
hipError_t warm_degrid_stream(); // ... and of tron_degrid_stream.hip
hipError_t warm_cgnr();          // ... and of tron_cgnr.hip
// CGNR vector kernels (tron_cgnr.hip), batched over slices; per-slice scalars live on the device
hipError_t launch_cg_scale_norm2(float2 *x, size_t n, int nslices, float scale, double *partial, hipStream_t s);
hipError_t launch_cg_wnorm2(const float2 *v, size_t n, int nslices, int nchan, int nro, float a, float b, double *partial, int parts_cap,
                            int *nparts, hipStream_t s);
hipError_t launch_cg_windows(float2 *r, const float2 *y, size_t n, size_t hop, int nslices, hipStream_t s);
hipError_t launch_cg_finish(const double *partial, int nparts, double *num, float *coef, int mode, int nslices, hipStream_t s);
hipError_t launch_cg_axpy(float2 *y, const float2 *x, const float *coef, float sign, size_t n, int nslices, hipStream_t s);
hipError_t launch_cg_update(float2 *x, float2 *pt, const float2 *zt, const float *alpha, const float *beta, size_t n, int nslices, int last, hipStream_t s);
hipError_t launch_coil_combine(float2 *out, const float2 *coil, int nimg, int nc, int nt, int mode, int npatch, int nslices, hipStream_t s);
constexpr int kCgPartials = 64;  // = kCgBlocks
// tiled degridding (tron_degrid_tile.hip), W <= 4
hipError_t launch_degrid_tile(const DegridParams &p, int kb_mode, hipStream_t s);
// the same for launches of many images: one workgroup walks a run of images of its tile, the next tile buffer arriving by
// LDS-DMA while the current one is sampled (tron_degrid_stream.hip); launch only what degrid_stream_supported accepts
bool degrid_stream_supported(const DegridParams &p, int kb_mode);   // fast weights: W <= 2 (wider unrolled weight sets spill at 128 registers)
hipError_t launch_degrid_stream(const DegridParams &p, int kb_mode, hipStream_t s);
// fused pruned inverse FFT + crop + deapodise + SoS for nxos = 512, nx = 256 (tron_fft512.hip)
// rzero: grid points at integer radius > rzero hold zeros by construction and are not read (0 = read everything)
hipError_t launch_fft512_adjoint(const float2 *grid, float2 *tmp, float2 *out, const float2 *tw, const float *inv_deapod, int rzero,
                                 int nchan, int nslices, hipStream_t s);
// the same without coil combination: out[slice][nchan * (row * 256 + col) + c] = coil images * scale; partial (may be null):
// fft512_coils_partials(nslices) * nchan sums of |out|^2 per slice (one per workgroup; summed in index order by cg_finish)
hipError_t launch_fft512_adjoint_coils(const float2 *grid, float2 *tmp, float2 *out, const float2 *tw, const float *inv_deapod, int rzero,
                                       int nchan, int nslices, float scale, double *partial, hipStream_t s);
int fft512_coils_partials(int nslices);
// fused pad + deapodise + shift + pruned forward FFT for nx = 256, nxos = 512 (tron_fft512.hip)
// rzero: grid points at centred radius > rzero are not stored (no degridded sample's footprint reaches them; 0 = store all)
hipError_t launch_fft512_forward(const float2 *img, float2 *tmp, float2 *out, const float2 *tw, const float *inv_deapod, int rzero, int rot,
                                 int nchan, int nimg, hipStream_t s);

}  // namespace tron
