// Private declarations shared by the host-side translation units of libtronhip: the plan object, the error macros and
// the pipeline functions (tron_plan.cpp: life cycle + device-resident C ABI; tron_pipeline.cpp: batched adjoint / forward /
// CGNR pipelines; tron_hostio.cpp: host-buffer entry points and in-process multi-GPU).
#pragma once

#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <set>
#include <tuple>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/tron_hip.h"
#include "tron_host.h"
#include "tron_internal.h"

namespace tron {

extern std::once_flag g_fft_once;

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(TRON_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define FFT_TRY(expr)                                                                           \
    do {                                                                                        \
        rocfft_status s_ = (expr);                                                              \
        if (s_ != rocfft_status_success)                                                        \
            return fail(TRON_ERR_FFT, "%s failed: rocfft_status %d (%s:%d)", #expr, (int)s_, __FILE__, __LINE__); \
    } while (0)

enum { STAGE_GRID = 0, STAGE_FFT = 1, STAGE_POST = 2, STAGE_PRE = 3, STAGE_DEGRID = 4, STAGE_COUNT = 5 };

struct FftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
    size_t work_bytes = 0;
};

}  // namespace tron

// Everything that depends on the spoke ANGLES (src/tron.cu:509-511, 555-559: a function of skip_angles): the (cos, sin) table, the
// angle-sorted spoke lists, the centre kernel's block windows and the arc / scatter kernels' run tables.  A plan holds two sets:
// `cur` is the one launches read, tron_plan_retarget builds the other on a stream of its own while the launches already queued
// run on (tron_traj.cpp, tron_traj_dev.hip).
struct TrajTables {
    bool allocated = false;
    bool ok = true;                       // run tables usable (false: they overflowed -- the binned kernel takes every tile of this set)
    int skip_angles = 0;
    float *h_trig = nullptr;              // pinned host: (cos, sin) by libm, as the reference computes them
    float2 *d_trig = nullptr;
    unsigned short *d_order = nullptr;    // [window][npe] window-relative spoke index, ascending line angle (mod pi); kept: centre kernel
    float *d_phi = nullptr;               // [window][npe] that line angle
    float2 *d_cs = nullptr;               // [window][npe] (cos, sin) of that spoke
    unsigned short *d_order_q = nullptr;  // windows of more than kArcMaxNpe spokes: the same lists per pass, [pass][window][spokes of the pass]
    float *d_phi_q = nullptr;
    float2 *d_cs_q = nullptr;
    uint32_t *d_cen_win = nullptr;        // [window][block] the block's run of the sorted list (centre kernel)
    int4 *d_arc_hdr = nullptr;
    uint4 *d_arc_ent = nullptr;
    uint32_t *d_arc_win = nullptr;
    uint32_t *d_arc_off = nullptr;        // scatter kernel: record offset of every run entry
    unsigned char *d_arc_rec = nullptr;   //                 per group of 64 records of every run: 80 bytes (arc_prep_kernel)
    int *d_arc_rbase = nullptr;           //                 [window][tile] first record of the run
    int *d_alloc = nullptr;               // [pass][2][window] arc_prep_kernel's allocation counters
    unsigned int *d_flag = nullptr;       // arc_prep_kernel's overflow flags of this set
    unsigned int *h_flag = nullptr;       // ... copied here (pinned) behind the build
    hipEvent_t ev_built = nullptr;        // the build of this set (on the stream it was queued on)
    hipEvent_t ev_released = nullptr;     // recorded on the plan's stream when the plan turned away from this set: its last reader
    bool released = false;
};

struct tron_plan {
    tron_config cfg;
    tron_dims d;
    int nchan = 0;
    int share_z0 = 0, share_nz = 0;   // the slices this plan will be asked for (a per-GPU worker's block; the whole volume otherwise)
    int kb_mode = TRON_KB_EXACT;
    int chunk = 1;                 // slices (adjoint) or images (forward) per batch
    hipStream_t stream = nullptr;
    hipStream_t stream_build = nullptr;   // tron_plan_retarget: the next table set is built here, beside the launches on `stream`
    TrajTables traj[2];
    int cur = 0;                   // the set launches read
    bool retarget_pending = false; // traj[1 - cur] is being built: the next launch waits for it and turns to it
    // device tables
    size_t ntrig = 0;
    uint32_t *d_band = nullptr;
    int *d_tile_order = nullptr;
    int *d_tile_order32 = nullptr;   // 32x32 tiles of the binned (fast) gridding kernel
    int dg_group_end[4] = {0, 0, 0, 0};   // forward: run-length classes of the streaming degridding kernel over that order
    bool binned = false;
    // small launches of the binned kernel: heavy (k-space-centre) tiles dealt to several workgroups each
    int *d_tile_order32_split = nullptr, *d_split_slots = nullptr;
    int split_entries = 0, nsplit_slots = 0, max_parts = 0, split_below = 0;
    float2 *d_partial = nullptr;
    // centre relief of the binned gridding kernel (GridParams::inner_r0): entry list with the inner tile's parts, its slot
    int *d_tile_order32_relief = nullptr, *d_relief_slots = nullptr;
    int relief_entries = 0, relief_parts = 0, relief_r0 = 0;
    // the same with the inner tile dealt to more workgroups, for launches of fewer than 32 slices (its serial chain bounds them)
    int *d_tile_order32_relief_small = nullptr, *d_relief_slots_small = nullptr;
    int relief_entries_small = 0, relief_parts_small = 0;
    float2 *d_relief_partial = nullptr;
    size_t relief_slices = 0;
    size_t partial_slices = 0;
    // arc gridding kernel (tron_grid_arc.hip): per (window, tile) run tables built at plan creation, Kaiser-Bessel table
    bool arc = false;
    bool scatter = false;                 // ... gridded by grid_scatter_kernel (one or two channels, tron_grid_scatter.hip): same tables, one batch per run
    float scat_wsum = 0, scat_wmax = 0;
    int scat_tile = 32;                   // ... on 32 x 32 or 64 x 64 tiles (its run tables are made for one of them)
    int scat_tile_max = 64;               // ... 32 once 64-tile tables have overflowed (tron_plan_create's retry)
    int relief_r0_binned = 0;             // the binned kernel's own inner radius (scatter plans lower relief_r0: restored when they fall back)
    int arc_rec_cap = 0;                  // scatter kernel: groups of 64 records per window its member tables hold
    int *d_tile_order64 = nullptr;        // 64-tiles, centre first
    float2 *d_kb_lut = nullptr;
    // ... and the k-space centre's kernel (tron_grid_centre.hip): the block groups (the sorted spoke lists are TrajTables')
    float4 *d_cen_gwin = nullptr;         // per block: the angular window of its run (lo, hi, all | wrap << 1), tron_traj_dev.hip
    uint4 *d_cen_grec = nullptr;
    unsigned *d_cen_ticket = nullptr;
    float *d_cen_parts = nullptr;
    uint4 *d_cen_grec_parts = nullptr;          // the work units with the busy blocks in parts (launches of fewer than cen_parts_below slices)
    int cen_nblocks = 0, cen_nheavy = 0, cen_nunits_parts = 0, cen_parts_below = 64;
    int cen_ngroups = 0;
    int arc_cap = 0, arc_nrec = 0, arc_zper = 0;
    int arc_passes = 1, arc_pass_npe = 0;       // windows of more than kArcMaxNpe spokes: passes over arc_pass_npe spokes each (tron_plan.cpp)
    size_t arc_nwin = 0;                        // windows the run tables hold (per pass)
    size_t arc_ntiles = 0;                      // tiles per window of the run tables (32-tiles; 64-tiles for the scatter kernel's large tiles)
    float lut_scale = 0;
    int lut_entries = 0, lut_bias = 0;
    double lut_err = 0;
    float *d_deapod = nullptr;
    unsigned int *d_errflag = nullptr;
    int ntiles = 0, tiles_per_row = 0;
    // kernel constants
    float beta = 0, dcf_a = 0, dcf_b = 0, scale = 0;
    double kb_poly_err = 0;        // max relative error of the fast Kaiser-Bessel polynomial
    float kb_poly[tron::kKbPolyTerms];
    // work buffers
    float2 *d_grid = nullptr;      // chunk * nchan * nxos^2
    void *d_stage_in = nullptr;    // host-API staging
    size_t stage_in_bytes = 0;
    void *d_stage_out = nullptr;
    size_t stage_out_bytes = 0;
    // CGNR (niter > 0): the forward operator's tables and the iteration's vectors, batched over the slices of a chunk
    float *d_deapod_fwd = nullptr;   // 1/w, n = nxos, sigma = 1 (src/tron.cu:643)
    float2 *d_trig_fwd = nullptr;    // linear angles in the degridding kernel's own convention (src/tron.cu:555); null: share d_trig
    float2 *d_cg_r = nullptr, *d_cg_v = nullptr, *d_cg_zt = nullptr, *d_cg_pt = nullptr, *d_cg_x = nullptr;
    double *d_cg_partial = nullptr, *d_cg_num = nullptr;
    float *d_cg_coef = nullptr;      // [2][cg_slices]: alpha, beta
    int cg_parts = 0;                // partial sums per slice the buffer holds
    int cg_slices = 0;
    float2 *d_coil_tmp = nullptr;    // uncombined coil images of a batch (Walsh combination, nt > 1)
    int coil_tmp_slices = 0;
    hipStream_t stream_up = nullptr, stream_down = nullptr;   // host-buffer entry point: upload / download lanes
    std::vector<hipEvent_t> ev_pipe;                          // its chunk events (created on demand, reused)
    float2 *d_trig_tmp = nullptr;  // stage-level gridding calls
    int chunk_cap = 0;             // most slices / images one batch may hold (1.5 x chunk for the adjoint)
    int work_units = 0;            // slices / images the work buffers hold NOW (forward plans grow them on demand)
    bool fft512 = false;           // fused pruned FFT path (nxos 512 -> nx 256)
    float2 *d_tw512 = nullptr;     // exp(+2 pi i k / 512)
    float2 *d_fft_tmp = nullptr;   // chunk * nchan * 256 * 512
    std::map<std::pair<int, int>, tron::FftPlan> fft;   // (batch, direction) -> plan
    // timing
    bool slices_per_pass = true;   // TRON_SLICES_PER_PASS=0 turns the linear-angle slice grouping off (A/B, tests)
    bool poison = false;           // TRON_DEBUG=poison (tests): NaN-fill the work grid
    double create_s[5] = {0, 0, 0, 0, 0};   // tron_plan_create_times
    double retarget_s[2] = {0, 0};          // the last tron_plan_retarget: host seconds of the call, of which the (cos, sin) table
    bool degrid_simple = false, degrid_tile_only = false;    // environment knobs, read once at plan creation (never on the launch path)
    const char *last_degrid_kernel = "";   // tron_plan_degrid_kernel_name
    bool pin_host = false;         // hipHostRegister the caller's buffers in tron_recon_radial2d[_range]
    bool timing = false;
    bool sync_each = false;        // TRON_DEBUG=sync: synchronise after every launch and name the failing stage
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[tron::STAGE_COUNT];
    double ms_acc[tron::STAGE_COUNT] = {0, 0, 0, 0, 0};
    uint64_t launches[tron::STAGE_COUNT] = {0, 0, 0, 0, 0};
};

namespace tron {

struct StageTimer {
    tron_plan *p;
    int stage;
    hipStream_t st_;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    StageTimer(tron_plan *plan, int st, hipStream_t stream = nullptr) : p(plan), stage(st), st_(stream ? stream : plan->stream)
    {
        if (p->timing) {
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, st_);
        }
    }
    ~StageTimer()
    {
        if (p->timing) {
            hipEventRecord(e1, st_);
            p->ev[stage].push_back({e0, e1});
        }
    }
};
template <typename T>
int upload(T **dptr, const void *host, size_t bytes)
{
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(dptr), bytes ? bytes : 1));
    if (bytes) HIP_TRY(hipMemcpy(*dptr, host, bytes, hipMemcpyHostToDevice));
    return TRON_OK;
}

// tron_plan.cpp: a plan that will only ever run slices [z0, z0 + zcount) (tron_recon_radial2d_multi's per-GPU workers):
// batch sizes, work buffers and the arc kernel's run tables are sized for that block
int plan_create_share(tron_plan **out, const tron_config *cfg, const tron_dims *dims, int z0, int zcount);

// tron_traj.cpp: the angle-dependent tables (TrajTables).  traj_build queues the whole build of set T for `skip_angles` on `st`
// (host: the (cos, sin) table by libm on a few threads; device: sort by line angle, centre windows, run tables) and records
// T.ev_built; traj_finish waits for it and reads the overflow flags into T.ok.
int traj_alloc(tron_plan *p, TrajTables &T);
int traj_build(tron_plan *p, TrajTables &T, int skip_angles, hipStream_t st);
int traj_finish(tron_plan *p, TrajTables &T);
void traj_free(TrajTables &T);
int traj_turn(tron_plan *p);       // a pending retarget: wait for the other set and make it current (every launch path calls this first)
inline const TrajTables &traj_cur(const tron_plan *p) { return p->traj[p->cur]; }
inline bool arc_ready(const tron_plan *p) { return p->arc && p->traj[p->cur].ok; }

// tron_pipeline.cpp
int drain_timers(tron_plan *p);
int get_fft(tron_plan *p, int batch, int inverse, FftPlan **out);
int run_fft(tron_plan *p, float2 *buf, int batch, int inverse);
int stage_check(tron_plan *p, const char *what);
void fill_grid_consts(const tron_plan *p, GridParams &g);
int ensure_work(tron_plan *p, int units);
int ensure_buffer(void **buf, size_t *have, size_t want);
int check_errflag(tron_plan *p);
int combine_coils(tron_plan *p, float2 *d_out, const float2 *d_coil, int cz);
// out_scale / norm_partial: uncombined output only (combine = 0): the images times out_scale and, on the fused 512 / 256 path,
// fft512_coils_partials(batch) * nchan partial sums of |image|^2 per slice (CGNR); norm_partial_done tells whether they were written
int adjoint_run(tron_plan *p, void *d_out, const void *d_in_z0, int zfirst, int zcount, int combine, int in_stride_spokes = 0,
                float out_scale = 1.f, double *norm_partial = nullptr, int *norm_parts = nullptr);
int forward_run(tron_plan *p, void *d_out, const void *d_in, int nimg, const float2 *trig = nullptr, int trig_img_stride = 0,
                const float *deapod = nullptr);
int cgnr_run(tron_plan *p, void *d_out, const void *d_in_z0, int zfirst, int zcount, int combine);

}  // namespace tron
