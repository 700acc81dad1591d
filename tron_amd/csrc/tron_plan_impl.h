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

struct tron_plan {
    tron_config cfg;
    tron_dims d;
    int nchan = 0;
    int share_z0 = 0, share_nz = 0;   // the slices this plan will be asked for (a per-GPU worker's block; the whole volume otherwise)
    int kb_mode = TRON_KB_EXACT;
    int chunk = 1;                 // slices (adjoint) or images (forward) per batch
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // FFT lane of the adjoint pipeline (gridding stays on `stream`)
    hipEvent_t ev_g[2] = {nullptr, nullptr}, ev_f[2] = {nullptr, nullptr};   // grid done / buffer free, per buffer
    bool dual = false;
    bool fft_pending[2] = {false, false};   // two-lane pipeline: an FFT launch that reads work buffer b may still be in flight (ev_f[b])
    // device tables
    float2 *d_trig = nullptr;
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
    float scat_wsum = 0;
    int scat_tile = 32;                   // ... on 32 x 32 or 64 x 64 tiles (its run tables are made for one of them)
    uint32_t *d_arc_off = nullptr;        // scatter kernel: record offset of every run entry
    unsigned char *d_arc_rec = nullptr;   //                 per group of 64 records of every run: 80 bytes (arc_prep_kernel)
    int *d_arc_rbase = nullptr;           //                 [window][tile] first record of the run
    int arc_rec_cap = 0;
    int *d_tile_order64 = nullptr;        // 64-tiles, centre first
    int4 *d_arc_hdr = nullptr;
    uint4 *d_arc_ent = nullptr;
    uint32_t *d_arc_win = nullptr;
    float2 *d_kb_lut = nullptr;
    // ... and the k-space centre's kernel (tron_grid_centre.hip): the sorted spoke lists, the block groups
    unsigned short *d_cen_order = nullptr;
    uint32_t *d_cen_win = nullptr;
    float2 *d_cen_cs = nullptr;
    uint4 *d_cen_grec = nullptr;
    unsigned *d_cen_ticket = nullptr;
    float *d_cen_parts = nullptr;
    uint4 *d_cen_grec_parts = nullptr;          // the work units with the busy blocks in parts (launches of fewer than cen_parts_below slices)
    int cen_nblocks = 0, cen_nheavy = 0, cen_nunits_parts = 0, cen_parts_below = 64;
    int cen_ngroups = 0;
    bool centre_kernel = true;            // TRON_CENTRE_KERNEL=binned (A/B): the inner tile on the binned kernel + grid_reduce_parts_kernel, as in round 3
    int arc_cap = 0, arc_nrec = 0, arc_zper = 1;
    int arc_passes = 1, arc_pass_npe = 0;       // windows of more than kArcMaxNpe spokes: passes over arc_pass_npe spokes each (tron_plan.cpp)
    size_t arc_nwin = 0;                        // windows the run tables hold (per pass)
    float lut_scale = 0;
    int lut_entries = 0, lut_bias = 0;
    double lut_err = 0;
    hipStream_t stream_inner = nullptr;   // the inner tile's parts (binned kernel) run beside the arc kernel
    hipEvent_t ev_inner[2] = {nullptr, nullptr};
    bool inner_beside = true;             // TRON_ARC_INNER_STREAM=0 (A/B): the inner tile's launch in front of the arc kernel, same stream
    float *d_deapod = nullptr;
    unsigned int *d_errflag = nullptr;
    int ntiles = 0, tiles_per_row = 0;
    // kernel constants
    float beta = 0, dcf_a = 0, dcf_b = 0, scale = 0;
    double kb_poly_err = 0;        // max relative error of the fast Kaiser-Bessel polynomial
    float kb_poly[tron::kKbPolyTerms];
    // work buffers
    float2 *d_grid = nullptr;      // chunk * nchan * nxos^2
    float2 *d_grid2 = nullptr;     // second Cartesian buffer (dual-stream pipeline)
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
    int grid_lds_pad = 0;          // TRON_GRID_LDS_PAD (two-lane experiments): LDS request of the binned gridding kernel
    bool slices_per_pass = true;   // TRON_SLICES_PER_PASS=0 turns the linear-angle slice grouping off (A/B, tests)
    bool poison = false;           // TRON_POISON_GRID (tests): NaN-fill the work grid
    double create_s[5] = {0, 0, 0, 0, 0};   // tron_plan_create_times
    int debug_skip = 0;            // environment knobs, read once at plan creation (never on the launch path)
    bool degrid_simple = false, degrid_tile_only = false, no_disc = false;
    const char *last_degrid_kernel = "";   // tron_plan_degrid_kernel_name
    bool pin_host = false;         // hipHostRegister the caller's buffers in tron_recon_radial2d[_range]
    bool timing = false;
    bool sync_each = false;        // TRON_SYNC_EACH=1: synchronise after every launch and name the failing stage
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
                bool defer_join = false, float out_scale = 1.f, double *norm_partial = nullptr, int *norm_parts = nullptr);
int forward_run(tron_plan *p, void *d_out, const void *d_in, int nimg, const float2 *trig = nullptr, int trig_img_stride = 0,
                const float *deapod = nullptr);
int cgnr_run(tron_plan *p, void *d_out, const void *d_in_z0, int zfirst, int zcount, int combine);

}  // namespace tron
