// Host orchestration of libtronhip: plan life cycle, the batched adjoint / forward pipelines and
// the C ABI of include/tron_hip.h.  Replaces src/tron.cu:579-649 and :726-786 of the reference
// (tron_init / tron_shutdown / tron_nufft_adj_radial2d / tron_nufft_radial2d / recon_radial2d).
//
// Differences of structure (not of results) from the reference:
//   * all slices of a run are batched: one gridding launch and one fused FFT + tail (two kernels at the
//     512 -> 256 size, batched rocFFT + one tail kernel otherwise) per chunk of slices, instead of
//     8 kernels + 2 copies per slice on alternating streams;
//   * the spoke stream is uploaded ONCE and sliding windows (src/tron.cu:738-739) are views into
//     it; the reference re-uploads every window (each spoke ~10x for the whole-body run);
//   * the Cartesian data is coil-planar and stored in FFT-native order, so both fftshift passes,
//     crop and the density pre-compensation pass disappear into index arithmetic.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/tron_hip.h"
#include "tron_host.h"
#include "tron_internal.h"

namespace tron {

static thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

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

using namespace tron;

struct tron_plan {
    tron_config cfg;
    tron_dims d;
    int nchan = 0;
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
    bool binned = false;
    // small launches of the binned kernel: heavy (k-space-centre) tiles dealt to several workgroups each
    int *d_tile_order32_split = nullptr, *d_split_slots = nullptr;
    int split_entries = 0, nsplit_slots = 0, max_parts = 0, split_below = 0;
    float2 *d_partial = nullptr;
    size_t partial_slices = 0;
    float *d_deapod = nullptr;
    unsigned int *d_errflag = nullptr;
    int ntiles = 0, tiles_per_row = 0;
    // kernel constants
    float beta = 0, dcf_a = 0, dcf_b = 0, scale = 0;
    double kb_poly_err = 0;        // max relative error of the fast Kaiser-Bessel polynomial
    float kb_poly[kKbPolyTerms];
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
    float *d_cg_coef = nullptr;
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
    std::map<std::pair<int, int>, FftPlan> fft;   // (batch, direction) -> plan
    // timing
    int grid_lds_pad = 0;          // TRON_GRID_LDS_PAD (two-lane experiments): LDS request of the binned gridding kernel
    bool slices_per_pass = true;   // TRON_SLICES_PER_PASS=0 turns the linear-angle slice grouping off (A/B, tests)
    bool poison = false;           // TRON_POISON_GRID (tests): NaN-fill the work grid
    int debug_skip = 0;            // environment knobs, read once at plan creation (never on the launch path)
    bool degrid_simple = false, no_disc = false;
    bool pin_host = false;         // hipHostRegister the caller's buffers in tron_recon_radial2d[_range]
    bool timing = false;
    bool sync_each = false;        // TRON_SYNC_EACH=1: synchronise after every launch and name the failing stage
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[STAGE_COUNT];
    double ms_acc[STAGE_COUNT] = {0, 0, 0, 0, 0};
    uint64_t launches[STAGE_COUNT] = {0, 0, 0, 0, 0};
};

namespace {

std::once_flag g_fft_once;

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

int drain_timers(tron_plan *p)
{
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->stream2) HIP_TRY(hipStreamSynchronize(p->stream2));
    for (int s = 0; s < STAGE_COUNT; ++s) {
        for (auto &pr : p->ev[s]) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
            p->ms_acc[s] += ms;
            p->launches[s] += 1;
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
        p->ev[s].clear();
    }
    return TRON_OK;
}

int get_fft(tron_plan *p, int batch, int inverse, FftPlan **out)
{
    auto key = std::make_pair(batch, inverse);
    auto it = p->fft.find(key);
    if (it == p->fft.end()) {
        FftPlan f;
        const size_t lengths[2] = {(size_t)p->d.nxos, (size_t)p->d.nyos};      // fastest (columns) first; square except non-square forward plans
        // cufftPlan2d / cufftPlanMany of src/tron.cu:205-220: unnormalised C2C; CUFFT_INVERSE (+i)
        // for the adjoint (:632), CUFFT_FORWARD (-i) for the forward transform (:645)
        FFT_TRY(rocfft_plan_create(&f.plan, rocfft_placement_inplace,
                                   inverse ? rocfft_transform_type_complex_inverse : rocfft_transform_type_complex_forward,
                                   rocfft_precision_single, 2, lengths, (size_t)batch, nullptr));
        FFT_TRY(rocfft_execution_info_create(&f.info));
        FFT_TRY(rocfft_execution_info_set_stream(f.info, p->stream));
        FFT_TRY(rocfft_plan_get_work_buffer_size(f.plan, &f.work_bytes));
        if (f.work_bytes) {
            HIP_TRY(hipMalloc(&f.work, f.work_bytes));
            FFT_TRY(rocfft_execution_info_set_work_buffer(f.info, f.work, f.work_bytes));
        }
        // rocFFT builds its twiddle tables with a kernel on a stream of its own; make sure that has
        // finished before the first execution on ours (a non-blocking stream does not wait for it)
        HIP_TRY(hipDeviceSynchronize());
        it = p->fft.emplace(key, f).first;
    }
    *out = &it->second;
    return TRON_OK;
}

int run_fft(tron_plan *p, float2 *buf, int batch, int inverse)
{
    FftPlan *f = nullptr;
    int rc = get_fft(p, batch, inverse, &f);
    if (rc) return rc;
    StageTimer t(p, STAGE_FFT);
    void *bufs[1] = {buf};
    FFT_TRY(rocfft_execute(f->plan, bufs, nullptr, f->info));
    return TRON_OK;
}

template <typename T>
int upload(T **dptr, const void *host, size_t bytes)
{
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(dptr), bytes ? bytes : 1));
    if (bytes) HIP_TRY(hipMemcpy(*dptr, host, bytes, hipMemcpyHostToDevice));
    return TRON_OK;
}

int stage_check(tron_plan *p, const char *what)
{
    if (!p->sync_each) return TRON_OK;
    hipError_t e = hipStreamSynchronize(p->stream);
    if (e == hipSuccess && p->stream2) e = hipStreamSynchronize(p->stream2);
    if (e != hipSuccess) return fail(TRON_ERR_HIP, "stage '%s' failed: %s", what, hipGetErrorString(e));
    fprintf(stderr, "[tronhip] stage %s ok\n", what);
    return TRON_OK;
}

void fill_grid_consts(const tron_plan *p, GridParams &g)
{
    g.band = p->d_band;
    g.tile_order = p->d_tile_order;
    g.errflag = p->d_errflag;
    g.nxos = p->d.nxos;
    g.nro = p->d.nro;
    g.npe = p->d.npe1work;
    g.nchan = p->nchan;
    g.tiles_per_row = p->tiles_per_row;
    g.ntiles = p->ntiles;
    g.coil0 = 0;
    g.W = p->cfg.kernwidth;
    g.beta = p->beta;
    g.scale = p->scale;
    g.dcf_a = p->dcf_a;
    g.dcf_b = p->dcf_b;
    memcpy(g.kb_poly, p->kb_poly, sizeof(g.kb_poly));
    g.debug = p->debug_skip;
    g.lds_pad = p->grid_lds_pad;
}

// Adjoint for slices [zfirst, zfirst+zcount).  d_in_z0 points at the first spoke of slice
// zfirst's window; d_out at that slice's output.
// Work buffers (Cartesian grid, FFT intermediate) for `units` slices / images per batch.  Adjoint plans allocate their
// full batch at creation; forward plans start empty and grow to what a call actually transforms (the host entry point
// only ever asks for one image: no 1.5 GiB of work space for it).
int ensure_work(tron_plan *p, int units)
{
    if (units <= p->work_units) return TRON_OK;
    const tron_dims &d = p->d;
    const size_t per_unit = (size_t)p->nchan * d.nxos * d.nyos * sizeof(float2);
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->stream2) HIP_TRY(hipStreamSynchronize(p->stream2));
    for (float2 **b : {&p->d_grid, &p->d_grid2, &p->d_fft_tmp})
        if (*b) { HIP_TRY(hipFree(*b)); *b = nullptr; }
    p->work_units = 0;
    if (hipMalloc(reinterpret_cast<void **>(&p->d_grid), (size_t)units * per_unit) != hipSuccess)
        return fail(TRON_ERR_NOMEM, "cannot allocate %zu bytes of Cartesian work space", (size_t)units * per_unit);
    if (p->poison) hipMemset(p->d_grid, 0xff, (size_t)units * per_unit);
    if (p->fft512 && hipMalloc(reinterpret_cast<void **>(&p->d_fft_tmp), (size_t)units * p->nchan * 256 * 512 * sizeof(float2)) != hipSuccess)
        return fail(TRON_ERR_NOMEM, "cannot allocate the FFT intermediate buffer");
    if (p->dual && hipMalloc(reinterpret_cast<void **>(&p->d_grid2), (size_t)units * per_unit) != hipSuccess)
        return fail(TRON_ERR_NOMEM, "cannot allocate the second Cartesian buffer");
    p->work_units = units;
    return TRON_OK;
}

// in_stride_spokes: spokes between the windows of consecutive slices in d_in_z0 (0 = prof_slide: views into the stream)
// defer_join: leave the FFT lane running when the call returns (device-resident entry point: the caller synchronises with
// tron_plan_sync); the next call's first gridding launches then overlap this call's last FFT passes.
int adjoint_run_raw(tron_plan *p, void *d_out, const void *d_in_z0, int zfirst, int zcount, int combine, int in_stride_spokes,
                    bool defer_join)
{
    const tron_dims &d = p->d;
    const size_t n2 = (size_t)d.nxos * d.nxos;
    const size_t elem = p->cfg.input_half ? 4 : 8;
    const int golden = p->cfg.golden_angle;
    // Two lanes: gridding is VALU/LDS bound and leaves HBM idle, the FFT passes are HBM bound and leave the
    // VALUs idle.  All gridding launches go to `stream`, all FFT launches to `stream2`; chunk k's FFT waits
    // for chunk k's gridding, and gridding of chunk k+2 waits until the FFT has released buffer k&1.
    // (worth it only when each lane still gets full-size launches: +4.9 % at 8 coils x 256 slices, +2.9 % at 128, 0 below)
    const bool dual = p->dual && p->fft512 && combine && zcount >= 2 * p->chunk;
    // equal batches: a short last launch would be bound by the centre tile's serial chain (e.g. 128 slices = 64 + 64, not 85 + 43)
    // (the work buffers hold 1.5 x chunk so that the batches can be evened out upwards)
    int nbatch = std::max(1, (zcount + p->chunk / 2) / p->chunk);
    if ((zcount + nbatch - 1) / nbatch > p->chunk_cap) nbatch = (zcount + p->chunk_cap - 1) / p->chunk_cap;
    const int even = (zcount + nbatch - 1) / nbatch;
    const int step = dual ? std::max(1, std::min(even, (zcount + 1) / 2)) : even;
    if (int erc = ensure_work(p, std::min(step, std::max(zcount, 1)))) return erc;
    if (!dual)      // one lane: everything of an earlier two-lane call that may still be running on the FFT lane comes first
        for (int b = 0; b < 2; ++b)
            if (p->fft_pending[b]) {
                HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_f[b], 0));
                p->fft_pending[b] = false;
            }
    int lane_idx = 0;
    for (int z0 = 0; z0 < zcount; z0 += step, ++lane_idx) {
        const int cz = std::min(step, zcount - z0);
        const int b = dual ? (lane_idx & 1) : 0;
        hipStream_t st = p->stream;
        hipStream_t st_fft = dual ? p->stream2 : p->stream;
        float2 *grid_buf = b ? p->d_grid2 : p->d_grid;
        float2 *tmp_buf = p->d_fft_tmp;
        if (dual && p->fft_pending[b]) HIP_TRY(hipStreamWaitEvent(st, p->ev_f[b], 0));   // the FFT (of this or an earlier call) that last read buffer b
        GridParams g;
        memset(&g, 0, sizeof(g));
        fill_grid_consts(p, g);
        const int in_stride = in_stride_spokes > 0 ? in_stride_spokes : d.prof_slide;
        g.nudata = static_cast<const unsigned char *>(d_in_z0) + (size_t)z0 * in_stride * d.nro * p->nchan * elem;
        g.udata = grid_buf;
        g.trig = p->d_trig + (golden ? (size_t)(zfirst + z0) * d.prof_slide : 0);
        g.in_slice_stride = (long long)in_stride * d.nro * p->nchan;
        g.trig_slice_stride = golden ? d.prof_slide : 0;
        g.nslices = cz;
        g.apply_dcf = 1;
        g.out_z = (long long)p->nchan * n2;
        g.out_c = (long long)n2;
        g.out_p = 1;
        g.out_shift = 1;
        // the fused FFT never reads beyond the sampled disc, so the gridding kernel need not store zeros there
        const int rzero = (p->fft512 && combine && !p->no_disc) ? (int)floorf((float)(d.nxos / 2 - 1) + p->cfg.kernwidth) + 1 : 0;
        g.skip_outside = rzero > 0 ? 1 : 0;
        {
            StageTimer t(p, STAGE_GRID, st);
            if (p->binned) {
                g.tile_order = p->d_tile_order32;
                // linear angles: every slice has the same trajectory (src/tron.cu:509 depends on pe only), so with few coils
                // several slices share one pass of the kernel (clipping, weights and the sort paid once per group)
                const int vs = (!golden && cz > 1 && p->nchan <= 4 && p->slices_per_pass) ? std::max(1, 8 / p->nchan) : 1;
                if (vs > 1) {
                    g.vslices = vs;
                    g.nslices_total = cz;
                    g.nslices = (cz + vs - 1) / vs;
                } else if (cz < p->split_below && p->nsplit_slots > 0) {
                    // a launch this small would be bound by the centre tiles' serial chains: split them over spoke ranges
                    if (p->partial_slices < (size_t)cz) {
                        if (p->d_partial) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(p->d_partial)); p->d_partial = nullptr; }
                        const size_t want = (size_t)std::min(p->split_below, std::max(cz, 8));
                        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_partial),
                                          want * p->nsplit_slots * p->max_parts * p->nchan * kBinnedTile * kBinnedTile * sizeof(float2)));
                        p->partial_slices = want;
                    }
                    g.tile_order = p->d_tile_order32_split;
                    g.tile_entries = p->split_entries;
                    g.nsplit_slots = p->nsplit_slots;
                    g.max_parts = p->max_parts;
                    g.split_slots = p->d_split_slots;
                    g.partial = p->d_partial;
                }
                HIP_TRY(launch_grid_binned(g, p->cfg.input_half, st));
            } else {
                HIP_TRY(launch_grid(g, p->kb_mode, p->cfg.input_half, st));
            }
        }
        int rc = stage_check(p, "grid");
        if (rc) return rc;
        if (p->fft512 && combine) {
            // fused: pruned inverse FFT + crop + deapodise + root-sum-of-squares (tron_fft512.hip)
            if (dual) {
                HIP_TRY(hipEventRecord(p->ev_g[b], st));
                HIP_TRY(hipStreamWaitEvent(st_fft, p->ev_g[b], 0));
            }
            {
                StageTimer t(p, STAGE_FFT, st_fft);
                HIP_TRY(launch_fft512_adjoint(grid_buf, tmp_buf, static_cast<float2 *>(d_out) + (size_t)z0 * d.nx * d.nx,
                                              p->d_tw512, p->d_deapod, rzero, p->nchan, cz, st_fft));
            }
            if (dual) {
                HIP_TRY(hipEventRecord(p->ev_f[b], st_fft));
                p->fft_pending[b] = true;
            }
            if ((rc = stage_check(p, "fft512"))) return rc;
            continue;
        }
        rc = run_fft(p, p->d_grid, cz * p->nchan, 1);
        if (rc) return rc;
        if ((rc = stage_check(p, "fft"))) return rc;
        PostParams q;
        q.fft = p->d_grid;
        q.out = static_cast<float2 *>(d_out) + (size_t)z0 * d.nx * d.nx * (combine ? 1 : p->nchan);
        q.inv_deapod = p->d_deapod;
        q.nx = d.nx;
        q.nxos = d.nxos;
        q.nchan = p->nchan;
        q.nslices = cz;
        q.combine = combine;
        if (g.debug != 5) {
            StageTimer t(p, STAGE_POST);
            HIP_TRY(launch_post(q, p->stream));
        }
        if ((rc = stage_check(p, "post"))) return rc;
    }
    if (dual && lane_idx > 0 && !defer_join) {   // later work on the main stream (e.g. the download) waits for the FFT lane
        HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_f[(lane_idx - 1) & 1], 0));
    }
    return TRON_OK;
}

// trig / deapod default to the plan's own tables (forward plans); the CGNR path of an adjoint plan passes the forward
// operator's tables and a per-image angle stride (every slice has its own golden angles)
// coilcombinesos / coilcombinewalsh (src/tron.cu:764,766) of `cz` slices of coil images [z][nchan*id + c]
int combine_coils(tron_plan *p, float2 *d_out, const float2 *d_coil, int cz)
{
    if (p->cfg.coil_combine == 1 && p->d.nc > 16)
        return fail(TRON_ERR_UNSUPPORTED, "Walsh coil combination handles up to 16 coils (nc=%d)", p->d.nc);
    HIP_TRY(launch_coil_combine(d_out, d_coil, p->d.nx, p->d.nc, p->d.nt, p->cfg.coil_combine == 1 ? 1 : 0,
                                std::max(0, p->cfg.walsh_patch), cz, p->stream));
    return TRON_OK;
}

// The adjoint with the plan's coil combination.  Root-sum-of-squares of one repetition is fused into the pipeline's
// tail; Walsh's adaptive combination and nt > 1 (channel = coil + nc*repetition) run it uncombined into a scratch
// buffer and combine from there.
int adjoint_run(tron_plan *p, void *d_out, const void *d_in_z0, int zfirst, int zcount, int combine, int in_stride_spokes = 0,
                bool defer_join = false)
{
    if (!combine || (p->d.nt == 1 && p->cfg.coil_combine != 1))
        return adjoint_run_raw(p, d_out, d_in_z0, zfirst, zcount, combine, in_stride_spokes, defer_join);
    const tron_dims &d = p->d;
    const size_t N = (size_t)p->nchan * d.nx * d.ny, elem = p->cfg.input_half ? 4 : 8;
    const int step = std::max(1, std::min(p->chunk, zcount));
    if (p->coil_tmp_slices < step) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (p->d_coil_tmp) HIP_TRY(hipFree(p->d_coil_tmp));
        p->d_coil_tmp = nullptr; p->coil_tmp_slices = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_coil_tmp), step * N * sizeof(float2)));
        p->coil_tmp_slices = step;
    }
    const int in_stride = in_stride_spokes > 0 ? in_stride_spokes : d.prof_slide;
    for (int z0 = 0; z0 < zcount; z0 += step) {
        const int cz = std::min(step, zcount - z0);
        int rc = adjoint_run_raw(p, p->d_coil_tmp, static_cast<const unsigned char *>(d_in_z0) + (size_t)z0 * in_stride * d.nro * p->nchan * elem,
                                 zfirst + z0, cz, 0, in_stride_spokes, false);
        if (rc) return rc;
        if ((rc = combine_coils(p, static_cast<float2 *>(d_out) + (size_t)z0 * d.nt * d.nx * d.ny, p->d_coil_tmp, cz))) return rc;
    }
    return TRON_OK;
}

int forward_run(tron_plan *p, void *d_out, const void *d_in, int nimg, const float2 *trig = nullptr, int trig_img_stride = 0,
                const float *deapod = nullptr)
{
    const tron_dims &d = p->d;
    const size_t n2 = (size_t)d.nxos * d.nyos;
    const bool square = d.nx == d.ny;
    if (!trig) trig = p->d_trig;
    if (!deapod) deapod = p->d_deapod;
    if (int erc = ensure_work(p, std::max(1, std::min(p->chunk, nimg)))) return erc;
    for (int k0 = 0; k0 < nimg; k0 += p->chunk) {
        const int ck = std::min(p->chunk, nimg - k0);
        const float2 *img = static_cast<const float2 *>(d_in) + (size_t)k0 * p->nchan * d.nx * d.ny;
        if (p->fft512) {
            // fused: pad + deapodise + shift + pruned forward FFT (tron_fft512.hip)
            StageTimer t(p, STAGE_FFT);
            HIP_TRY(launch_fft512_forward(img, p->d_fft_tmp, p->d_grid, p->d_tw512, deapod, p->nchan, ck, p->stream));
        } else {
            PreParams a;
            a.img = img;
            a.fft = p->d_grid;
            a.inv_deapod = deapod;
            a.nx = d.nx;
            a.nxos = d.nxos;
            a.nchan = p->nchan;
            a.nimg = ck;
            a.ny = square ? 0 : d.ny;
            a.nyos = square ? 0 : d.nyos;
            {
                StageTimer t(p, STAGE_PRE);
                HIP_TRY(launch_pre(a, p->stream));
            }
            int rc = run_fft(p, p->d_grid, ck * p->nchan, 0);
            if (rc) return rc;
        }
        DegridParams g;
        memset(&g, 0, sizeof(g));
        g.udata = p->d_grid;
        g.nudata = static_cast<float2 *>(d_out) + (size_t)k0 * p->nchan * d.nro * d.npe1work;
        g.trig = trig + (size_t)k0 * trig_img_stride;
        g.trig_img_stride = trig_img_stride;
        g.tile_order = p->d_tile_order32;
        g.in_z = (long long)p->nchan * n2;
        g.in_c = (long long)n2;
        g.in_p = 1;
        g.in_shift = 1;
        g.in_transposed = p->fft512 ? 1 : 0;      // launch_fft512_forward stores the grid transposed
        g.debug = p->debug_skip;
        g.n = d.nxos;
        g.nrows = square ? 0 : d.nyos;
        g.nrep = p->nchan;
        g.nro = d.nro;
        g.npe = d.npe1work;
        g.nimg = ck;
        g.W = p->cfg.kernwidth;
        g.beta = p->beta;
        memcpy(g.kb_poly, p->kb_poly, sizeof(g.kb_poly));
        {
            StageTimer t(p, STAGE_DEGRID);
            if (p->cfg.kernwidth <= 3.f && !p->degrid_simple && square)
                HIP_TRY(launch_degrid_tile(g, p->kb_mode, p->stream));
            else
                HIP_TRY(launch_degrid(g, p->kb_mode, p->stream));      // also every non-square grid (supported, not tuned)
        }
    }
    return TRON_OK;
}

// CGNR, src/tron.cu:665-720 as Knopp et al. 2007 Alg. 1 intends it (the reference marks its own version "NOT WORKING
// CORRECTLY YET", :670; DESIGN.md lists the five repairs F1-F5), for slices [zfirst, zfirst+zcount) of a
// device-resident spoke stream, all slices of a chunk advancing together.  d_out: combine ? SoS images [z][nx*ny]
// : coil images [z][nchan*id + c].
int cgnr_run(tron_plan *p, void *d_out, const void *d_in_z0, int zfirst, int zcount, int combine)
{
    const tron_dims &d = p->d;
    if (p->cfg.input_half) return fail(TRON_ERR_UNSUPPORTED, "CGNR needs complex64 k-space (the residual lives in fp32)");
    const size_t n = (size_t)p->nchan * d.nro * d.npe1work;          // data-space elements per slice
    const size_t N = (size_t)p->nchan * d.nx * d.ny;                 // image-space elements per slice (F2)
    const size_t spoke_bytes = (size_t)d.nro * p->nchan * sizeof(float2);
    const int step = std::max(1, std::min(p->chunk, zcount));
    if (p->cg_slices < step) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (void *q : {(void *)p->d_cg_r, (void *)p->d_cg_v, (void *)p->d_cg_zt, (void *)p->d_cg_pt, (void *)p->d_cg_x,
                        (void *)p->d_cg_partial, (void *)p->d_cg_num, (void *)p->d_cg_coef})
            if (q) HIP_TRY(hipFree(q));
        p->d_cg_r = p->d_cg_v = p->d_cg_zt = p->d_cg_pt = p->d_cg_x = nullptr;
        p->d_cg_partial = p->d_cg_num = nullptr; p->d_cg_coef = nullptr; p->cg_slices = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_r), step * n * sizeof(float2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_v), step * n * sizeof(float2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_zt), step * N * sizeof(float2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_pt), step * N * sizeof(float2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_x), step * N * sizeof(float2)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_partial), (size_t)step * kCgPartials * sizeof(double)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_num), step * sizeof(double)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_coef), step * sizeof(float)));
        p->cg_slices = step;
    }
    const float unscale = (float)d.nxos * (float)d.npe1work;          // F3: the gridding kernel's 1/nxos/npe (src/tron.cu:532) divided out
    const int golden = p->cfg.golden_angle;
    hipStream_t st = p->stream;
    int rc;
    for (int z0 = 0; z0 < zcount; z0 += step) {
        const int cz = std::min(step, zcount - z0);
        const unsigned char *y = static_cast<const unsigned char *>(d_in_z0) + (size_t)z0 * d.prof_slide * spoke_bytes;
        // r = y: the (overlapping) windows of the stream, one contiguous copy per slice (:685; F5)
        for (int z = 0; z < cz; ++z)
            HIP_TRY(hipMemcpyAsync(p->d_cg_r + (size_t)z * n, y + (size_t)z * d.prof_slide * spoke_bytes, n * sizeof(float2),
                                   hipMemcpyDeviceToDevice, st));
        // ztilde = A^H W r (:686), ptilde = ztilde (:687), x = 0 (:683)
        if ((rc = adjoint_run(p, p->d_cg_zt, p->d_cg_r, zfirst + z0, cz, 0, d.npe1work))) return rc;
        HIP_TRY(launch_cg_scale_norm2(p->d_cg_zt, N, cz, unscale, p->d_cg_partial, st));
        HIP_TRY(launch_cg_finish(p->d_cg_partial, p->d_cg_num, p->d_cg_coef, 0, cz, st));
        HIP_TRY(hipMemcpyAsync(p->d_cg_pt, p->d_cg_zt, cz * N * sizeof(float2), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemsetAsync(p->d_cg_x, 0, cz * N * sizeof(float2), st));
        // the forward operator's angles: the slice's own index range (F4); golden angles are shared with the adjoint's table
        const float2 *trig = golden ? p->d_trig + (size_t)(zfirst + z0) * d.prof_slide : (p->d_trig_fwd ? p->d_trig_fwd : p->d_trig);
        const int trig_stride = golden ? d.prof_slide : 0;
        for (int t = 0; t < p->cfg.niter; ++t) {
            if ((rc = forward_run(p, p->d_cg_v, p->d_cg_pt, cz, trig, trig_stride, p->d_deapod_fwd))) return rc;   // v = A ptilde (:691)
            HIP_TRY(launch_cg_wnorm2(p->d_cg_v, n, cz, p->nchan, d.nro, p->dcf_a, p->dcf_b, p->d_cg_partial, st));  // <W v, v> (:693,696)
            HIP_TRY(launch_cg_finish(p->d_cg_partial, p->d_cg_num, p->d_cg_coef, 1, cz, st));                       // alpha (:697; F1)
            HIP_TRY(launch_cg_axpy(p->d_cg_x, p->d_cg_pt, p->d_cg_coef, 1.f, N, cz, st));                            // x += alpha ptilde (:699)
            if (t == p->cfg.niter - 1) break;                                                                       // (:701)
            HIP_TRY(launch_cg_axpy(p->d_cg_r, p->d_cg_v, p->d_cg_coef, -1.f, n, cz, st));                           // r -= alpha v (:703)
            if ((rc = adjoint_run(p, p->d_cg_zt, p->d_cg_r, zfirst + z0, cz, 0, d.npe1work))) return rc;            // ztilde = A^H W r (:707)
            HIP_TRY(launch_cg_scale_norm2(p->d_cg_zt, N, cz, unscale, p->d_cg_partial, st));
            HIP_TRY(launch_cg_finish(p->d_cg_partial, p->d_cg_num, p->d_cg_coef, 2, cz, st));                       // beta (:709; F1)
            HIP_TRY(launch_cg_xpby(p->d_cg_pt, p->d_cg_zt, p->d_cg_coef, N, cz, st));                               // ptilde = ztilde + beta ptilde (:710)
        }
        if (combine) {
            if ((rc = combine_coils(p, static_cast<float2 *>(d_out) + (size_t)z0 * d.nt * d.nx * d.ny, p->d_cg_x, cz))) return rc;        // (:764)
        } else
            HIP_TRY(hipMemcpyAsync(static_cast<float2 *>(d_out) + (size_t)z0 * N, p->d_cg_x, cz * N * sizeof(float2), hipMemcpyDeviceToDevice, st));   // (:713)
    }
    return TRON_OK;
}

int ensure_buffer(void **buf, size_t *have, size_t want)
{
    if (*have >= want && *buf) return TRON_OK;
    if (*buf) HIP_TRY(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(buf, want));
    *have = want;
    return TRON_OK;
}

int check_errflag(tron_plan *p)
{
    unsigned int flag = 0;
    HIP_TRY(hipMemcpy(&flag, p->d_errflag, sizeof(flag), hipMemcpyDeviceToHost));
    if (flag) {
        HIP_TRY(hipMemset(p->d_errflag, 0, sizeof(flag)));
        return fail(TRON_ERR_HIP, "gridding kernel reported an internal overflow (flag %u)", flag);
    }
    return TRON_OK;
}

}  // namespace

extern "C" const char *tron_last_error(void) { return g_last_error.c_str(); }

extern "C" const char *tron_version(void) { return "tronhip 0.1 (gfx950)"; }

extern "C" int tron_plan_create(tron_plan **out, const tron_config *cfg, const tron_dims *dims)
{
    if (!out || !cfg || !dims) return fail(TRON_ERR_INVALID, "tron_plan_create: null argument");
    *out = nullptr;
    const tron_dims &d = *dims;
    if (cfg->niter < 0 || (cfg->niter > 0 && cfg->input_half))
        return fail(TRON_ERR_UNSUPPORTED, "-i %d: CGNR needs niter >= 0 and complex64 k-space", cfg->niter);
    if (d.nt < 1 || d.nc < 1 || (long long)d.nc * d.nt > 4096)
        return fail(TRON_ERR_INVALID, "nc=%d nt=%d: channel count outside [1, 4096]", d.nc, d.nt);
    if (!(cfg->kernwidth > 0.f) || cfg->kernwidth > 4.f)
        return fail(TRON_ERR_UNSUPPORTED, "kernel width %g outside (0, 4]", cfg->kernwidth);
    // the adjoint is square by construction (nx = ny = nro/2, src/tron.cu:910-911); the forward transform also takes
    // non-square images (the reference's "TODO: implement non-square images", :945)
    if (d.nxos < 2 || d.nxos > 16384 || d.nyos < 2 || d.nyos > 16384 || d.nx < 1 || d.ny < 1 ||
        (cfg->adjoint && (d.nxos != d.nyos || d.nx != d.ny)))
        return fail(TRON_ERR_INVALID, "grid %dx%d (image %dx%d) is not a supported size", d.nxos, d.nyos, d.nx, d.ny);
    if (cfg->adjoint && d.nx > d.nxos)
        return fail(TRON_ERR_INVALID, "adjoint needs gridos >= 1 (nx=%d > nxos=%d)", d.nx, d.nxos);
    if (cfg->adjoint && (long long)d.nro / 2 + ((long long)(d.nxos / 2 - 1) * d.nro) / d.nxos >= d.nro)
        return fail(TRON_ERR_INVALID, "readout index would leave the spoke (nro=%d nxos=%d)", d.nro, d.nxos);
    if (cfg->input_half && !cfg->adjoint)
        return fail(TRON_ERR_UNSUPPORTED, "half-precision input is only defined for the adjoint (k-space) direction");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(TRON_ERR_HIP, "device %d requested but %d HIP device(s) present", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    (void)hipGetLastError();       // a stale error of an earlier, failed call must not be reported by this one
    std::call_once(g_fft_once, [] { rocfft_setup(); });
    // make every code object resident before anything is queued on a non-blocking stream
    HIP_TRY(warm_kernels());
    HIP_TRY(warm_grid_binned());
    HIP_TRY(warm_fft512());
    HIP_TRY(warm_degrid_tile());
    HIP_TRY(warm_cgnr());
    HIP_TRY(hipDeviceSynchronize());

    tron_plan *p = new tron_plan();
    p->cfg = *cfg;
    p->d = d;
    p->nchan = d.nc * d.nt;
    p->beta = kb_beta(cfg->kernwidth);
    p->kb_mode = cfg->kb_mode == TRON_KB_FAST ? TRON_KB_FAST : TRON_KB_EXACT;
    p->kb_poly_err = kb_poly_fit(cfg->kernwidth, p->kb_poly, kKbPolyTerms);
    if (p->kb_mode == TRON_KB_FAST && !(p->kb_poly_err < 1e-7)) p->kb_mode = TRON_KB_EXACT;   // polynomial too short for this beta: stay exact
    dcf_constants(d.nro, d.npe1work, &p->dcf_a, &p->dcf_b);
    p->scale = grid_scale(d.nxos, d.npe1work);

    const size_t n2 = (size_t)d.nxos * d.nyos;
    const size_t per_unit = (size_t)p->nchan * n2 * sizeof(float2);
    int units = cfg->adjoint ? d.nz : 1;
    // The heaviest tile (the k-space centre, crossed by every spoke) is one wave's serial work, so a
    // launch needs enough slices in flight to cover that critical path: batch up to 1 GiB of grid.
    // ... or 64 slices when the coils are many (still at most 6 GiB of grid: 288 GB of HBM make that cheap)
    size_t auto_chunk = std::max<size_t>(1, ((size_t)1 << 30) / per_unit);
    if (auto_chunk < 64) auto_chunk = std::max<size_t>(auto_chunk, std::min<size_t>(64, ((size_t)6 << 30) / per_unit));
    int chunk = cfg->chunk_slices > 0 ? cfg->chunk_slices : (int)std::max<size_t>(1, auto_chunk);
    if (const char *env = getenv("TRON_CHUNK_SLICES")) chunk = std::max(1, atoi(env));
    p->chunk = std::max(1, std::min(chunk, std::max(units, 1)));
    p->chunk_cap = std::max(p->chunk, std::min(std::max(units, 1), p->chunk + p->chunk / 2));
    if (!cfg->adjoint) p->chunk = p->chunk_cap = std::max(1, chunk);

    int rc = TRON_OK;
    auto bail = [&](int code) { tron_plan_destroy(p); return code; };
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(TRON_ERR_HIP, "hipStreamCreate failed"));

    p->ntrig = trig_table_size(*cfg, d);
    {
        std::vector<float> trig(2 * p->ntrig);
        build_trig_table(*cfg, d, trig.data(), p->ntrig);
        if ((rc = upload(&p->d_trig, trig.data(), trig.size() * sizeof(float)))) return bail(rc);
    }
    {   // 32x32 tiles, centre first: binned gridding and tiled degridding
        std::vector<int> order;
        build_tile_order(d.nxos, kBinnedTile, order);
        if ((rc = upload(&p->d_tile_order32, order.data(), order.size() * sizeof(int)))) return bail(rc);
    }
    if (cfg->adjoint) {
        std::vector<uint32_t> band(n2);
        build_band_table(d.nxos, cfg->kernwidth, band.data());
        if ((rc = upload(&p->d_band, band.data(), band.size() * sizeof(uint32_t)))) return bail(rc);
        std::vector<int> order;
        build_tile_order(d.nxos, kTile, order);
        p->tiles_per_row = (d.nxos + kTile - 1) / kTile;
        p->ntiles = (int)order.size();
        if ((rc = upload(&p->d_tile_order, order.data(), order.size() * sizeof(int)))) return bail(rc);
        p->binned = p->kb_mode == TRON_KB_FAST && cfg->kernwidth <= 3.f;
        if (const char *gk = getenv("TRON_GRID_KERNEL")) p->binned = p->binned && strcmp(gk, "gather") != 0;
        if (p->binned) {
            std::vector<int> sorder, slots;
            int target = 2500;                                            // records per workgroup and image
            if (const char *e = getenv("TRON_SPLIT_TARGET")) target = std::max(64, atoi(e));
            p->max_parts = 8;
            build_split_tile_order(d.nxos, kBinnedTile, d.npe1work, cfg->kernwidth, target, p->max_parts, sorder, slots);
            p->split_entries = (int)sorder.size();
            p->nsplit_slots = (int)slots.size();
            p->split_below = 64;                                          // launches of fewer slices use the split list
            if (const char *e = getenv("TRON_SPLIT_BELOW")) p->split_below = atoi(e);
            if (p->nsplit_slots > 0) {
                if ((rc = upload(&p->d_tile_order32_split, sorder.data(), sorder.size() * sizeof(int)))) return bail(rc);
                if ((rc = upload(&p->d_split_slots, slots.data(), slots.size() * sizeof(int)))) return bail(rc);
            }
        }
        std::vector<float> dea((size_t)d.nx * d.nx);
        build_deapod_table(d.nx, cfg->kernwidth, cfg->gridos, dea.data());        // src/tron.cu:635
        if ((rc = upload(&p->d_deapod, dea.data(), dea.size() * sizeof(float)))) return bail(rc);
        if (cfg->niter > 0) {
            // CGNR applies the forward operator too (src/tron.cu:691): its deapodisation table and, for linear angles,
            // the degridding kernel's own angle convention (:555) unless cgnr_consistent asks for the gridding one (:509; Q5)
            std::vector<float> deaf(n2);
            build_deapod_table(d.nxos, cfg->kernwidth, 1.f, deaf.data());         // src/tron.cu:643
            if ((rc = upload(&p->d_deapod_fwd, deaf.data(), deaf.size() * sizeof(float)))) return bail(rc);
            if (!cfg->golden_angle && !cfg->cgnr_consistent) {
                tron_config fc = *cfg;
                fc.adjoint = 0;
                std::vector<float> tf(2 * (size_t)d.npe1work);
                build_trig_table(fc, d, tf.data(), (size_t)d.npe1work);
                if ((rc = upload(&p->d_trig_fwd, tf.data(), tf.size() * sizeof(float)))) return bail(rc);
            }
        }
    } else {
        std::vector<float> dea(n2);
        if (d.nxos == d.nyos) build_deapod_table(d.nxos, cfg->kernwidth, 1.f, dea.data());             // src/tron.cu:643
        else build_deapod_table_rect(d.nyos, d.nxos, cfg->kernwidth, 1.f, dea.data());
        if ((rc = upload(&p->d_deapod, dea.data(), dea.size() * sizeof(float)))) return bail(rc);
    }
    unsigned int zero = 0;
    if ((rc = upload(&p->d_errflag, &zero, sizeof(zero)))) return bail(rc);
    p->poison = getenv("TRON_POISON_GRID") != nullptr;   // tests: NaN-fill the work grid so a read of a never-written point shows up
    if (d.nxos == 512 && d.nx == 256 && d.nyos == 512 && d.ny == 256) {
        p->fft512 = true;
        if (const char *ff = getenv("TRON_FFT")) p->fft512 = strcmp(ff, "rocfft") != 0;
    }
    if (p->fft512) {
        std::vector<float> tw(2 * 512);
        for (int k = 0; k < 512; ++k) {
            tw[2 * k] = (float)cos(2.0 * M_PI * k / 512.0);
            tw[2 * k + 1] = (float)sin(2.0 * M_PI * k / 512.0);
        }
        if ((rc = upload(&p->d_tw512, tw.data(), tw.size() * sizeof(float)))) return bail(rc);
        // Two lanes (gridding on `stream`, FFT passes on `stream2`, two Cartesian buffers): round 1 measured +2 % and left
        // it off; with the round-2 FFT passes it is +4.9 % at 8 coils x 256 slices (+2.7 % at 6 coils, +2 % at 4, 0 at 1),
        // so adjoint plans with at least two full batches and more than one channel have it on.  TRON_DUAL_STREAM=0/1 overrides.
        p->dual = cfg->adjoint && p->nchan > 1 && d.nz >= 2 * p->chunk;
        if (const char *ds = getenv("TRON_DUAL_STREAM")) p->dual = d.nz > 1 && atoi(ds) != 0;
        if (p->dual) {
            // TRON_CU_SPLIT=k: give the (HBM-bound) FFT lane every k-th CU and the (VALU/LDS-bound) gridding lane
            // the rest, so the two overlap in space instead of queueing behind each other
            int split = 0;
            if (const char *cs = getenv("TRON_CU_SPLIT")) split = atoi(cs);
            if (split >= 2) {
                uint32_t mask_fft[8], mask_grid[8];
                for (int w = 0; w < 8; ++w) { mask_fft[w] = 0; mask_grid[w] = 0; }
                for (int cu = 0; cu < 256; ++cu) {
                    if (cu % split == 0) mask_fft[cu / 32] |= 1u << (cu % 32);
                    else mask_grid[cu / 32] |= 1u << (cu % 32);
                }
                hipStream_t masked = nullptr;
                if (hipExtStreamCreateWithCUMask(&p->stream2, 8, mask_fft) != hipSuccess ||
                    hipExtStreamCreateWithCUMask(&masked, 8, mask_grid) != hipSuccess)
                    return bail(fail(TRON_ERR_HIP, "cannot create CU-masked streams"));
                hipStreamDestroy(p->stream);
                p->stream = masked;
            } else {
                // the FFT lane's workgroups are short and light: at a higher stream priority they slip into every slot the
                // gridding lane frees instead of queueing behind its backlog (TRON_FFT_PRIO: 0 = same priority)
                int lo = 0, hi = 0, prio = 0;
                hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = least urgent (numerically greatest), hi = most urgent
                const char *fp = getenv("TRON_FFT_PRIO");
                const int want = fp ? atoi(fp) : -1;                // -1: most urgent, +1: least urgent
                prio = want < 0 ? hi : (want > 0 ? lo : 0);
                if (hipStreamCreateWithPriority(&p->stream2, hipStreamNonBlocking, prio) != hipSuccess)
                    return bail(fail(TRON_ERR_HIP, "cannot create the FFT lane"));
            }
            for (int i = 0; i < 2; ++i)
                if (hipEventCreateWithFlags(&p->ev_g[i], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&p->ev_f[i], hipEventDisableTiming) != hipSuccess)
                    return bail(fail(TRON_ERR_HIP, "cannot create pipeline events"));
        }
    }
    // adjoint: the whole batch now (an out-of-memory plan fails here, not mid-run); forward: on first use, sized by the call
    if (cfg->adjoint && (rc = ensure_work(p, p->chunk_cap))) return bail(rc);
    if (!p->fft512) {   // rocFFT plans for the batch sizes this plan will certainly see: no plan creation (and device
        FftPlan *f = nullptr;       // synchronisation) in the middle of the first pipeline
        if ((rc = get_fft(p, (cfg->adjoint ? std::min(p->chunk, std::max(d.nz, 1)) : 1) * p->nchan, cfg->adjoint ? 1 : 0, &f))) return bail(rc);
    }
    if (cfg->verbose) {
        printf("tronhip: device %d, %s, nchan %d, grid %d^2 -> image %d^2, %d spokes/image, chunk %d, KB %s\n",
               cfg->device, cfg->adjoint ? "adjoint" : "forward", p->nchan, d.nxos, d.nx, d.npe1work, p->chunk,
               p->kb_mode == TRON_KB_FAST ? "fast" : "exact");
        printf("tronhip: fast Kaiser-Bessel polynomial max relative error %.2e\n", p->kb_poly_err);
    }
    if (const char *se = getenv("TRON_SYNC_EACH")) p->sync_each = atoi(se) != 0;
    p->debug_skip = 0;
    if (const char *dbg = getenv("TRON_DEBUG_SKIP")) p->debug_skip = atoi(dbg);
    p->degrid_simple = getenv("TRON_DEGRID_SIMPLE") != nullptr;
    if (const char *sp = getenv("TRON_SLICES_PER_PASS")) p->slices_per_pass = atoi(sp) != 0;
    if (const char *lp = getenv("TRON_GRID_LDS_PAD")) p->grid_lds_pad = atoi(lp);
    p->no_disc = getenv("TRON_NO_DISC") != nullptr;
    p->pin_host = cfg->pin_host != 0;
    if (const char *ph = getenv("TRON_PIN_HOST")) p->pin_host = atoi(ph) != 0;
    *out = p;
    return TRON_OK;
}

extern "C" int tron_plan_destroy(tron_plan *p)
{
    if (!p) return TRON_OK;
    hipSetDevice(p->cfg.device);
    if (p->stream) hipStreamSynchronize(p->stream);
    if (p->stream2) hipStreamSynchronize(p->stream2);
    for (int s = 0; s < STAGE_COUNT; ++s)
        for (auto &pr : p->ev[s]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto &kv : p->fft) {
        if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
        if (kv.second.plan) rocfft_plan_destroy(kv.second.plan);
        if (kv.second.work) hipFree(kv.second.work);
    }
    hipFree(p->d_trig);
    hipFree(p->d_band);
    hipFree(p->d_tile_order);
    hipFree(p->d_tile_order32);
    hipFree(p->d_coil_tmp);
    hipFree(p->d_deapod_fwd);
    hipFree(p->d_trig_fwd);
    hipFree(p->d_cg_r); hipFree(p->d_cg_v); hipFree(p->d_cg_zt); hipFree(p->d_cg_pt); hipFree(p->d_cg_x);
    hipFree(p->d_cg_partial); hipFree(p->d_cg_num); hipFree(p->d_cg_coef);
    hipFree(p->d_tile_order32_split);
    hipFree(p->d_split_slots);
    hipFree(p->d_partial);
    hipFree(p->d_deapod);
    hipFree(p->d_errflag);
    hipFree(p->d_grid);
    hipFree(p->d_stage_in);
    hipFree(p->d_stage_out);
    hipFree(p->d_trig_tmp);
    hipFree(p->d_tw512);
    hipFree(p->d_fft_tmp);
    hipFree(p->d_grid2);
    for (int i = 0; i < 2; ++i) {
        if (p->ev_g[i]) hipEventDestroy(p->ev_g[i]);
        if (p->ev_f[i]) hipEventDestroy(p->ev_f[i]);
    }
    for (hipEvent_t e : p->ev_pipe) hipEventDestroy(e);
    if (p->stream_up) hipStreamDestroy(p->stream_up);
    if (p->stream_down) hipStreamDestroy(p->stream_down);
    if (p->stream2) hipStreamDestroy(p->stream2);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
    return TRON_OK;
}

extern "C" int tron_plan_sync(tron_plan *p)
{
    if (!p) return fail(TRON_ERR_INVALID, "tron_plan_sync: null plan");
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->stream2) HIP_TRY(hipStreamSynchronize(p->stream2));
    p->fft_pending[0] = p->fft_pending[1] = false;
    return check_errflag(p);
}

extern "C" int tron_nufft_adj_radial2d(tron_plan *p, void *d_out, const void *d_in, int zfirst, int zcount, int combine)
{
    if (!p || !d_out || !d_in) return fail(TRON_ERR_INVALID, "tron_nufft_adj_radial2d: null argument");
    if (!p->cfg.adjoint) return fail(TRON_ERR_INVALID, "plan was created for the forward direction");
    const tron_dims &d = p->d;
    if (zfirst < 0 || zcount < 0 || zfirst + zcount > d.nz)
        return fail(TRON_ERR_INVALID, "slice range [%d,%d) outside [0,%d)", zfirst, zfirst + zcount, d.nz);
    const long long last = (long long)(zfirst + zcount - 1) * d.prof_slide + d.npe1work;
    if (zcount > 0 && last > (long long)d.npe1 * d.npe2)
        return fail(TRON_ERR_INVALID, "slice %d would read spokes up to %lld but the input holds %lld (the reference reads out of bounds here)",
                    zfirst + zcount - 1, last, (long long)d.npe1 * d.npe2);
    HIP_TRY(hipSetDevice(p->cfg.device));
    const size_t elem = p->cfg.input_half ? 4 : 8;
    const unsigned char *in = static_cast<const unsigned char *>(d_in) + (size_t)zfirst * d.prof_slide * d.nro * p->nchan * elem;
    return adjoint_run(p, d_out, in, zfirst, zcount, combine, 0, true);   // asynchronous: the FFT lane is joined by tron_plan_sync / the next call
}

extern "C" int tron_cgnr_radial2d(tron_plan *p, void *d_out, const void *d_in, int zfirst, int zcount, int combine)
{
    if (!p || !d_out || !d_in) return fail(TRON_ERR_INVALID, "tron_cgnr_radial2d: null argument");
    if (!p->cfg.adjoint) return fail(TRON_ERR_INVALID, "plan was created for the forward direction");
    if (p->cfg.niter <= 0) return fail(TRON_ERR_INVALID, "plan was created with niter = 0");
    const tron_dims &d = p->d;
    if (zfirst < 0 || zcount < 0 || zfirst + zcount > d.nz)
        return fail(TRON_ERR_INVALID, "slice range [%d,%d) outside [0,%d)", zfirst, zfirst + zcount, d.nz);
    if (zcount > 0 && (long long)(zfirst + zcount - 1) * d.prof_slide + d.npe1work > (long long)d.npe1 * d.npe2)
        return fail(TRON_ERR_INVALID, "slice %d would read past the spoke stream", zfirst + zcount - 1);
    HIP_TRY(hipSetDevice(p->cfg.device));
    const unsigned char *in = static_cast<const unsigned char *>(d_in) + (size_t)zfirst * d.prof_slide * d.nro * p->nchan * sizeof(float2);
    return cgnr_run(p, d_out, in, zfirst, zcount, combine);
}

extern "C" int tron_nufft_radial2d(tron_plan *p, void *d_out, const void *d_in, int nimg)
{
    if (!p || !d_out || !d_in) return fail(TRON_ERR_INVALID, "tron_nufft_radial2d: null argument");
    if (p->cfg.adjoint) return fail(TRON_ERR_INVALID, "plan was created for the adjoint direction");
    if (nimg < 0) return fail(TRON_ERR_INVALID, "nimg < 0");
    HIP_TRY(hipSetDevice(p->cfg.device));
    return forward_run(p, d_out, d_in, nimg);
}

// Adjoint of slices [zfirst, zfirst+zcount) from host memory: h_in_block points at the first spoke of slice zfirst's
// window, h_out_block at that slice's image.
static int adjoint_block(tron_plan *p, tron_float2 *h_out_block, const void *h_in_block, int zfirst, int zcount)
{
    const tron_dims &d = p->d;
    const size_t elem = p->cfg.input_half ? 4 : 8;
    int rc;
    // Every spoke the range touches is uploaded ONCE; windows are views (src/tron.cu:738-748).  The range is cut
    // into chunks of p->chunk slices and run as a three-lane pipeline -- upload(k+1) || kernels(k) || download(k-1)
    // on three streams chained by events -- where the reference alternates two streams per slice and re-uploads
    // every window (src/tron.cu:732-783).  Chunk k+1 uploads only the spokes chunk k did not.
    size_t spoke_bytes = 0, nspokes = 0, in_bytes = 0, out_elems = 0, out_bytes = 0;
    if (__builtin_mul_overflow((size_t)d.nro * elem, (size_t)p->nchan, &spoke_bytes) ||
        __builtin_mul_overflow((size_t)(zcount - 1), (size_t)d.prof_slide, &nspokes) ||
        __builtin_add_overflow(nspokes, (size_t)d.npe1work, &nspokes) ||
        __builtin_mul_overflow(nspokes, spoke_bytes, &in_bytes) ||
        __builtin_mul_overflow((size_t)zcount * d.nt, (size_t)d.nx * d.ny, &out_elems) ||
        __builtin_mul_overflow(out_elems, sizeof(float2), &out_bytes))
        return fail(TRON_ERR_INVALID, "slice range [%d,%d): staging size overflows", zfirst, zfirst + zcount);
    if ((rc = ensure_buffer(&p->d_stage_in, &p->stage_in_bytes, in_bytes))) return rc;
    if ((rc = ensure_buffer(&p->d_stage_out, &p->stage_out_bytes, out_bytes))) return rc;
    if (!p->stream_up) HIP_TRY(hipStreamCreateWithFlags(&p->stream_up, hipStreamNonBlocking));
    if (!p->stream_down) HIP_TRY(hipStreamCreateWithFlags(&p->stream_down, hipStreamNonBlocking));
    const unsigned char *src = reinterpret_cast<const unsigned char *>(h_in_block);
    tron_float2 *dst = h_out_block;
    // Pinning the caller's buffers makes the copies truly asynchronous (and the two directions concurrent); it
    // costs a page walk of the whole range, so it is opt-in (cfg.pin_host / TRON_PIN_HOST=1): pageable copies are
    // staged by the runtime at the same PCIe rate and still overlap the kernels of the previous chunk.
    bool pinned_in = false, pinned_out = false;
    if (p->pin_host) {
        pinned_in = hipHostRegister(const_cast<unsigned char *>(src), in_bytes, hipHostRegisterDefault) == hipSuccess;
        pinned_out = hipHostRegister(dst, out_bytes, hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();
    }
    const int step = std::max(1, std::min(p->chunk, zcount));
    const int nchunks = (zcount + step - 1) / step;
    while ((int)p->ev_pipe.size() < 2 * nchunks) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->ev_pipe.push_back(e);
    }
    const size_t img_elems = (size_t)d.nt * d.nx * d.ny;
    size_t uploaded = 0;                                          // spokes of the range already on the device
    rc = TRON_OK;
    hipError_t he = hipSuccess;
    for (int k = 0; k < nchunks && rc == TRON_OK && he == hipSuccess; ++k) {
        const int z0 = k * step, cz = std::min(step, zcount - z0);
        const size_t need = (size_t)(z0 + cz - 1) * d.prof_slide + d.npe1work;
        if (need > uploaded) {
            he = hipMemcpyAsync(static_cast<unsigned char *>(p->d_stage_in) + uploaded * spoke_bytes, src + uploaded * spoke_bytes,
                                (need - uploaded) * spoke_bytes, hipMemcpyHostToDevice, p->stream_up);
            uploaded = need;
        }
        if (he == hipSuccess) he = hipEventRecord(p->ev_pipe[2 * k], p->stream_up);
        if (he == hipSuccess) he = hipStreamWaitEvent(p->stream, p->ev_pipe[2 * k], 0);
        if (he != hipSuccess) break;
        rc = (p->cfg.niter > 0 ? cgnr_run : [](tron_plan *pp, void *o, const void *i, int zf, int zc, int cb) { return adjoint_run(pp, o, i, zf, zc, cb); })
                (p, static_cast<float2 *>(p->d_stage_out) + (size_t)z0 * img_elems,
                 static_cast<const unsigned char *>(p->d_stage_in) + (size_t)z0 * d.prof_slide * spoke_bytes,
                 zfirst + z0, cz, 1);                             // niter > 0: src/tron.cu:754-755; + coilcombinesos, :764
        if (rc != TRON_OK) break;
        he = hipEventRecord(p->ev_pipe[2 * k + 1], p->stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(p->stream_down, p->ev_pipe[2 * k + 1], 0);
        if (he == hipSuccess)
            he = hipMemcpyAsync(dst + (size_t)z0 * img_elems, static_cast<float2 *>(p->d_stage_out) + (size_t)z0 * img_elems,
                                (size_t)cz * img_elems * sizeof(float2), hipMemcpyDeviceToHost, p->stream_down);
    }
    hipError_t s1 = hipStreamSynchronize(p->stream_up), s2 = hipStreamSynchronize(p->stream), s3 = hipStreamSynchronize(p->stream_down);
    if (pinned_in) hipHostUnregister(const_cast<unsigned char *>(src));
    if (pinned_out) hipHostUnregister(dst);
    if (rc != TRON_OK) return rc;
    for (hipError_t e : {he, s1, s2, s3})
        if (e != hipSuccess) return fail(TRON_ERR_HIP, "host-buffer pipeline failed: %s", hipGetErrorString(e));
    return tron_plan_sync(p);
}

extern "C" int tron_recon_radial2d_range(tron_plan *p, tron_float2 *h_out, const tron_float2 *h_in, int zfirst, int zcount)
{
    if (!p || !h_out || !h_in) return fail(TRON_ERR_INVALID, "tron_recon_radial2d: null argument");
    const tron_dims &d = p->d;
    if (zfirst < 0 || zcount < 0 || zfirst + zcount > d.nz)
        return fail(TRON_ERR_INVALID, "slice range [%d,%d) outside [0,%d)", zfirst, zfirst + zcount, d.nz);
    if (zcount == 0) return TRON_OK;
    HIP_TRY(hipSetDevice(p->cfg.device));
    int rc;
    if (p->cfg.adjoint) {
        const size_t elem = p->cfg.input_half ? 4 : 8;
        const long long last = (long long)(zfirst + zcount - 1) * d.prof_slide + d.npe1work;
        if (last > (long long)d.npe1 * d.npe2)
            return fail(TRON_ERR_INVALID, "slice %d would read spokes up to %lld but the input holds %lld (the reference reads out of bounds here)",
                        zfirst + zcount - 1, last, (long long)d.npe1 * d.npe2);
        const size_t spoke_bytes = (size_t)d.nro * p->nchan * elem;
        return adjoint_block(p, h_out + (size_t)d.nt * d.nx * d.ny * zfirst,                     // img_offset, src/tron.cu:740,768
                             reinterpret_cast<const unsigned char *>(h_in) + (size_t)zfirst * d.prof_slide * spoke_bytes, zfirst, zcount);
    }
    // forward: every z reads h_in + nc*nt*nro*(z*prof_slide) (src/tron.cu:738-739,750) -- with the
    // default prof_slide that is slice 0 for every z -- and writes block z (src/tron.cu:776)
    const size_t in_elems = (size_t)p->nchan * d.nx * d.ny;
    const size_t out_elems = (size_t)p->nchan * d.nro * d.npe1work;
    if ((rc = ensure_buffer(&p->d_stage_in, &p->stage_in_bytes, in_elems * sizeof(float2)))) return rc;
    if ((rc = ensure_buffer(&p->d_stage_out, &p->stage_out_bytes, out_elems * sizeof(float2)))) return rc;
    for (int z = zfirst; z < zfirst + zcount; ++z) {
        if ((uint64_t)(z + 1) * out_elems * sizeof(float2) > d.out_bytes) break;   // h_out is sized for npe2 blocks (src/tron.cu:960)
        const size_t data_offset = (size_t)p->nchan * d.nro * ((size_t)z * p->cfg.prof_slide);
        if (data_offset + in_elems > d.in_elems)
            return fail(TRON_ERR_INVALID, "forward slice %d would read past the input (offset %zu)", z, data_offset);
        HIP_TRY(hipMemcpyAsync(p->d_stage_in, h_in + data_offset, in_elems * sizeof(float2), hipMemcpyHostToDevice, p->stream));
        if ((rc = forward_run(p, p->d_stage_out, p->d_stage_in, 1))) return rc;
        HIP_TRY(hipMemcpyAsync(h_out + out_elems * z, p->d_stage_out, out_elems * sizeof(float2), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    return tron_plan_sync(p);
}

extern "C" int tron_recon_radial2d(tron_plan *p, tron_float2 *h_out, const tron_float2 *h_in)
{
    if (!p) return fail(TRON_ERR_INVALID, "tron_recon_radial2d: null plan");
    return tron_recon_radial2d_range(p, h_out, h_in, 0, p->d.nz);
}

extern "C" int tron_recon_radial2d_block(tron_plan *p, tron_float2 *h_out_block, const void *h_in_block, int zfirst, int zcount)
{
    if (!p || !h_out_block || !h_in_block) return fail(TRON_ERR_INVALID, "tron_recon_radial2d_block: null argument");
    if (!p->cfg.adjoint) return fail(TRON_ERR_UNSUPPORTED, "tron_recon_radial2d_block: defined for the adjoint (one forward run is one image)");
    const tron_dims &d = p->d;
    if (zfirst < 0 || zcount < 0 || zfirst + zcount > d.nz)
        return fail(TRON_ERR_INVALID, "slice range [%d,%d) outside [0,%d)", zfirst, zfirst + zcount, d.nz);
    if (zcount == 0) return TRON_OK;
    if ((long long)(zfirst + zcount - 1) * d.prof_slide + d.npe1work > (long long)d.npe1 * d.npe2)
        return fail(TRON_ERR_INVALID, "slice %d would read past the spoke stream", zfirst + zcount - 1);
    HIP_TRY(hipSetDevice(p->cfg.device));
    return adjoint_block(p, h_out_block, h_in_block, zfirst, zcount);
}

// One host worker thread and one plan per device, contiguous slice blocks written straight into the caller's output:
// the reference's compiled-out MULTI_GPU round-robin (src/tron.cu:582-597,735-736) made contiguous; no inter-GPU traffic.
extern "C" int tron_recon_radial2d_multi(const tron_config *cfg, const tron_dims *dims, const int *devices, int n_devices,
                                         tron_float2 *h_out, const tron_float2 *h_in)
{
    if (!cfg || !dims || !h_out || !h_in) return fail(TRON_ERR_INVALID, "tron_recon_radial2d_multi: null argument");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (n_devices <= 0) n_devices = ndev;
    if (n_devices < 1) return fail(TRON_ERR_HIP, "no HIP device");
    std::vector<int> devs(n_devices);
    for (int g = 0; g < n_devices; ++g) {
        devs[g] = devices ? devices[g] : g;
        if (devs[g] < 0 || devs[g] >= ndev) return fail(TRON_ERR_HIP, "device %d requested but %d HIP device(s) present", devs[g], ndev);
    }
    const int nz = dims->nz;
    const int workers = (cfg->adjoint && nz > 1) ? std::min(n_devices, nz) : 1;   // a forward run is one image
    std::vector<int> rcs(workers, TRON_OK);
    std::vector<std::string> msgs(workers);
    // the workers' slice blocks share spokes (windows overlap) and pages: pin both buffers ONCE, visible to every device
    bool pinned_in = false, pinned_out = false;
    const size_t in_bytes = (size_t)dims->in_elems * (cfg->input_half ? 4 : 8);
    if (cfg->pin_host && workers > 1) {
        HIP_TRY(hipSetDevice(devs[0]));
        pinned_in = hipHostRegister(const_cast<tron_float2 *>(h_in), in_bytes, hipHostRegisterPortable) == hipSuccess;
        pinned_out = hipHostRegister(h_out, (size_t)dims->out_bytes, hipHostRegisterPortable) == hipSuccess;
        (void)hipGetLastError();
    }
    auto work = [&](int g) {
        tron_config c = *cfg;
        c.device = devs[g];
        if (workers > 1) c.pin_host = 0;
        tron_plan *plan = nullptr;
        int rc = tron_plan_create(&plan, &c, dims);
        if (rc == TRON_OK) {
            const int z0 = (int)((long long)g * nz / workers), z1 = (int)((long long)(g + 1) * nz / workers);
            rc = tron_recon_radial2d_range(plan, h_out, h_in, z0, z1 - z0);
        }
        if (rc != TRON_OK) msgs[g] = tron_last_error();          // the message lives in this worker's thread-local slot
        tron_plan_destroy(plan);
        rcs[g] = rc;
    };
    if (workers == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int g = 0; g < workers; ++g) th.emplace_back(work, g);
        for (auto &t : th) t.join();
    }
    if (pinned_in) hipHostUnregister(const_cast<tron_float2 *>(h_in));
    if (pinned_out) hipHostUnregister(h_out);
    for (int g = 0; g < workers; ++g)
        if (rcs[g] != TRON_OK) return fail(rcs[g], "device worker %d (HIP device %d): %s", g, devs[g], msgs[g].c_str());
    return TRON_OK;
}

extern "C" int tron_precompensate(tron_plan *p, void *d_nudata)
{
    if (!p || !d_nudata) return fail(TRON_ERR_INVALID, "tron_precompensate: null argument");
    if (!p->cfg.adjoint) return fail(TRON_ERR_INVALID, "plan was created for the forward direction");
    HIP_TRY(hipSetDevice(p->cfg.device));
    HIP_TRY(launch_precompensate(static_cast<float2 *>(d_nudata), p->nchan, p->d.nro, p->d.npe1work, p->dcf_a, p->dcf_b, p->stream));
    return TRON_OK;
}

extern "C" int tron_gridradial2d(tron_plan *p, void *d_udata, const void *d_nudata, int skip)
{
    if (!p || !d_udata || !d_nudata) return fail(TRON_ERR_INVALID, "tron_gridradial2d: null argument");
    if (!p->cfg.adjoint) return fail(TRON_ERR_INVALID, "plan was created for the forward direction");
    HIP_TRY(hipSetDevice(p->cfg.device));
    const tron_dims &d = p->d;
    std::vector<float> trig(2 * (size_t)d.npe1work);
    build_trig_table_window(d.npe1work, skip, p->cfg.golden_angle, trig.data());
    if (!p->d_trig_tmp) HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_trig_tmp), trig.size() * sizeof(float)));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(p->d_trig_tmp, trig.data(), trig.size() * sizeof(float), hipMemcpyHostToDevice));
    GridParams g;
    memset(&g, 0, sizeof(g));
    fill_grid_consts(p, g);
    g.nudata = d_nudata;
    g.udata = static_cast<float2 *>(d_udata);
    g.trig = p->d_trig_tmp;
    g.in_slice_stride = 0;
    g.trig_slice_stride = 0;
    g.nslices = 1;
    g.apply_dcf = 0;                      // the caller ran precompensate, as at src/tron.cu:628-629
    g.out_z = 0;
    g.out_c = 1;                          // udata[nchan*id + ch], src/tron.cu:534
    g.out_p = p->nchan;
    g.out_shift = 0;
    StageTimer t(p, STAGE_GRID);
    if (p->binned) {
        g.tile_order = p->d_tile_order32;
        HIP_TRY(launch_grid_binned(g, 0, p->stream));
    } else {
        HIP_TRY(launch_grid(g, p->kb_mode, 0, p->stream));
    }
    return TRON_OK;
}

extern "C" int tron_degridradial2d(tron_plan *p, void *d_nudata, const void *d_udata)
{
    if (!p || !d_udata || !d_nudata) return fail(TRON_ERR_INVALID, "tron_degridradial2d: null argument");
    if (p->cfg.adjoint) return fail(TRON_ERR_INVALID, "plan was created for the adjoint direction");
    HIP_TRY(hipSetDevice(p->cfg.device));
    const tron_dims &d = p->d;
    DegridParams g;
    memset(&g, 0, sizeof(g));
    g.udata = static_cast<const float2 *>(d_udata);
    g.nudata = static_cast<float2 *>(d_nudata);
    g.trig = p->d_trig;
    g.tile_order = p->d_tile_order32;
    g.in_z = 0;
    g.in_c = 1;                           // udata[nrep*(i*n+j) + c], src/tron.cu:571-573
    g.in_p = p->nchan;
    g.in_shift = 0;
    g.n = d.nxos;
    g.nrep = p->nchan;
    g.nro = d.nro;
    g.npe = d.npe1work;
    g.nimg = 1;
    g.W = p->cfg.kernwidth;
    g.beta = p->beta;
    memcpy(g.kb_poly, p->kb_poly, sizeof(g.kb_poly));
    StageTimer t(p, STAGE_DEGRID);
    if (p->cfg.kernwidth <= 3.f && !p->degrid_simple)
        HIP_TRY(launch_degrid_tile(g, p->kb_mode, p->stream));
    else
        HIP_TRY(launch_degrid(g, p->kb_mode, p->stream));
    return TRON_OK;
}

// Gridding and FFT launches of consecutive batches overlap on two streams when the plan has a second lane; enable = 0
// serialises them on one stream (each kernel then runs alone: what a per-kernel duration should be measured on),
// enable = 1 restores the plan's default.  Returns TRON_OK; *had_two_lanes (optional) tells whether the plan has the lane.
extern "C" int tron_plan_two_lanes(tron_plan *p, int enable, int *had_two_lanes)
{
    if (!p) return fail(TRON_ERR_INVALID, "null plan");
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->stream2) HIP_TRY(hipStreamSynchronize(p->stream2));
    p->fft_pending[0] = p->fft_pending[1] = false;
    if (had_two_lanes) *had_two_lanes = p->stream2 != nullptr && p->d_grid2 != nullptr;
    p->dual = enable != 0 && p->stream2 != nullptr && p->d_grid2 != nullptr;
    return TRON_OK;
}

extern "C" int tron_plan_timing(tron_plan *p, int enable)
{
    if (!p) return fail(TRON_ERR_INVALID, "null plan");
    int rc = drain_timers(p);
    p->timing = enable != 0;
    return rc;
}

extern "C" int tron_plan_timing_get(tron_plan *p, int stage, double *ms, uint64_t *launches)
{
    if (!p || stage < 0 || stage >= STAGE_COUNT) return fail(TRON_ERR_INVALID, "bad stage");
    int rc = drain_timers(p);
    if (rc) return rc;
    if (ms) *ms = p->ms_acc[stage];
    if (launches) *launches = p->launches[stage];
    return TRON_OK;
}

extern "C" int tron_plan_timing_reset(tron_plan *p)
{
    if (!p) return fail(TRON_ERR_INVALID, "null plan");
    int rc = drain_timers(p);
    for (int s = 0; s < STAGE_COUNT; ++s) { p->ms_acc[s] = 0; p->launches[s] = 0; }
    return rc;
}

extern "C" int tron_device_count(int *count)
{
    if (!count) return fail(TRON_ERR_INVALID, "null argument");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return fail(TRON_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return TRON_OK;
}

extern "C" int tron_device_malloc(void **d_ptr, size_t bytes)
{
    if (!d_ptr) return fail(TRON_ERR_INVALID, "null argument");
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1));
    return TRON_OK;
}

extern "C" int tron_device_free(void *d_ptr)
{
    HIP_TRY(hipFree(d_ptr));
    return TRON_OK;
}

extern "C" int tron_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes)
{
    HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return TRON_OK;
}

extern "C" int tron_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes)
{
    HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return TRON_OK;
}
