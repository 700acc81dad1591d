// The batched pipelines of libtronhip on the device: adjoint (gridding -> FFT -> crop / deapodise / combine), forward
// (pad / deapodise -> FFT -> degridding), CGNR on top of the two, and their work buffers.  Replaces
// tron_nufft_adj_radial2d / tron_nufft_radial2d / tron_cgnr_radial2d of the reference (src/tron.cu:623-720).
#include "tron_plan_impl.h"

namespace tron {

int drain_timers(tron_plan *p)
{
    HIP_TRY(hipStreamSynchronize(p->stream));
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
        std::call_once(g_fft_once, [] { rocfft_setup(); });
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
        // ... and its transform kernels are loaded lazily at their first launch, an upload that a non-blocking stream does not
        // wait for either (the same race the warm_* launches close for this library's own code objects; on a
        // fresh box the first process of a test run lost it about one time in three: a wrong image or a memory fault in a rocFFT
        // size).  The first transform of every (size, direction) of a process therefore runs on the NULL stream, batch 1, on scratch.
        {
            static std::mutex warm_mu;
            static std::set<std::tuple<int, int, int, int>> warmed;
            std::lock_guard<std::mutex> lock(warm_mu);
            const auto wkey = std::make_tuple(p->cfg.device, p->d.nxos, p->d.nyos, inverse);
            if (!warmed.count(wkey)) {
                rocfft_plan wp = nullptr;
                rocfft_execution_info wi = nullptr;
                void *scratch = nullptr, *wwork = nullptr;
                size_t wbytes = 0;
                bool ok = rocfft_plan_create(&wp, rocfft_placement_inplace,
                                             inverse ? rocfft_transform_type_complex_inverse : rocfft_transform_type_complex_forward,
                                             rocfft_precision_single, 2, lengths, 1, nullptr) == rocfft_status_success &&
                          rocfft_execution_info_create(&wi) == rocfft_status_success &&
                          rocfft_execution_info_set_stream(wi, nullptr) == rocfft_status_success &&
                          rocfft_plan_get_work_buffer_size(wp, &wbytes) == rocfft_status_success &&
                          hipMalloc(&scratch, lengths[0] * lengths[1] * sizeof(float2)) == hipSuccess &&
                          hipMemset(scratch, 0, lengths[0] * lengths[1] * sizeof(float2)) == hipSuccess;
                if (ok && wbytes) ok = hipMalloc(&wwork, wbytes) == hipSuccess && rocfft_execution_info_set_work_buffer(wi, wwork, wbytes) == rocfft_status_success;
                if (ok) {
                    void *bufs[1] = {scratch};
                    ok = rocfft_execute(wp, bufs, nullptr, wi) == rocfft_status_success;
                }
                const hipError_t se = hipDeviceSynchronize();
                if (wi) rocfft_execution_info_destroy(wi);
                if (wp) rocfft_plan_destroy(wp);
                hipFree(scratch);
                hipFree(wwork);
                if (!ok || se != hipSuccess) return fail(TRON_ERR_FFT, "rocFFT warm-up transform %dx%d failed", p->d.nxos, p->d.nyos);
                warmed.insert(wkey);
            }
        }
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

int stage_check(tron_plan *p, const char *what)
{
    if (!p->sync_each) return TRON_OK;
    hipError_t e = hipStreamSynchronize(p->stream);
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
    for (float2 **b : {&p->d_grid, &p->d_fft_tmp})
        if (*b) { HIP_TRY(hipFree(*b)); *b = nullptr; }
    p->work_units = 0;
    if (hipMalloc(reinterpret_cast<void **>(&p->d_grid), (size_t)units * per_unit) != hipSuccess)
        return fail(TRON_ERR_NOMEM, "cannot allocate %zu bytes of Cartesian work space", (size_t)units * per_unit);
    if (p->poison) {                                            // (on the plan's stream: see the note at arc_prep_kernel's launch in tron_plan.cpp)
        HIP_TRY(hipMemsetAsync(p->d_grid, 0xff, (size_t)units * per_unit, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    if (p->fft512 && hipMalloc(reinterpret_cast<void **>(&p->d_fft_tmp), (size_t)units * p->nchan * 256 * 512 * sizeof(float2)) != hipSuccess)
        return fail(TRON_ERR_NOMEM, "cannot allocate the FFT intermediate buffer");
    p->work_units = units;
    return TRON_OK;
}

// in_stride_spokes: spokes between the windows of consecutive slices in d_in_z0 (0 = prof_slide: views into the stream)
// One stream: gridding and the FFT passes of a batch follow each other on the plan's stream.  (Rounds 1-5 ran the FFT passes of
// batch k beside the gridding of batch k + 1 on a second stream with a second work grid: 3.67 ms of kernels in 3.53 ms per step, and
// the same bench line with the lanes serialised -- profiles/round5_bench_one_lane.json, round6_bench_one_lane.json; removed in round 6.)
int adjoint_run_raw(tron_plan *p, void *d_out, const void *d_in_z0, int zfirst, int zcount, int combine, int in_stride_spokes,
                    float out_scale = 1.f, double *norm_partial = nullptr, int *norm_parts = nullptr)
{
    const tron_dims &d = p->d;
    const size_t n2 = (size_t)d.nxos * d.nxos;
    const size_t elem = p->cfg.input_half ? 4 : 8;
    const int golden = p->cfg.golden_angle;
    const TrajTables &T = traj_cur(p);
    // equal batches: a short last launch would be bound by the centre tile's serial chain (e.g. 128 slices = 64 + 64, not 85 + 43)
    // (the work buffers hold 1.5 x chunk so that the batches can be evened out upwards)
    int nbatch = std::max(1, (zcount + p->chunk / 2) / p->chunk);
    if ((zcount + nbatch - 1) / nbatch > p->chunk_cap) nbatch = (zcount + p->chunk_cap - 1) / p->chunk_cap;
    const int step = (zcount + nbatch - 1) / nbatch;
    if (int erc = ensure_work(p, std::min(step, std::max(zcount, 1)))) return erc;
    hipStream_t st = p->stream;
    for (int z0 = 0; z0 < zcount; z0 += step) {
        const int cz = std::min(step, zcount - z0);
        float2 *grid_buf = p->d_grid;
        float2 *tmp_buf = p->d_fft_tmp;
        GridParams g;
        memset(&g, 0, sizeof(g));
        fill_grid_consts(p, g);
        const int in_stride = in_stride_spokes > 0 ? in_stride_spokes : d.prof_slide;
        g.nudata = static_cast<const unsigned char *>(d_in_z0) + (size_t)z0 * in_stride * d.nro * p->nchan * elem;
        g.udata = grid_buf;
        g.trig = T.d_trig + (golden ? (size_t)(zfirst + z0) * d.prof_slide : 0);
        g.in_slice_stride = (long long)in_stride * d.nro * p->nchan;
        g.trig_slice_stride = golden ? d.prof_slide : 0;
        g.nslices = cz;
        g.apply_dcf = 1;
        g.out_z = (long long)p->nchan * n2;
        g.out_c = (long long)n2;
        g.out_p = 1;
        g.out_shift = 1;
        // the fused FFT never reads beyond the sampled disc, so the gridding kernel need not store zeros there
        const int rzero = p->fft512 ? (int)floorf((float)(d.nxos / 2 - 1) + p->cfg.kernwidth) + 1 : 0;
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
                }
                int relief_parts = p->relief_parts;
                // every tile but the samples |r| < inner_r0 on the arc / scatter kernel, those on the centre kernel behind it -- when this set's run
                // tables are usable and cover the slices asked for; the binned kernel (inner tile in parts + reduce pass) otherwise
                const bool use_arc = arc_ready(p) && p->relief_entries > 0 && vs <= 1 && (reinterpret_cast<uintptr_t>(g.nudata) & 15) == 0
                                     && (!golden || (zfirst + z0 >= p->share_z0 && zfirst + z0 + cz <= p->share_z0 + p->share_nz));
                if (p->relief_entries > 0) {
                    // the samples next to the k-space centre go to the inner tile's workgroups; their parts are added
                    // onto the centre tiles by the reduce pass that follows the gridding kernel on this stream
                    const bool small = cz < 32 && p->relief_entries_small > 0 && vs <= 1;     // more, shorter inner-tile workgroups (16 slices: +9 %; 32+: 0)
                    relief_parts = small ? p->relief_parts_small : p->relief_parts;
                    const size_t need = (size_t)cz + (vs > 1 ? vs : 0);               // whole slice groups
                    const int parts_cap = std::max(p->relief_parts, p->relief_parts_small);
                    if (!use_arc && p->relief_slices < need) {       // (the inner tile's parts: the binned kernel's only)
                        if (p->d_relief_partial) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(p->d_relief_partial)); p->d_relief_partial = nullptr; }
                        const size_t want = std::max(need, (size_t)std::min(step, 64) + 8);
                        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_relief_partial),
                                          want * parts_cap * p->nchan * kBinnedTile * kBinnedTile * sizeof(float2)));
                        p->relief_slices = want;
                    }
                    g.tile_order = small ? p->d_tile_order32_relief_small : p->d_tile_order32_relief;
                    g.tile_entries = small ? p->relief_entries_small : p->relief_entries;
                    g.nsplit_slots = 1;
                    g.max_parts = relief_parts;
                    g.split_slots = small ? p->d_relief_slots_small : p->d_relief_slots;
                    g.partial = p->d_relief_partial;
                    g.inner_r0 = p->relief_r0;
                } else if (vs <= 1 && cz < p->split_below && p->nsplit_slots > 0) {
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
                if (use_arc) {
                    const size_t win0 = golden ? (size_t)(zfirst + z0 - p->share_z0) : 0;     // the run tables start at the plan's first slice
                    const size_t nt32 = (size_t)(d.nxos / kBinnedTile) * (d.nxos / kBinnedTile);
                    g.arc_cap = p->arc_cap;
                    g.arc_nrec = p->arc_nrec;
                    g.arc_slice_stride = golden ? 1 : 0;
                    g.kb_lut = p->d_kb_lut;
                    g.lut_entries = p->lut_entries;
                    g.lut_scale = p->lut_scale;
                    g.lut_bias = p->lut_bias;
                    // slices one workgroup grids in turn (tile geometry and the window table are set up once per workgroup)
                    g.arc_zper = p->arc_zper > 0 ? p->arc_zper : (cz >= 64 ? 4 : (cz >= 32 ? 2 : 1));
                    for (int q = 0; q < p->arc_passes; ++q) {      // (one pass unless a window holds more than kArcMaxNpe spokes)
                        GridParams ga = g;
                        const size_t tab = (size_t)q * p->arc_nwin + win0;
                        const size_t nt_tab = p->arc_ntiles;
                        ga.arc_hdr = T.d_arc_hdr + tab * nt_tab;
                        ga.arc_ent = T.d_arc_ent + tab * p->arc_cap;
                        ga.arc_win = T.d_arc_win ? T.d_arc_win + tab * nt32 * 256 : nullptr;
                        ga.arc_off = T.d_arc_off ? T.d_arc_off + tab * p->arc_cap : nullptr;
                        ga.scat_tile = p->scat_tile;
                        ga.arc_rec = T.d_arc_rec ? T.d_arc_rec + tab * (size_t)p->arc_rec_cap * 80 : nullptr;
                        ga.arc_rbase = T.d_arc_rbase ? T.d_arc_rbase + tab * nt_tab : nullptr;
                        ga.arc_rec_cap = p->arc_rec_cap;
                        ga.npe = std::min(d.npe1work, (q + 1) * p->arc_pass_npe) - q * p->arc_pass_npe;
                        ga.arc_accumulate = q > 0;
                        ga.scat_wsum = p->scat_wsum;
                        ga.scat_wmax = p->scat_wmax;
                        if (p->scatter && p->scat_tile == 64) {
                            ga.tile_order = p->d_tile_order64;          // (its own list of 64-tiles; no inner-tile entries in front)
                            HIP_TRY(launch_grid_scatter(ga, p->cfg.input_half, 0, st));
                        } else if (p->scatter) HIP_TRY(launch_grid_scatter(ga, p->cfg.input_half, relief_parts, st));
                        else HIP_TRY(launch_grid_arc(ga, p->cfg.input_half, relief_parts, st));
                    }
                    const size_t woff = win0 * (size_t)d.npe1work;
                    g.cen_order = T.d_order + woff;
                    g.cen_win = T.d_cen_win + win0 * (size_t)p->cen_nblocks;
                    g.cen_cs = T.d_cs + woff;
                    const bool parts = cz < p->cen_parts_below && p->cen_nheavy > 0;      // small launch: the busy blocks in parts
                    g.cen_grec = parts ? p->d_cen_grec_parts : p->d_cen_grec;
                    g.cen_ticket = p->d_cen_ticket;
                    g.cen_parts = p->d_cen_parts;
                    g.cen_nblocks = p->cen_nblocks;
                    g.cen_nheavy = parts ? p->cen_nheavy : 0;
                    g.cen_ngroups = parts ? p->cen_nunits_parts : p->cen_ngroups;
                    HIP_TRY(launch_grid_centre(g, p->cfg.input_half, st));
                } else {
                    HIP_TRY(launch_grid_binned(g, p->cfg.input_half, st));
                }
            } else {
                HIP_TRY(launch_grid(g, p->kb_mode, p->cfg.input_half, st));
            }
        }
        int rc = stage_check(p, "grid");
        if (rc) return rc;
        if (p->fft512 && combine) {
            // fused: pruned inverse FFT + crop + deapodise + root-sum-of-squares (tron_fft512.hip)
            {
                StageTimer t(p, STAGE_FFT, st);
                HIP_TRY(launch_fft512_adjoint(grid_buf, tmp_buf, static_cast<float2 *>(d_out) + (size_t)z0 * d.nx * d.nx,
                                              p->d_tw512, p->d_deapod, rzero, p->nchan, cz, st));
            }
            if ((rc = stage_check(p, "fft512"))) return rc;
            continue;
        }
        if (p->fft512) {
            // fused, uncombined: pruned inverse FFT + crop + deapodise per coil image, interleaved by coil (CGNR, Walsh, nt > 1)
            StageTimer t(p, STAGE_FFT, st);
            const int parts = fft512_coils_partials(cz) * p->nchan;
            if (norm_parts) *norm_parts = parts;
            HIP_TRY(launch_fft512_adjoint_coils(grid_buf, tmp_buf, static_cast<float2 *>(d_out) + (size_t)z0 * d.nx * d.nx * p->nchan,
                                                p->d_tw512, p->d_deapod, rzero, p->nchan, cz, out_scale,
                                                norm_partial ? norm_partial + (size_t)z0 * parts : nullptr, st));
            if ((rc = stage_check(p, "fft512 coils"))) return rc;
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
        {
            StageTimer t(p, STAGE_POST);
            HIP_TRY(launch_post(q, p->stream));
        }
        if ((rc = stage_check(p, "post"))) return rc;
    }
    return TRON_OK;
}

// trig / deapod default to the plan's own tables (forward plans); the CGNR path of an adjoint plan passes the forward
// operator's tables and a per-image angle stride (every slice has its own golden angles)
// coilcombinesos / coilcombinewalsh (src/tron.cu:764,766) of `cz` slices of coil images [z][nchan*id + c]
int combine_coils(tron_plan *p, float2 *d_out, const float2 *d_coil, int cz)
{
    HIP_TRY(launch_coil_combine(d_out, d_coil, p->d.nx, p->d.nc, p->d.nt, p->cfg.coil_combine == 1 ? 1 : 0,
                                std::max(0, p->cfg.walsh_patch), cz, p->stream));
    return TRON_OK;
}

// The adjoint with the plan's coil combination.  Root-sum-of-squares of one repetition is fused into the pipeline's
// tail; Walsh's adaptive combination and nt > 1 (channel = coil + nc*repetition) run it uncombined into a scratch
// buffer and combine from there.
int adjoint_run(tron_plan *p, void *d_out, const void *d_in_z0, int zfirst, int zcount, int combine, int in_stride_spokes,
                float out_scale, double *norm_partial, int *norm_parts)
{
    if (int trc = traj_turn(p)) return trc;
    if (!combine || (p->d.nt == 1 && p->cfg.coil_combine != 1))
        return adjoint_run_raw(p, d_out, d_in_z0, zfirst, zcount, combine, in_stride_spokes, out_scale, norm_partial, norm_parts);
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
                                 zfirst + z0, cz, 0, in_stride_spokes);
        if (rc) return rc;
        if ((rc = combine_coils(p, static_cast<float2 *>(d_out) + (size_t)z0 * d.nt * d.nx * d.ny, p->d_coil_tmp, cz))) return rc;
    }
    return TRON_OK;
}

int forward_run(tron_plan *p, void *d_out, const void *d_in, int nimg, const float2 *trig, int trig_img_stride,
                const float *deapod)
{
    if (int trc = traj_turn(p)) return trc;
    const tron_dims &d = p->d;
    const size_t n2 = (size_t)d.nxos * d.nyos;
    const bool square = d.nx == d.ny;
    if (!trig) trig = traj_cur(p).d_trig;
    if (!deapod) deapod = p->d_deapod;
    if (int erc = ensure_work(p, std::max(1, std::min(p->chunk, nimg)))) return erc;
    for (int k0 = 0; k0 < nimg; k0 += p->chunk) {
        const int ck = std::min(p->chunk, nimg - k0);
        const float2 *img = static_cast<const float2 *>(d_in) + (size_t)k0 * p->nchan * d.nx * d.ny;
        // the fused FFT stores every grid line rotated by the streaming degridder's halo (tuning knob TRON_GRID_ROT)
        static const int rot_env = tuning_env("TRON_GRID_ROT") ? atoi(tuning_env("TRON_GRID_ROT")) : -1;
        const int grid_rot = rot_env >= 0 ? rot_env : (((int)ceilf(p->cfg.kernwidth) + 1) & ~1);
        if (p->fft512) {
            // fused: pad + deapodise + shift + pruned forward FFT (tron_fft512.hip)
            StageTimer t(p, STAGE_FFT);
            // samples lie within nxos/2 of the centre, their footprints (zero-weight slots included) within W + 2 more
            const int rzero = d.nxos / 2 + (int)ceilf(p->cfg.kernwidth) + 4;
            HIP_TRY(launch_fft512_forward(img, p->d_fft_tmp, p->d_grid, p->d_tw512, deapod, rzero, grid_rot, p->nchan, ck, p->stream));
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
        g.tile_order = square ? p->d_tile_order32 : nullptr;     // centre-first tile order for square grids, raster otherwise
        g.in_z = (long long)p->nchan * n2;
        g.in_c = (long long)n2;
        g.in_p = 1;
        g.in_shift = 1;
        g.in_transposed = p->fft512 ? 1 : 0;      // launch_fft512_forward stores the grid transposed
        g.in_rot = p->fft512 ? grid_rot : 0;
        g.n = d.nxos;
        g.nrows = square ? 0 : d.nyos;
        g.nrep = p->nchan;
        g.nro = d.nro;
        g.npe = d.npe1work;
        g.nimg = ck;
        g.W = p->cfg.kernwidth;
        g.beta = p->beta;
        memcpy(g.kb_poly, p->kb_poly, sizeof(g.kb_poly));
        memcpy(g.group_end, p->dg_group_end, sizeof(g.group_end));
        g.group_max = (square && !p->degrid_tile_only) ? std::min(16, ck / 4) : 0;   // runs of images only where they leave enough workgroups
        {
            StageTimer t(p, STAGE_DEGRID);
            const bool stream = !p->degrid_simple && degrid_stream_supported(g, p->kb_mode);
            p->last_degrid_kernel = stream ? "degrid_stream_kernel" : p->degrid_simple ? "degrid_kernel" : "degrid_tile_kernel";
            if (stream)
                HIP_TRY(launch_degrid_stream(g, p->kb_mode, p->stream)); // many images on a grid of whole tiles
            else if (!p->degrid_simple)
                HIP_TRY(launch_degrid_tile(g, p->kb_mode, p->stream));  // every width the plan accepts (W <= 4), square or not
            else
                HIP_TRY(launch_degrid(g, p->kb_mode, p->stream));       // TRON_DEGRID_KERNEL=simple: the thread-per-sample audit kernel
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
    if (int trc = traj_turn(p)) return trc;
    const tron_dims &d = p->d;
    const TrajTables &T = traj_cur(p);
    if (p->cfg.input_half) return fail(TRON_ERR_UNSUPPORTED, "CGNR needs complex64 k-space (the residual lives in fp32)");
    const size_t n = (size_t)p->nchan * d.nro * d.npe1work;          // data-space elements per slice
    const size_t N = (size_t)p->nchan * d.nx * d.ny;                 // image-space elements per slice (F2)
    const size_t spoke_bytes = (size_t)d.nro * p->nchan * sizeof(float2);
    const int step = std::max(1, std::min(p->chunk, zcount));
    // partial sums per slice: kCgPartials from the norm kernels, up to 64 column blocks x nchan from the fused FFT tail
    const int parts_cap = std::max(kCgPartials, 64 * p->nchan);
    if (p->cg_slices < step || p->cg_parts < parts_cap) {
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
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_partial), (size_t)step * parts_cap * sizeof(double)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_num), step * sizeof(double)));
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cg_coef), 2 * (size_t)step * sizeof(float)));
        p->cg_slices = step;
        p->cg_parts = parts_cap;
    }
    const float unscale = (float)d.nxos * (float)d.npe1work;          // F3: the gridding kernel's 1/nxos/npe (src/tron.cu:532) divided out
    const int golden = p->cfg.golden_angle;
    hipStream_t st = p->stream;
    float *alpha = p->d_cg_coef, *beta = p->d_cg_coef + p->cg_slices;
    int rc;
    // ztilde = A^H W r, times `unscale`, and its squared norm per slice (mode 0: stored; mode 2: beta, :709).  On the fused
    // 512 / 256 path the FFT tail scales and leaves the partial sums; otherwise one pass over ztilde does both.
    auto residual_image = [&](int z0, int cz, int mode) -> int {
        int parts = 0;
        int r2 = adjoint_run(p, p->d_cg_zt, p->d_cg_r, zfirst + z0, cz, 0, d.npe1work, unscale, p->d_cg_partial, &parts);
        if (r2) return r2;
        if (parts == 0) {
            HIP_TRY(launch_cg_scale_norm2(p->d_cg_zt, N, cz, unscale, p->d_cg_partial, st));
            parts = kCgPartials;
        }
        HIP_TRY(launch_cg_finish(p->d_cg_partial, parts, p->d_cg_num, beta, mode, cz, st));
        return TRON_OK;
    };
    for (int z0 = 0; z0 < zcount; z0 += step) {
        const int cz = std::min(step, zcount - z0);
        const unsigned char *y = static_cast<const unsigned char *>(d_in_z0) + (size_t)z0 * d.prof_slide * spoke_bytes;
        // r = y: the (overlapping) windows of the stream, one contiguous copy per slice (:685; F5) -- one launch for the batch
        HIP_TRY(launch_cg_windows(p->d_cg_r, reinterpret_cast<const float2 *>(y), n, (size_t)d.prof_slide * d.nro * p->nchan, cz, st));
        // ztilde = A^H W r (:686), ptilde = ztilde (:687), x = 0 (:683)
        if ((rc = residual_image(z0, cz, 0))) return rc;
        HIP_TRY(hipMemcpyAsync(p->d_cg_pt, p->d_cg_zt, cz * N * sizeof(float2), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemsetAsync(p->d_cg_x, 0, cz * N * sizeof(float2), st));
        // the forward operator's angles: the slice's own index range (F4); golden angles are shared with the adjoint's table
        const float2 *trig = golden ? T.d_trig + (size_t)(zfirst + z0) * d.prof_slide : (p->d_trig_fwd ? p->d_trig_fwd : T.d_trig);
        const int trig_stride = golden ? d.prof_slide : 0;
        for (int t = 0; t < p->cfg.niter; ++t) {
            if ((rc = forward_run(p, p->d_cg_v, p->d_cg_pt, cz, trig, trig_stride, p->d_deapod_fwd))) return rc;   // v = A ptilde (:691)
            // alpha = |ztilde|^2 / <W v, v> (:693-697; F1)
            int wparts = 0;
            HIP_TRY(launch_cg_wnorm2(p->d_cg_v, n, cz, p->nchan, d.nro, p->dcf_a, p->dcf_b, p->d_cg_partial, p->cg_parts, &wparts, st));
            HIP_TRY(launch_cg_finish(p->d_cg_partial, wparts, p->d_cg_num, alpha, 1, cz, st));
            if (t == p->cfg.niter - 1) {                                                                           // (:701)
                HIP_TRY(launch_cg_update(p->d_cg_x, p->d_cg_pt, p->d_cg_zt, alpha, beta, N, cz, 1, st));            // x += alpha ptilde (:699)
                break;
            }
            HIP_TRY(launch_cg_axpy(p->d_cg_r, p->d_cg_v, alpha, -1.f, n, cz, st));                                  // r -= alpha v (:703)
            if ((rc = residual_image(z0, cz, 2))) return rc;                                                       // ztilde = A^H W r (:707), beta (:709; F1)
            // x += alpha ptilde (:699) and ptilde = ztilde + beta ptilde (:710) in one pass over ptilde
            HIP_TRY(launch_cg_update(p->d_cg_x, p->d_cg_pt, p->d_cg_zt, alpha, beta, N, cz, 0, st));
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
        HIP_TRY(hipMemsetAsync(p->d_errflag, 0, sizeof(flag), p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        return fail(TRON_ERR_HIP, "gridding kernel reported an internal overflow (flag %u)", flag);
    }
    return TRON_OK;
}

}  // namespace tron
