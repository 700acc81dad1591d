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
// (The pipelines themselves live in tron_pipeline.cpp, the host-buffer entry points in tron_hostio.cpp.)
#include "tron_plan_impl.h"

#include <fcntl.h>
#include <time.h>
#include <unistd.h>

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
}  // namespace tron

using namespace tron;

namespace tron {
std::once_flag g_fft_once;   // rocfft_setup() on the first rocFFT plan (get_fft): the fused 512 / 256 path never pays for it
}  // namespace tron

extern "C" const char *tron_last_error(void) { return g_last_error.c_str(); }

extern "C" int tron_plan_create_times(const tron_plan *plan, double seconds[5])
{
    if (!plan || !seconds) return fail(TRON_ERR_INVALID, "tron_plan_create_times: null argument");
    for (int i = 0; i < 5; ++i) seconds[i] = plan->create_s[i];
    return TRON_OK;
}

extern "C" const char *tron_version(void) { return "tronhip 0.1 (gfx950)"; }

// Everything of an arc / scatter plan that follows from (p->scatter, p->scat_tile, p->relief_r0) and not from the spoke angles: the centre
// kernel's block lists, the sizes of the run tables (TrajTables).  Called again when the first build of the tables overflows and the plan
// takes the next formulation down (64-tiles -> 32-tiles -> arc kernel -> binned kernel).
static int arc_setup(tron_plan *p, const std::vector<uint32_t> &band)
{
    const tron_config *cfg = &p->cfg;
    const tron_dims &d = p->d;
    int rc = TRON_OK;
    for (void **q : {(void **)&p->d_cen_gwin, (void **)&p->d_cen_grec, (void **)&p->d_cen_grec_parts, (void **)&p->d_cen_ticket, (void **)&p->d_cen_parts,
                     (void **)&p->d_tile_order64})
        if (*q) { hipFree(*q); *q = nullptr; }
    p->arc_rec_cap = 0;
    {
        // The sizes of the angle-dependent tables (TrajTables: built below, once everything they read is in place, and again by
        // tron_plan_retarget) and everything the centre kernel needs that does NOT depend on the angles.
        const size_t nwin = cfg->golden_angle ? (size_t)p->share_nz : 1;
        const int npe = d.npe1work, nt32 = (d.nxos / kBinnedTile) * (d.nxos / kBinnedTile);
        {   // centre kernel: the 2x2 blocks of the origin-centred 32 x 32 square that a sample |r| < inner_r0 can reach, nearest the
            // origin (most spokes) first
            std::vector<int> groups;
            const float reach = (float)(p->relief_r0 - 1) + cfg->kernwidth + 1.0f;
            auto d2 = [](int g) { const float x = 2.f * (g & 255) - 15.f, y = 2.f * (g >> 8) - 15.f; return x * x + y * y; };   // block centre
            for (int j = 0; j < 16; ++j)
                for (int i = 0; i < 16; ++i) {
                    const float ax = fabsf(2.f * i - 15.f) - 0.5f, ay = fabsf(2.f * j - 15.f) - 0.5f;                           // its nearest point
                    if (ax * ax + ay * ay <= reach * reach) groups.push_back(i | (j << 8));
                }
            std::stable_sort(groups.begin(), groups.end(), [&](int a, int b) { return d2(a) < d2(b); });
            // per block: the band of its four points (src/tron.cu:498-502) as masks over |r| and the largest |r| < inner_r0
            // inside it; a block no such sample reaches is dropped
            std::vector<uint32_t> grec;
            {
                std::vector<int> kept;
                const int n = d.nxos, h = n / 2;
                for (int g : groups) {
                    const int X0 = 2 * (g & 255) - 16, Y0 = 2 * (g >> 8) - 16;
                    uint32_t bm[4];
                    int bandhi = -1, bandlo = 1 << 20;
                    for (int q = 0; q < 4; ++q) {
                        const int X = X0 + (q & 1), Y = Y0 + (q >> 1);
                        const uint32_t bnd = band[(size_t)(Y + h) * n + (X + h)];
                        const int lo = (int)(bnd & 0xffffu), hi = std::min((int)(bnd >> 16), 31);
                        bm[q] = 0u;
                        if (lo <= hi) {
                            bm[q] = (0xffffffffu >> (31 - (hi - lo))) << lo;
                            bandhi = std::max(bandhi, hi);
                            bandlo = std::min(bandlo, lo);
                        }
                    }
                    const int rcap = std::min(p->relief_r0 - 1, bandhi);
                    if (bandlo > rcap) continue;
                    kept.push_back(g);
                    const uint32_t rec[8] = {(uint32_t)g, (uint32_t)rcap, 0u, 0u, bm[0], bm[1], bm[2], bm[3]};
                    grec.insert(grec.end(), rec, rec + 8);
                }
                groups.swap(kept);
            }
            // the angular window of every block's run of a window's sorted list (geometry only); the runs themselves are TrajTables'
            std::vector<float> gwin(4 * groups.size());
            build_centre_group_windows(groups.data(), (int)groups.size(), cfg->kernwidth, gwin.data());
            if ((rc = upload(&p->d_cen_gwin, gwin.data(), gwin.size() * sizeof(float)))) return rc;
            // A block next to the origin meets every spoke, one at the rim 50: a busy block is worked on in up to four PARTS (runs
            // of its window, ~128 spokes each), so that no work item is much longer than the others -- the longest one is the
            // floor under a launch (40 us for a whole window of 400 spokes, against 55 us for a launch of 32 slices).
            // Launches of fewer than 64 slices only (cen_parts_below): 32 slices 56 -> 41 us; a launch of 128 has enough
            // items to hide its longest ones and pays for the parts' hand-over instead (115 -> 122 us), so it takes whole blocks.
            // (Parts by the EXPECTED share of a window's spokes, gwin[4 g + 3]: the same lists whatever the angles, so a retargeted
            //  plan sums in the order a fresh one would.  Until round 5 the mean over the plan's windows decided.)
            const int nblocks = (int)groups.size();
            std::vector<uint32_t> units, whole;
            int nheavy = 0;
            for (int g = 0; g < nblocks; ++g) {
                const double mean = (double)gwin[4 * (size_t)g + 3] * npe;
                const int per_part = 200;
                const int parts = std::max(1, std::min(4, (int)ceil(mean / per_part)));
                const int heavy = parts > 1 ? nheavy++ : 0;
                const uint32_t *r = &grec[8 * (size_t)g];
                for (int q = 0; q < parts; ++q) {
                    const uint32_t rec[8] = {r[0], r[1], (uint32_t)q | ((uint32_t)parts << 8) | ((uint32_t)heavy << 16), (uint32_t)g, r[4], r[5], r[6], r[7]};
                    units.insert(units.end(), rec, rec + 8);
                }
                const uint32_t rec[8] = {r[0], r[1], 1u << 8, (uint32_t)g, r[4], r[5], r[6], r[7]};
                whole.insert(whole.end(), rec, rec + 8);
            }
            p->cen_nblocks = nblocks;
            p->cen_nheavy = nheavy;
            p->cen_ngroups = nblocks;
            p->cen_nunits_parts = (int)(units.size() / 8);
            if ((rc = upload(&p->d_cen_grec, whole.data(), whole.size() * sizeof(uint32_t)))) return rc;
            if ((rc = upload(&p->d_cen_grec_parts, units.data(), units.size() * sizeof(uint32_t)))) return rc;
            const size_t slots = (size_t)p->chunk_cap * (size_t)(p->nchan <= 4 ? 1 : (p->nchan + 7) / 8) * (size_t)nheavy;
            if (hipMalloc(reinterpret_cast<void **>(&p->d_cen_ticket), (8 * 16 + slots) * sizeof(unsigned)) != hipSuccess ||
                (slots > 0 && hipMalloc(reinterpret_cast<void **>(&p->d_cen_parts), slots * 4 * 64 * sizeof(float)) != hipSuccess))
                return (fail(TRON_ERR_NOMEM, "centre kernel work counters"));
        }
        // Windows of more than kArcMaxNpe spokes: the run tables are built, and the arc kernel run, once per PASS over the spokes
        // [q sub, (q + 1) sub) of every window (each pass's list = the sorted list with the other spokes left out); the passes
        // after the first add to the grid.  The centre kernel takes the whole window at once.
        const int npass = (npe + kArcPassNpe - 1) / kArcPassNpe, sub = (npe + npass - 1) / npass;
        p->arc_passes = npass;
        p->arc_nwin = nwin;
        p->arc_pass_npe = sub;
        // a spoke crosses at most 2 * nxos / 32 + 3 tiles (+ their halos): 56 run entries per spoke bound every window
        p->arc_cap = sub * (2 * (d.nxos / kBinnedTile) + 24);
        p->arc_nrec = p->scatter ? 32767 : grid_arc_nrec(p->nchan, cfg->input_half);      // (scatter kernel: one batch per run)
        // Scatter kernel: 64 x 64 tiles where the grid's centre is a corner of four of them, the channels go one per pass (two channels
        // per pass would need 100 KB of sums) and a centre tile's run -- ~0.6 of a window's spokes -- fits the 512 run entries
        p->scat_tile = 32;
        if (p->scatter && p->nchan != 2 && (d.nxos / 2) % 64 == 0 && d.nxos >= 256 && sub <= 800 && p->scat_tile_max >= 64) p->scat_tile = 64;
        if (const char *e = tuning_env("TRON_SCAT_TILE")) { if (atoi(e) == 32 || (atoi(e) == 64 && (d.nxos / 2) % 64 == 0 && p->nchan != 2 && p->scat_tile_max >= 64)) p->scat_tile = atoi(e); }
        p->arc_ntiles = p->scatter && p->scat_tile == 64 ? (size_t)(d.nxos / 64) * (d.nxos / 64) : (size_t)nt32;
        if (p->scatter && p->scat_tile == 64) {
            std::vector<int> o64;
            build_tile_order(d.nxos, 64, o64);
            if ((rc = upload(&p->d_tile_order64, o64.data(), o64.size() * sizeof(int)))) return rc;
        }
        // scatter kernel's member tables: one byte per record of every run + 16 bits per group of 64 records.  A spoke holds nxos - 1
        // radii (whatever its readout length), a record lies in 1.13 (64-tiles) to 1.27 (32-tiles) runs on average, a little more where few
        // spokes make the corner segments count, every run ends on a partly filled group (+ slack: the kernel copies whole rounds)
        if (p->scatter) p->arc_rec_cap = (int)(((size_t)sub * d.nxos * 3 / 2) / 64 + p->arc_ntiles + 16);
    }
    return rc;
}

extern "C" int tron_plan_create(tron_plan **out, const tron_config *cfg, const tron_dims *dims)
{
    if (!out || !cfg || !dims) return fail(TRON_ERR_INVALID, "tron_plan_create: null argument");
    return tron::plan_create_share(out, cfg, dims, 0, dims->nz);
}

int tron::plan_create_share(tron_plan **out, const tron_config *cfg, const tron_dims *dims, int share_z0, int share_nz)
{
    if (!out || !cfg || !dims) return fail(TRON_ERR_INVALID, "tron_plan_create: null argument");
    *out = nullptr;
    // Test hook (TRON_TUNING=1 only): TRON_DEBUG=cold_fault=<path> makes the FIRST plan creation on a box fail -- the process that
    // finds <path> missing creates it and fails; every later one runs normally.  That is the shape of the cold-start faults of
    // DESIGN.md 4.5 (first GPU process on a fresh box), and tests/test_gpu_cold_start.py uses it to prove that such a fault is seen.
    // (O_CREAT | O_EXCL: exactly one creator fails, however many plans -- tron_recon_radial2d_multi's workers -- are created at once)
    std::string mark;
    if (debug_token("cold_fault", &mark) && !mark.empty()) {
        const int fd = open(mark.c_str(), O_CREAT | O_EXCL | O_WRONLY, 0644);
        if (fd >= 0) {
            close(fd);
            return fail(TRON_ERR_HIP, "injected cold-start fault (TRON_DEBUG=cold_fault=%s)", mark.c_str());
        }
    }
    const tron_dims &d = *dims;
    if (share_z0 < 0 || share_nz < 0 || share_z0 + share_nz > std::max(d.nz, 1))
        return fail(TRON_ERR_INVALID, "slice share [%d,%d) outside [0,%d)", share_z0, share_z0 + share_nz, d.nz);
    // coil combination (src/tron.cu:764-766): fail here, not after the first batch has been queued
    if (cfg->coil_combine != 0 && cfg->coil_combine != 1)
        return fail(TRON_ERR_INVALID, "coil_combine %d: 0 (root sum of squares) or 1 (Walsh)", cfg->coil_combine);
    if (cfg->coil_combine == 1 && (d.nc > 16 || cfg->walsh_patch < 0 || cfg->walsh_patch > 16))
        return fail(TRON_ERR_UNSUPPORTED, "Walsh coil combination handles up to 16 coils and patch half-widths 0..16 (nc=%d, walsh_patch=%d)",
                    d.nc, cfg->walsh_patch);
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
    struct timespec pt0;
    clock_gettime(CLOCK_MONOTONIC, &pt0);
    auto since = [&pt0]() {
        struct timespec b;
        clock_gettime(CLOCK_MONOTONIC, &b);
        return (b.tv_sec - pt0.tv_sec) + 1e-9 * (b.tv_nsec - pt0.tv_nsec);
    };
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(TRON_ERR_HIP, "device %d requested but %d HIP device(s) present", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    (void)hipGetLastError();       // a stale error of an earlier, failed call must not be reported by this one
    HIP_TRY(hipFree(nullptr));     // (the runtime's own start-up, timed apart from the code objects)
    const double t_hipinit = since();
    // make every code object resident before anything is queued on a non-blocking stream
    HIP_TRY(warm_kernels());
    HIP_TRY(warm_grid_binned());
    HIP_TRY(warm_grid_arc());
    HIP_TRY(warm_grid_centre());
    HIP_TRY(warm_grid_scatter());
    HIP_TRY(warm_fft512());
    HIP_TRY(warm_degrid_tile());
    HIP_TRY(warm_degrid_stream());
    HIP_TRY(warm_cgnr());
    HIP_TRY(warm_traj());
    HIP_TRY(hipDeviceSynchronize());
    tron_plan *p = new tron_plan();
    p->cfg = *cfg;
    {   // The plan's stream and the process's first host-to-device copies, still on the runtime's account: the FIRST hipStreamCreate of a process
        // takes 8 ms and its first copy of a megabyte from pageable memory 7.4-7.7 ms (the copy path's staging buffers), 0.2 + 0.07 ms ever after
        // (profiles/round6_plan_time_breakdown.log; until round 6 they were booked as "tables", 15 of the 18 ms a process's first plan showed there).
        const unsigned int zero0 = 0;
        hipError_t e0 = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
        if (e0 == hipSuccess) e0 = hipMalloc(reinterpret_cast<void **>(&p->d_errflag), sizeof(zero0));
        if (e0 == hipSuccess) e0 = hipMemcpy(p->d_errflag, &zero0, sizeof(zero0), hipMemcpyHostToDevice);
        static std::once_flag first_copy;                     // (the staging path of copies from pageable memory comes up at the first LARGE one: 7.4 ms, once per process)
        std::call_once(first_copy, [&]() {
            void *scratch = nullptr;
            std::vector<unsigned char> host((size_t)1 << 20, 0);
            if (e0 == hipSuccess && hipMalloc(&scratch, host.size()) == hipSuccess) {
                (void)hipMemcpy(scratch, host.data(), host.size(), hipMemcpyHostToDevice);
                (void)hipFree(scratch);
            }
            (void)hipGetLastError();
        });
        if (e0 != hipSuccess) {
            tron_plan_destroy(p);
            return fail(TRON_ERR_HIP, "cannot create the plan's stream: %s", hipGetErrorString(e0));
        }
    }
    const double t_runtime = since();                // HIP runtime + code objects + first stream and copy (the first plan of a process pays for all of them)

    p->cfg = *cfg;
    p->d = d;
    p->share_z0 = share_z0;
    p->share_nz = cfg->adjoint ? std::max(share_nz, 1) : 1;
    p->nchan = d.nc * d.nt;
    p->beta = kb_beta(cfg->kernwidth);
    p->kb_mode = cfg->kb_mode == TRON_KB_FAST ? TRON_KB_FAST : TRON_KB_EXACT;
    {   // kb_terms(ceil(W)) coefficients in the last slots, zeros in front (tron_internal.h)
        const int nt = kb_terms((int)ceilf(cfg->kernwidth));
        for (int t = 0; t < kKbPolyTerms; ++t) p->kb_poly[t] = 0.f;
        p->kb_poly_err = kb_poly_fit(cfg->kernwidth, p->kb_poly + (kKbPolyTerms - nt), nt);
    }
    if (p->kb_mode == TRON_KB_FAST && !(p->kb_poly_err < 1e-7)) p->kb_mode = TRON_KB_EXACT;   // polynomial too short for this beta: stay exact
    dcf_constants(d.nro, d.npe1work, &p->dcf_a, &p->dcf_b);
    p->scale = grid_scale(d.nxos, d.npe1work);

    const size_t n2 = (size_t)d.nxos * d.nyos;
    const size_t per_unit = (size_t)p->nchan * n2 * sizeof(float2);
    int units = cfg->adjoint ? p->share_nz : 1;
    // The heaviest tile (the k-space centre, crossed by every spoke) is one wave's serial work, so a
    // launch needs enough slices in flight to cover that critical path: batch up to 1 GiB of grid.
    // ... or 64 slices when the coils are many (still at most 6 GiB of grid: 288 GB of HBM make that cheap)
    // (4 GiB: 256 slices x 8 coils in ONE chain of launches -- round 6, late: +1.7 % short, +2 % sustained over two chains of 128, same box,
    //  alternating, twice; rounds 3-6 batched 2 GiB, which had been +5 % over 1 GiB)
    size_t auto_chunk = std::max<size_t>(1, ((size_t)4 << 30) / per_unit);
    if (auto_chunk < 64) auto_chunk = std::max<size_t>(auto_chunk, std::min<size_t>(64, ((size_t)6 << 30) / per_unit));
    int chunk = cfg->chunk_slices > 0 ? cfg->chunk_slices : (int)std::max<size_t>(1, auto_chunk);
    p->chunk = std::max(1, std::min(chunk, std::max(units, 1)));
    p->chunk_cap = std::max(p->chunk, std::min(std::max(units, 1), p->chunk + p->chunk / 2));
    if (!cfg->adjoint) p->chunk = p->chunk_cap = std::max(1, chunk);

    int rc = TRON_OK;
    std::vector<std::pair<const char *, double>> laps;      // -v: where the table time goes
    auto lap = [&](const char *what) { laps.emplace_back(what, since()); };
    lap("start");
    // The deapodisation table (65 k libm sinhf / sinf at the metric shape, 260 k for a forward plan: the longest host table by far) is
    // filled by a helper thread while this one builds and uploads everything else; joined where it is uploaded.
    std::vector<float> dea(cfg->adjoint ? (size_t)d.nx * d.nx : n2);
    std::thread dea_thread([&]() {
        if (cfg->adjoint) build_deapod_table(d.nx, cfg->kernwidth, cfg->gridos, dea.data());                          // src/tron.cu:635
        else if (d.nxos == d.nyos) build_deapod_table(d.nxos, cfg->kernwidth, 1.f, dea.data());                     // src/tron.cu:643
        else build_deapod_table_rect(d.nyos, d.nxos, cfg->kernwidth, 1.f, dea.data());
    });
    auto bail = [&](int code) { if (dea_thread.joinable()) dea_thread.join(); tron_plan_destroy(p); return code; };

    p->ntrig = trig_table_size(*cfg, d);
    {   // 32x32 tiles, centre first: binned gridding and tiled degridding
        std::vector<int> order;
        build_tile_order(d.nxos, kBinnedTile, order);
        if ((rc = upload(&p->d_tile_order32, order.data(), order.size() * sizeof(int)))) return bail(rc);
        if (!cfg->adjoint || cfg->niter > 0) {      // CGNR (an adjoint plan) runs the forward operator too: its centre tiles need the short runs as well
            const int target = 24000;                                         // samples per run of images of one tile (degrid_stream_kernel)
            build_degrid_groups(d.nxos, kBinnedTile, d.npe1work, d.nro, target, p->dg_group_end);
        }
    }
    lap("tile order");
    if (cfg->adjoint) {
        std::vector<uint32_t> band(n2);
        build_band_table(d.nxos, cfg->kernwidth, band.data());
        lap("band table");
        if ((rc = upload(&p->d_band, band.data(), band.size() * sizeof(uint32_t)))) return bail(rc);
        lap("band upload");
        p->tiles_per_row = (d.nxos + kTile - 1) / kTile;
        p->ntiles = p->tiles_per_row * p->tiles_per_row;
        p->binned = p->kb_mode == TRON_KB_FAST && cfg->kernwidth <= 3.f;
        if (const char *gk = tuning_env("TRON_GRID_KERNEL")) p->binned = p->binned && strcmp(gk, "gather") != 0;
        if (!p->binned) {                                                 // the order-preserving gather kernel's 16 x 16 tiles, centre first
            std::vector<int> order;
            build_tile_order(d.nxos, kTile, order);
            if ((rc = upload(&p->d_tile_order, order.data(), order.size() * sizeof(int)))) return bail(rc);
        }
        if (p->binned) {
            // centre relief: the samples next to the k-space centre get workgroups of their own (tron_grid_binned.hip)
            std::vector<int> rorder, rslots;
            int r0 = 0;
            const char *re = tuning_env("TRON_CENTRE_RELIEF");
            if (!(re && atoi(re) == 0) && build_centre_relief_order(d.nxos, kBinnedTile, d.npe1work, cfg->kernwidth, 8, r0, rorder, rslots)) {
                p->relief_entries = (int)rorder.size();
                p->relief_parts = (rslots[0] >> 20) & 15;
                p->relief_r0 = r0;
                if ((rc = upload(&p->d_tile_order32_relief, rorder.data(), rorder.size() * sizeof(int)))) return bail(rc);
                if ((rc = upload(&p->d_relief_slots, rslots.data(), rslots.size() * sizeof(int)))) return bail(rc);
                // small launches: 15 parts, i.e. about half the serial chain per workgroup (a 32-slice launch lasts ~0.35 ms,
                // an inner-tile workgroup of 1/8 of the spokes 0.2 ms)
                std::vector<int> sorder, sslots;
                int r1 = 0;
                if (build_centre_relief_order(d.nxos, kBinnedTile, d.npe1work, cfg->kernwidth, 15, r1, sorder, sslots, 700) && r1 == r0 &&
                    ((sslots[0] >> 20) & 15) > p->relief_parts) {
                    p->relief_entries_small = (int)sorder.size();
                    p->relief_parts_small = (sslots[0] >> 20) & 15;
                    if ((rc = upload(&p->d_tile_order32_relief_small, sorder.data(), sorder.size() * sizeof(int)))) return bail(rc);
                    if ((rc = upload(&p->d_relief_slots_small, sslots.data(), sslots.size() * sizeof(int)))) return bail(rc);
                }
            } else {
                // no centre relief (small grids, TRON_CENTRE_RELIEF=0): small launches split the k-space-centre tiles over spoke ranges instead
                std::vector<int> sorder, slots;
                const int target = 2500;                                  // records per workgroup and image
                p->max_parts = 8;
                build_split_tile_order(d.nxos, kBinnedTile, d.npe1work, cfg->kernwidth, target, p->max_parts, sorder, slots);
                p->split_entries = (int)sorder.size();
                p->nsplit_slots = (int)slots.size();
                p->split_below = 64;                                      // launches of fewer slices use the split list
                if (const char *e = tuning_env("TRON_SPLIT_BELOW")) p->split_below = atoi(e);
                if (p->nsplit_slots > 0) {
                    if ((rc = upload(&p->d_tile_order32_split, sorder.data(), sorder.size() * sizeof(int)))) return bail(rc);
                    if ((rc = upload(&p->d_split_slots, slots.data(), slots.size() * sizeof(int)))) return bail(rc);
                }
            }
            // arc kernel: everything but the inner tile, when the trajectory and the sample layout allow it
            p->arc = p->relief_entries > 0 && grid_arc_supported(p->nchan, d.nxos, d.nro, d.npe1work, cfg->kernwidth, cfg->input_half);
            // one or two channels: lane = sample, fixed-point sums in LDS (tron_grid_scatter.hip) on the arc kernel's tables
            p->scatter = p->relief_entries > 0 && grid_scatter_supported(p->nchan, d.nxos, d.nro, d.npe1work, cfg->kernwidth, cfg->input_half)
                         && scatter_band_is_analytic(d.nxos, cfg->kernwidth, band.data());
            // Two channels of fp32 k-space stay with the arc kernel: there the scatter's two 64-bit LDS atomics per point are what bounds it
            // (round 5: 149 k slices/s against 157 k); complex-half k-space with two channels has no arc kernel (16-byte copies = four
            // channels) and takes it.  TRON_GRID_KERNEL=scatter (tuning) forces it wherever it is supported.
            const char *gk = tuning_env("TRON_GRID_KERNEL");
            if (p->nchan == 2 && !cfg->input_half && !(gk && strcmp(gk, "scatter") == 0)) p->scatter = false;
            if (gk) {
                p->scatter = p->scatter && strcmp(gk, "binned") != 0 && strcmp(gk, "arc") != 0;
                p->arc = p->arc && strcmp(gk, "binned") != 0;
            }
            lap("relief orders");
            p->relief_r0_binned = p->relief_r0;
            if (p->scatter) {
                p->arc = true;
                // The arc formulation starves next to the k-space centre (a block there meets every spoke), which is why the samples |r| < 14 have
                // a kernel of their own; a sample-driven kernel does not, so its plans leave that kernel only the samples |r| < 5 (a quadrant
                // tile must still hold one side of a spoke only: more than W sqrt(2) = 2.83): 9 of a spoke's samples instead of 27.
                // (A centre tile's run holds every spoke whose line is inside tile + W at radius r0: the quadrant's 90 degrees + 2 asin(W sqrt(2) / r0) --
                // 0.76 of a window's spokes at r0 = 5, 0.59 at 14 -- and a run has 512 entries: windows (passes) of more than 640 spokes keep 14.)
                const int npass_ = (d.npe1work + kArcPassNpe - 1) / kArcPassNpe, sub_ = (d.npe1work + npass_ - 1) / npass_;
                const int r0 = sub_ <= 640 ? 5 : p->relief_r0;
                p->relief_r0 = std::min(p->relief_r0, r0);
            }
            lap("kernel choice");
            if (p->arc && (rc = arc_setup(p, band))) return bail(rc);
            lap("centre tables");
            if (p->arc) {
                std::vector<float> lut(6 * (size_t)kArcLutEntries);
                p->lut_entries = build_kb_pair_lut(cfg->kernwidth, kArcLutEntries, lut.data(), &p->lut_scale, &p->lut_bias, &p->lut_err);
                if (p->lut_entries <= 0) return bail(fail(TRON_ERR_UNSUPPORTED, "no Kaiser-Bessel pair table for width %g", (double)cfg->kernwidth));
                if ((rc = upload(&p->d_kb_lut, lut.data(), lut.size() * sizeof(float)))) return bail(rc);
                // what ONE spoke can add to one grid point, in units of the largest weighted sample: the window products along a line at unit
                // spacing sum to at most 1.65 K(0)^2 for every direction and offset at W = 2 (axis 1.61, diagonal 1.65, 2 : 1 1.64; less for
                // narrower windows) -- 1.75 with margin.  (4 K(0)^2 until round 5: one to two bits of the fixed-point sums given away)
                p->scat_wsum = (float)(1.75 * kb_peak(cfg->kernwidth) * kb_peak(cfg->kernwidth));
                p->scat_wmax = (float)(1.00001 * kb_peak(cfg->kernwidth) * kb_peak(cfg->kernwidth));
                p->arc_zper = 0;                                    // 0: by launch size (tron_pipeline.cpp)
            }
        }
        lap("pair table");
        dea_thread.join();
        lap("deapod join");
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
        dea_thread.join();
        if ((rc = upload(&p->d_deapod, dea.data(), dea.size() * sizeof(float)))) return bail(rc);
    }
    unsigned int zero = 0;
    if (!p->d_errflag && (rc = upload(&p->d_errflag, &zero, sizeof(zero)))) return bail(rc);
    p->poison = debug_token("poison");                       // tests (TRON_DEBUG=poison): NaN-fill the work grid so a read of a never-written point shows up
    if (d.nxos == 512 && d.nx == 256 && d.nyos == 512 && d.ny == 256) {
        p->fft512 = true;
        if (const char *ff = tuning_env("TRON_FFT")) p->fft512 = strcmp(ff, "rocfft") != 0;
    }
    if (p->fft512) {
        std::vector<float> tw(2 * 512);
        for (int k = 0; k < 512; ++k) {
            tw[2 * k] = (float)cos(2.0 * M_PI * k / 512.0);
            tw[2 * k + 1] = (float)sin(2.0 * M_PI * k / 512.0);
        }
        if ((rc = upload(&p->d_tw512, tw.data(), tw.size() * sizeof(float)))) return bail(rc);
    }
    // The angle-dependent tables of the plan's own skip_angles: the build tron_plan_retarget repeats for later ones (tron_traj.cpp), here on
    // the plan's stream and waited for.  Run tables that overflow (more spokes through a tile than a run holds, ...) send the plan to the
    // next formulation down -- 64-tiles -> 32-tiles -> arc kernel -> binned kernel -- instead of straight to the binned one (round 5).
    lap("fft tables");
    const double t_arc0 = since();
    for (;;) {
        TrajTables &T = p->traj[0];
        if ((rc = traj_alloc(p, T)) || (rc = traj_build(p, T, cfg->skip_angles, p->stream)) || (rc = traj_finish(p, T))) return bail(rc);
        if (T.ok || !p->arc) break;
        // never silent: a slower kernel than the shape would normally get (tron_plan_grid_kernel_name tells which one a plan runs)
        const bool can_arc = grid_arc_supported(p->nchan, d.nxos, d.nro, d.npe1work, cfg->kernwidth, cfg->input_half);
        const unsigned flag = *T.h_flag;
        if (p->scatter && p->scat_tile == 64) {
            p->scat_tile_max = 32;
            fprintf(stderr, "tronhip: the scatter kernel's run tables overflowed on 64 x 64 tiles (flag %u): trying 32 x 32 tiles\n", flag);
        } else if (p->scatter && can_arc) {
            p->scatter = false;
            p->relief_r0 = p->relief_r0_binned;
            fprintf(stderr, "tronhip: the scatter kernel's run tables overflowed (flag %u): trying the arc kernel\n", flag);
        } else {
            p->arc = p->scatter = false;                      // the binned kernel takes every tile; later retargets rebuild the (cos, sin) table only
            p->relief_r0 = p->relief_r0_binned;
            fprintf(stderr, "tronhip: run tables overflowed (flag %u): falling back to the binned gridding kernel\n", flag);
            break;
        }
        traj_free(T);
        std::vector<uint32_t> band(n2);
        build_band_table(d.nxos, cfg->kernwidth, band.data());
        if ((rc = arc_setup(p, band))) return bail(rc);
    }
    const double t_arc1 = since();
    const double t_tables = since();
    // adjoint: the whole batch now (an out-of-memory plan fails here, not mid-run); forward: on first use, sized by the call
    if (cfg->adjoint && (rc = ensure_work(p, p->chunk_cap))) return bail(rc);
    const double t_work = since();
    if (!p->fft512) {   // rocFFT plans for the batch sizes this plan will certainly see: no plan creation (and device
        FftPlan *f = nullptr;       // synchronisation) in the middle of the first pipeline
        if ((rc = get_fft(p, (cfg->adjoint ? std::min(p->chunk, std::max(d.nz, 1)) : 1) * p->nchan, cfg->adjoint ? 1 : 0, &f))) return bail(rc);
    }
    p->create_s[1] = t_runtime; p->create_s[2] = t_tables - t_runtime; p->create_s[3] = t_arc1 - t_arc0; p->create_s[4] = t_work - t_tables;
    p->create_s[0] = since();
    if (cfg->verbose) {
        printf("tronhip: device %d, %s, nchan %d, grid %d^2 -> image %d^2, %d spokes/image, chunk %d, KB %s\n",
               cfg->device, cfg->adjoint ? "adjoint" : "forward", p->nchan, d.nxos, d.nx, d.npe1work, p->chunk,
               p->kb_mode == TRON_KB_FAST ? "fast" : "exact");
        printf("tronhip: fast Kaiser-Bessel polynomial (%d terms) max error relative to the peak %.2e\n", kb_terms((int)ceilf(cfg->kernwidth)), p->kb_poly_err);
        printf("tronhip: gridding kernel: %s\n", tron_plan_grid_kernel_name(p));
        printf("tronhip: plan %.3f s = HIP runtime + code objects %.3f, tables %.3f (of which the arc kernel's run tables %.3f), work buffers %.3f\n",
               since(), t_runtime, t_tables - t_runtime, t_arc1 - t_arc0, t_work - t_tables);
    if (cfg->verbose) printf("tronhip: of the start-up, the HIP runtime itself %.3f s, loading the code objects %.3f s\n", t_hipinit, t_runtime - t_hipinit);
        printf("tronhip: tables (ms):");
        for (size_t i = 1; i < laps.size(); ++i) printf(" %s %.2f", laps[i].first, 1e3 * (laps[i].second - laps[i - 1].second));
        printf(" | trajectory tables %.2f\n", 1e3 * (t_arc1 - t_arc0));
    }
    p->sync_each = debug_token("sync");                      // TRON_DEBUG=sync: synchronise after every stage and name the failing one
    if (const char *dk = tuning_env("TRON_DEGRID_KERNEL")) {     // simple: the thread-per-sample audit kernel; tile: never the streaming kernel
        p->degrid_simple = strcmp(dk, "simple") == 0;
        p->degrid_tile_only = strcmp(dk, "tile") == 0;
    }
    if (const char *sp = tuning_env("TRON_SLICES_PER_PASS")) p->slices_per_pass = atoi(sp) != 0;
    p->pin_host = cfg->pin_host != 0;
    *out = p;
    return TRON_OK;
}

extern "C" int tron_plan_destroy(tron_plan *p)
{
    if (!p) return TRON_OK;
    hipSetDevice(p->cfg.device);
    if (p->stream) hipStreamSynchronize(p->stream);
    if (p->stream_build) hipStreamSynchronize(p->stream_build);
    for (int s = 0; s < STAGE_COUNT; ++s)
        for (auto &pr : p->ev[s]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    for (auto &kv : p->fft) {
        if (kv.second.info) rocfft_execution_info_destroy(kv.second.info);
        if (kv.second.plan) rocfft_plan_destroy(kv.second.plan);
        if (kv.second.work) hipFree(kv.second.work);
    }
    traj_free(p->traj[0]);
    traj_free(p->traj[1]);
    hipFree(p->d_band);
    hipFree(p->d_tile_order);
    hipFree(p->d_tile_order32);
    hipFree(p->d_coil_tmp);
    hipFree(p->d_deapod_fwd);
    hipFree(p->d_trig_fwd);
    hipFree(p->d_cg_r); hipFree(p->d_cg_v); hipFree(p->d_cg_zt); hipFree(p->d_cg_pt); hipFree(p->d_cg_x);
    hipFree(p->d_cg_partial); hipFree(p->d_cg_num); hipFree(p->d_cg_coef);
    hipFree(p->d_kb_lut);
    hipFree(p->d_cen_gwin);
    hipFree(p->d_cen_grec);
    hipFree(p->d_cen_grec_parts);
    hipFree(p->d_cen_ticket);
    hipFree(p->d_cen_parts);
    hipFree(p->d_tile_order32_split);
    hipFree(p->d_split_slots);
    hipFree(p->d_partial);
    hipFree(p->d_tile_order32_relief);
    hipFree(p->d_tile_order64);
    hipFree(p->d_relief_slots);
    hipFree(p->d_tile_order32_relief_small);
    hipFree(p->d_relief_slots_small);
    hipFree(p->d_relief_partial);
    hipFree(p->d_deapod);
    hipFree(p->d_errflag);
    hipFree(p->d_grid);
    hipFree(p->d_stage_in);
    hipFree(p->d_stage_out);
    hipFree(p->d_trig_tmp);
    hipFree(p->d_tw512);
    hipFree(p->d_fft_tmp);
    for (hipEvent_t e : p->ev_pipe) hipEventDestroy(e);
    if (p->stream_up) hipStreamDestroy(p->stream_up);
    if (p->stream_down) hipStreamDestroy(p->stream_down);
    if (p->stream_build) hipStreamDestroy(p->stream_build);
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
    return TRON_OK;
}

extern "C" const char *tron_plan_grid_kernel_name(const tron_plan *p)
{
    if (!p || !p->cfg.adjoint) return "";
    // linear angles with few channels: several slices share one pass of the binned kernel (slice groups, tron_pipeline.cpp), whatever tables the plan holds
    if (p->binned && !p->cfg.golden_angle && p->nchan <= 4 && p->slices_per_pass && p->d.nz > 1) return "grid_binned_kernel (linear-angle slice groups)";
    if (arc_ready(p) && p->scatter) return "grid_scatter_kernel (+ grid_centre_kernel for |r| < inner_r0)";
    if (arc_ready(p)) return "grid_arc_kernel (+ grid_centre_kernel for |r| < inner_r0)";
    if (p->binned) return p->relief_entries > 0 ? "grid_binned_kernel (+ grid_reduce_parts_kernel)" : "grid_binned_kernel";
    return "grid_tile_kernel";
}

extern "C" const char *tron_plan_degrid_kernel_name(const tron_plan *p)
{
    return p ? p->last_degrid_kernel : "";
}

extern "C" int tron_plan_shader_clock(tron_plan *p, double *mhz)
{
    if (!p || !mhz) return fail(TRON_ERR_INVALID, "tron_plan_shader_clock: null argument");
    HIP_TRY(hipSetDevice(p->cfg.device));
    unsigned long long *d = nullptr, h[4] = {0, 0, 0, 0};
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), sizeof(h)));
    hipError_t e = launch_clock_probe(d, p->stream);             // behind whatever the plan has queued: the clock that work ran at
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (e == hipSuccess) e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return fail(TRON_ERR_HIP, "clock probe failed: %s", hipGetErrorString(e));
    *mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    return TRON_OK;
}

extern "C" int tron_plan_sync(tron_plan *p)
{
    if (!p) return fail(TRON_ERR_INVALID, "tron_plan_sync: null plan");
    HIP_TRY(hipStreamSynchronize(p->stream));
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
    return adjoint_run(p, d_out, in, zfirst, zcount, combine, 0);   // asynchronous: tron_plan_sync waits for it
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
    if (int trc = traj_turn(p)) return trc;
    const tron_dims &d = p->d;
    DegridParams g;
    memset(&g, 0, sizeof(g));
    g.udata = static_cast<const float2 *>(d_udata);
    g.nudata = static_cast<float2 *>(d_nudata);
    g.trig = traj_cur(p).d_trig;
    g.tile_order = p->d_tile_order32;
    g.in_z = 0;
    g.in_c = 1;                           // udata[nrep*(i*n+j) + c], src/tron.cu:571-573
    g.in_p = p->nchan;
    g.in_shift = 0;
    g.n = d.nxos;
    g.nrows = d.nyos != d.nxos ? d.nyos : 0;
    if (g.nrows) g.tile_order = nullptr;                       // raster order for non-square grids
    g.nrep = p->nchan;
    g.nro = d.nro;
    g.npe = d.npe1work;
    g.nimg = 1;
    g.W = p->cfg.kernwidth;
    g.beta = p->beta;
    memcpy(g.kb_poly, p->kb_poly, sizeof(g.kb_poly));
    StageTimer t(p, STAGE_DEGRID);
    p->last_degrid_kernel = p->degrid_simple ? "degrid_kernel" : "degrid_tile_kernel";
    if (!p->degrid_simple)
        HIP_TRY(launch_degrid_tile(g, p->kb_mode, p->stream));
    else
        HIP_TRY(launch_degrid(g, p->kb_mode, p->stream));
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

// PCI bus id of a HIP device ("0000:c1:00.0"): what sysfs keys its NUMA node by (tron_host_numa_cpulist); one-rank-per-GPU launchers
// bind each rank to its GPU's socket with it (tron_amd/launch.py: bind_near_gpu).
extern "C" int tron_device_pci_bus_id(int device, char *buf, int len)
{
    if (!buf || len < 16) return fail(TRON_ERR_INVALID, "tron_device_pci_bus_id: buffer of at least 16 bytes needed");
    HIP_TRY(hipDeviceGetPCIBusId(buf, len, device));
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
