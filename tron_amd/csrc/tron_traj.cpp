// The angle-dependent half of a plan (TrajTables, tron_plan_impl.h) and tron_plan_retarget.
//
// The reference fixes its spoke angles per kernel call -- gridradial2d / degridradial2d take skip_angles + the slice's offset and
// evaluate PHI * float(pe + skip) per thread (src/tron.cu:509-511, 555-559, 629-630) -- so a new acquisition costs it nothing beyond
// the call.  This library moves everything that depends on the angles alone to tables (the (cos, sin) list, the angle-sorted
// spoke lists, the centre kernel's block windows, the arc / scatter kernels' run tables), which made a new set of angles a new plan:
// 40 ms for the 256 windows of the metric shape, against 3.5 ms for gridding them (round 5).  Here the tables are one object of
// which a plan holds two; building one is a (cos, sin) table by libm on a few host threads plus three device launches, and
// tron_plan_retarget queues that for the idle set on a stream of its own while the launches already queued run on with the other.
#include "tron_plan_impl.h"

#include <time.h>

namespace tron {

static double now_s()
{
    struct timespec b;
    clock_gettime(CLOCK_MONOTONIC, &b);
    return b.tv_sec + 1e-9 * b.tv_nsec;
}

template <typename T>
static bool dev_alloc(T **ptr, size_t count)
{
    return hipMalloc(reinterpret_cast<void **>(ptr), std::max<size_t>(count, 1) * sizeof(T)) == hipSuccess;
}

void traj_free(TrajTables &T)
{
    if (T.h_trig) hipHostFree(T.h_trig);
    if (T.h_flag) hipHostFree(T.h_flag);
    for (void *q : {(void *)T.d_trig, (void *)T.d_order, (void *)T.d_phi, (void *)T.d_cs, (void *)T.d_order_q, (void *)T.d_phi_q, (void *)T.d_cs_q,
                    (void *)T.d_cen_win, (void *)T.d_arc_hdr, (void *)T.d_arc_ent, (void *)T.d_arc_win, (void *)T.d_arc_off, (void *)T.d_arc_rec,
                    (void *)T.d_arc_rbase, (void *)T.d_alloc, (void *)T.d_flag})
        if (q) hipFree(q);
    if (T.ev_built) hipEventDestroy(T.ev_built);
    if (T.ev_released) hipEventDestroy(T.ev_released);
    T = TrajTables();
}

// Sizes come from the plan (set by tron_plan_create before the first call): windows, spokes per window and pass, tiles, caps.
int traj_alloc(tron_plan *p, TrajTables &T)
{
    if (T.allocated) return TRON_OK;
    const tron_dims &d = p->d;
    bool ok = hipHostMalloc(reinterpret_cast<void **>(&T.h_trig), std::max<size_t>(p->ntrig, 1) * 2 * sizeof(float), hipHostMallocDefault) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void **>(&T.h_flag), sizeof(unsigned int), hipHostMallocDefault) == hipSuccess &&
              dev_alloc(&T.d_trig, p->ntrig) && dev_alloc(&T.d_flag, 1) &&
              hipEventCreateWithFlags(&T.ev_built, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&T.ev_released, hipEventDisableTiming) == hipSuccess;
    if (ok) *T.h_flag = 0;
    if (ok && p->arc) {
        const size_t nwin = p->arc_nwin, npe = (size_t)d.npe1work, npass = (size_t)p->arc_passes, sub = (size_t)p->arc_pass_npe;
        const size_t nt32 = (size_t)(d.nxos / kBinnedTile) * (d.nxos / kBinnedTile), nt_tab = p->arc_ntiles, ntab = nwin * npass;
        ok = dev_alloc(&T.d_order, nwin * npe) && dev_alloc(&T.d_phi, nwin * npe) && dev_alloc(&T.d_cs, nwin * npe) &&
             dev_alloc(&T.d_cen_win, nwin * (size_t)p->cen_nblocks) && dev_alloc(&T.d_alloc, 2 * ntab) &&
             dev_alloc(&T.d_arc_hdr, ntab * nt_tab) && dev_alloc(&T.d_arc_ent, ntab * (size_t)p->arc_cap);
        if (ok && npass > 1) ok = dev_alloc(&T.d_order_q, ntab * sub) && dev_alloc(&T.d_phi_q, ntab * sub) && dev_alloc(&T.d_cs_q, ntab * sub);
        if (ok && p->scatter)       // (scatter kernel: record offsets and member tables, no thread windows)
            ok = dev_alloc(&T.d_arc_off, ntab * (size_t)p->arc_cap) && dev_alloc(&T.d_arc_rec, (ntab * (size_t)p->arc_rec_cap + 16) * 80) &&
                 dev_alloc(&T.d_arc_rbase, ntab * nt_tab);
        else if (ok)
            ok = dev_alloc(&T.d_arc_win, ntab * nt32 * 256);
    }
    if (!ok) {
        (void)hipGetLastError();
        traj_free(T);
        return fail(TRON_ERR_NOMEM, "cannot allocate the plan's trajectory tables");
    }
    T.allocated = true;
    return TRON_OK;
}

// Queues the whole build of set T for `skip_angles` on `st` and records T.ev_built behind it.  Host work: the (cos, sin) table only
// (the reference's own expression and libm, tron_hostmath.cpp), into pinned memory.
int traj_build(tron_plan *p, TrajTables &T, int skip_angles, hipStream_t st)
{
    const tron_dims &d = p->d;
    const double t0 = now_s();
    tron_config c = p->cfg;
    c.skip_angles = skip_angles;
    static const int host_threads = std::max(1, std::min(16, (int)std::thread::hardware_concurrency() / 2));
    build_trig_table_mt(c, d, T.h_trig, p->ntrig, host_threads);
    p->retarget_s[1] = now_s() - t0;
    T.skip_angles = skip_angles;
    T.ok = true;
    HIP_TRY(hipMemcpyAsync(T.d_trig, T.h_trig, p->ntrig * 2 * sizeof(float), hipMemcpyHostToDevice, st));
    if (p->arc) {
        const int golden = p->cfg.golden_angle;
        const size_t nwin = p->arc_nwin, win_first = golden ? (size_t)p->share_z0 : 0;
        const int npe = d.npe1work, npass = p->arc_passes, sub = p->arc_pass_npe;
        const size_t nt32 = (size_t)(d.nxos / kBinnedTile) * (d.nxos / kBinnedTile), nt_tab = p->arc_ntiles;
        // every memset ON THE BUILD'S STREAM: one on the null stream is not ordered against a non-blocking stream
        HIP_TRY(hipMemsetAsync(T.d_flag, 0, sizeof(unsigned int), st));
        HIP_TRY(hipMemsetAsync(T.d_alloc, 0, 2 * nwin * (size_t)npass * sizeof(int), st));
        TrajSortParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.trig = T.d_trig + win_first * (size_t)d.prof_slide;
        sp.win_stride = golden ? d.prof_slide : 0;
        sp.nwin = (int)nwin; sp.npe = npe; sp.npass = npass; sp.sub = sub;
        sp.order = T.d_order; sp.phi = T.d_phi; sp.cs = T.d_cs;
        sp.order_q = T.d_order_q; sp.phi_q = T.d_phi_q; sp.cs_q = T.d_cs_q;
        HIP_TRY(launch_traj_sort(sp, st));
        TrajCentreParams cp;
        cp.phi = T.d_phi; cp.gwin = p->d_cen_gwin; cp.out = T.d_cen_win;
        cp.nwin = (int)nwin; cp.npe = npe; cp.ngroups = p->cen_nblocks;
        HIP_TRY(launch_traj_centre_windows(cp, st));
        for (int q = 0; q < npass; ++q) {
            const int lo = q * sub, hi = std::min(npe, lo + sub), nq = hi - lo;
            const size_t qoff = (size_t)q * nwin * sub;
            ArcPrepParams ap;
            memset(&ap, 0, sizeof(ap));
            ap.order = npass > 1 ? T.d_order_q + qoff : T.d_order;
            ap.phi = npass > 1 ? T.d_phi_q + qoff : T.d_phi;
            ap.cs = npass > 1 ? T.d_cs_q + qoff : T.d_cs;
            ap.hdr = T.d_arc_hdr + (size_t)q * nwin * nt_tab;
            ap.ent = T.d_arc_ent + (size_t)q * nwin * p->arc_cap;
            ap.win = T.d_arc_win ? T.d_arc_win + (size_t)q * nwin * nt32 * 256 : nullptr;
            ap.band = p->d_band;
            ap.alloc = T.d_alloc + (size_t)q * 2 * nwin;
            ap.errflag = T.d_flag;
            ap.off = T.d_arc_off ? T.d_arc_off + (size_t)q * nwin * p->arc_cap : nullptr;
            ap.tile = p->scatter ? p->scat_tile : 0;
            ap.rec = T.d_arc_rec ? T.d_arc_rec + (size_t)q * nwin * p->arc_rec_cap * 80 : nullptr;
            ap.rbase = T.d_arc_rbase ? T.d_arc_rbase + (size_t)q * nwin * nt_tab : nullptr;
            ap.ralloc = ap.alloc + nwin;
            ap.rec_cap = p->arc_rec_cap;
            ap.nxos = d.nxos; ap.nro = d.nro; ap.npe = nq; ap.ntiles = (int)nt_tab; ap.inner_r0 = p->relief_r0; ap.nrec = p->arc_nrec;
            ap.cap = p->arc_cap; ap.W = p->cfg.kernwidth; ap.flat = p->scatter ? 1 : 0;
            HIP_TRY(launch_arc_prep(ap, (int)nwin, st));
        }
        HIP_TRY(hipMemcpyAsync(T.h_flag, T.d_flag, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipEventRecord(T.ev_built, st));
    return TRON_OK;
}

int traj_finish(tron_plan *p, TrajTables &T)
{
    HIP_TRY(hipEventSynchronize(T.ev_built));
    const unsigned int flag = p->arc ? *T.h_flag : 0u;
    T.ok = flag == 0;
    return TRON_OK;
}

int traj_turn(tron_plan *p)
{
    if (!p->retarget_pending) return TRON_OK;
    TrajTables &next = p->traj[1 - p->cur];
    p->retarget_pending = false;
    int rc = traj_finish(p, next);
    if (rc) return rc;
    if (!next.ok)  // never silent: a 2x slower kernel the caller did not ask for (a plan CREATED with these angles would have gone down the
                   // same road, one formulation at a time; a retargeted plan keeps its table layout and goes straight to the binned kernel)
        fprintf(stderr, "tronhip: run tables overflowed for skip_angles = %d (flag %u): the binned gridding kernel takes these angles\n",
                next.skip_angles, *next.h_flag);
    // launches queued from now on read `next`; the set they leave is free for the next build once everything queued so far has run
    TrajTables &old = p->traj[p->cur];
    HIP_TRY(hipEventRecord(old.ev_released, p->stream));
    old.released = true;
    p->cur = 1 - p->cur;
    p->cfg.skip_angles = next.skip_angles;
    return TRON_OK;
}

}  // namespace tron

using namespace tron;

// The plan's spoke angles start at a new index: every later call on this plan grids / degrids with skip_angles = `skip_angles`
// (src/tron.cu:509, 555: PHI * float(pe + skip), the `-s` flag) -- a continuing golden-angle acquisition, batch after batch, on ONE
// plan.  Asynchronous: the tables of the new angles are built in the plan's second table set on a stream of its own, beside whatever
// the plan has queued; the next call on the plan waits for that build (not for the queued work) and reads the new set.  Results are
// those of a plan created with this skip_angles, bit for bit.  Linear angles (golden_angle = 0) do not depend on skip_angles: no-op.
extern "C" int tron_plan_retarget(tron_plan *p, int skip_angles)
{
    if (!p) return fail(TRON_ERR_INVALID, "tron_plan_retarget: null plan");
    const double t0 = now_s();
    HIP_TRY(hipSetDevice(p->cfg.device));
    int rc = traj_turn(p);                                    // a retarget nobody has used yet: finish it first (the sets alternate)
    if (rc) return rc;
    if (!p->cfg.golden_angle) { p->cfg.skip_angles = skip_angles; return TRON_OK; }
    if (!p->stream_build) HIP_TRY(hipStreamCreateWithFlags(&p->stream_build, hipStreamNonBlocking));
    TrajTables &T = p->traj[1 - p->cur];
    if ((rc = traj_alloc(p, T))) return rc;
    if (T.released) {                                         // its last readers are queued on the plan's stream: the build waits for them
        HIP_TRY(hipStreamWaitEvent(p->stream_build, T.ev_released, 0));
        T.released = false;
    }
    if ((rc = traj_build(p, T, skip_angles, p->stream_build))) return rc;
    p->retarget_pending = true;
    p->retarget_s[0] = now_s() - t0;
    return TRON_OK;
}

extern "C" int tron_plan_retarget_times(const tron_plan *p, double seconds[2])
{
    if (!p || !seconds) return fail(TRON_ERR_INVALID, "tron_plan_retarget_times: null argument");
    seconds[0] = p->retarget_s[0];
    seconds[1] = p->retarget_s[1];
    return TRON_OK;
}
