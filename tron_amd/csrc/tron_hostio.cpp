// Host-buffer entry points of libtronhip (= recon_radial2d, src/tron.cu:726-786): the chunked upload / compute / download
// pipeline, block-relative buffers for one-rank-per-GPU callers, and the in-process multi-GPU form (the reference's
// compiled-out MULTI_GPU branch, src/tron.cu:582-597,735-736).
#include "tron_plan_impl.h"

#include <pthread.h>
#include <sched.h>
#include <unistd.h>

using namespace tron;

// Registration of the caller's two buffers for a call (hipHostRegister maps whole pages of user memory for the GPU's copy engines).
// WHICH memory may be registered is the point of this struct.  Rounds 2-5 registered any buffer of a call that moved >= 8 MiB; round 6
// found (tests/test_gpu_retarget.py, then tools/probe/hostreg_heap.py) that a copy FROM a registered buffer that lives on the brk heap
// ends about every third process with "Memory access fault by GPU ... on address <page inside the registered range>" once the heap has been
// worked (earlier buffers registered, unregistered and freed there; glibc's dynamic mmap threshold grows to 32 MiB after the first large
// free, and 10 MB arrays then come from the heap): 13 of 36 processes with the heap buffer, 0 of 36 with the same bytes in a mapping of
// their own (MALLOC_MMAP_THRESHOLD_ fixed), 0 of 24 with pageable copies.  The driver's user-pointer mapping follows the heap's one
// growing and shrinking VMA badly; a mapping of its own it follows well.  So a buffer is registered only when
//   * it is at least kPinMinBytes = 32 MiB (glibc's largest dynamic mmap threshold: a malloc'd block of that size is always a mapping
//     of its own; below it the page walk of a registration costs what the pinned copy saves anyway), and
//   * it lies above the program break (never static data, never the brk heap, whatever the allocator),
// and everything else is copied as pageable memory (staged by the runtime at 0.89 of the pinned rate).
// Page-rounded ranges, and ONE registration of their union when the two buffers touch or overlap.
struct HostPins {
    static constexpr size_t kPinMinBytes = (size_t)32 << 20;
    void *base[2] = {nullptr, nullptr};
    int n = 0;
    static bool eligible(const void *ptr, size_t bytes)
    {
        return bytes >= kPinMinBytes && reinterpret_cast<uintptr_t>(ptr) >= reinterpret_cast<uintptr_t>(sbrk(0));
    }
    void pin(const void *a, size_t abytes, const void *b, size_t bbytes, unsigned flags)
    {
        // TRON_DEBUG=pin_any: the rule of rounds 2-5 (both buffers of a call that moves >= 8 MiB): tools/probe/hostreg_heap.py shows the fault with it
        static const bool any_dbg = debug_token("pin_any");
        const bool any = any_dbg && abytes + bbytes >= ((size_t)8 << 20);
        const uintptr_t pg = 4096;
        uintptr_t a0 = 0, a1 = 0, b0 = 0, b1 = 0;
        if (any || eligible(a, abytes)) { a0 = reinterpret_cast<uintptr_t>(a) & ~(pg - 1); a1 = (reinterpret_cast<uintptr_t>(a) + abytes + pg - 1) & ~(pg - 1); }
        if (any || eligible(b, bbytes)) { b0 = reinterpret_cast<uintptr_t>(b) & ~(pg - 1); b1 = (reinterpret_cast<uintptr_t>(b) + bbytes + pg - 1) & ~(pg - 1); }
        if (a1 > a0 && b1 > b0 && a1 >= b0 && b1 >= a0) {      // neighbours or overlapping: one range
            a0 = std::min(a0, b0); a1 = std::max(a1, b1);
            b0 = b1 = 0;
        }
        for (auto r : {std::make_pair(a0, a1), std::make_pair(b0, b1)})
            if (r.second > r.first && hipHostRegister(reinterpret_cast<void *>(r.first), r.second - r.first, flags) == hipSuccess)
                base[n++] = reinterpret_cast<void *>(r.first);
        (void)hipGetLastError();                               // (a range that cannot be registered -- already pinned by the caller, ... -- is copied as pageable memory)
    }
    void unpin()
    {
        for (int i = 0; i < n; ++i) hipHostUnregister(base[i]);
        n = 0;
    }
};

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
    // Pinning the caller's buffers makes the copies truly asynchronous (and the two directions concurrent): the default
    // (cfg.pin_host = 1, as the reference pins its output, src/tron.cu:967; the `tron` binary's streamed blocks turn it off, tron_main.cpp).
    // It costs a page walk of the whole range per call and is safe only for some memory: HostPins (above) decides per buffer; what
    // it leaves is copied as pageable memory, staged by the runtime at 0.89 of the rate and still overlapping the previous chunk's kernels.
    HostPins pins;
    if (p->pin_host) pins.pin(src, in_bytes, dst, out_bytes, hipHostRegisterDefault);
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
    pins.unpin();
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

// NUMA: binds the calling thread to the CPUs of the NUMA node the device hangs off (sysfs: the PCI function's numa_node, the
// node's cpulist).  Best effort: no node information (or a single-node host) leaves the thread where it is.
static void pin_thread_near_device(int device)
{
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return; }
    int cpus[4096];
    const int n = tron_host_numa_cpulist("/sys", bus, cpus, 4096);
    if (n <= 0) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int i = 0; i < n; ++i)
        if (cpus[i] >= 0 && cpus[i] < CPU_SETSIZE) CPU_SET(cpus[i], &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
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
    // the workers' slice blocks share spokes (windows overlap) and pages: pin both buffers ONCE -- hipHostRegisterPortable makes
    // the registration visible to every device's context, whichever device is current here
    HostPins pins;
    const size_t in_bytes = (size_t)dims->in_elems * (cfg->input_half ? 4 : 8);
    if (cfg->pin_host && workers > 1) {
        HIP_TRY(hipSetDevice(devs[0]));
        pins.pin(h_in, in_bytes, h_out, (size_t)dims->out_bytes, hipHostRegisterPortable);
    }
    auto work = [&](int g) {
        tron_config c = *cfg;
        c.device = devs[g];
        if (workers > 1) c.pin_host = 0;
        if (workers > 1) pin_thread_near_device(devs[g]);        // the worker feeds its GPU from the host buffer: run on that GPU's socket
        const int z0 = (int)((long long)g * nz / workers), z1 = (int)((long long)(g + 1) * nz / workers);
        tron_plan *plan = nullptr;
        // a plan for this worker's block only: batches, work buffers and run tables sized for z1 - z0 slices
        int rc = plan_create_share(&plan, &c, dims, c.adjoint ? z0 : 0, c.adjoint ? z1 - z0 : std::max(nz, 1));
        if (rc == TRON_OK) rc = tron_recon_radial2d_range(plan, h_out, h_in, z0, z1 - z0);
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
    pins.unpin();
    for (int g = 0; g < workers; ++g)
        if (rcs[g] != TRON_OK) return fail(rcs[g], "device worker %d (HIP device %d): %s", g, devs[g], msgs[g].c_str());
    return TRON_OK;
}
