/*
 * tron_hip.h -- C ABI of libtronhip, the MI355X (gfx950) implementation of TRON's 2-D
 * radial gridding / degridding path.
 *
 * Every entry point replaces one piece of the reference's `extern "C"` surface
 * (davidssmith/TRON, src/tron.cu:460-788, declared in src/tron.h:55-74) or of its main()
 * (src/tron.cu:813-995); the reference line each one stands for is cited on it.  The
 * reference keeps its configuration in file-static globals (src/tron.cu:54-87), so there
 * it is neither re-entrant nor callable with more than one configuration; here the same
 * values travel in `tron_config` / `tron_dims` and all device state lives in an opaque
 * `tron_plan`.
 *
 * Plain C types only: no HIP, no C++, no torch types.  Device buffers are `void*`
 * device addresses (anything hipMalloc / torch / cupy hands out).  All functions return
 * TRON_OK (0) or a TRON_ERR_* code and never call exit(); `tron_last_error()` returns
 * the message of the calling thread's last failure.
 *
 * Data layouts at the boundary are the reference's, complex = interleaved
 * (re, im) float pairs (CUDA float2):
 *   adjoint  in : k-space  h_in [c + nc*(t + nt*(ro + nro*pe))]          (src/tron.cu:519)
 *   adjoint  out: images   h_out[nt*nx*ny*z + row*nx + col]              (src/tron.cu:494,740,768)
 *   forward  in : images   h_in [c + nchan*(row*nx + col)]               (src/tron.cu:452-454)
 *   forward  out: k-space  h_out[nchan*nro*npe*z + c + nchan*(ro + nro*pe)] (src/tron.cu:550,776)
 * row <-> the sine axis, col <-> the cosine axis of the spoke angle.
 */
#ifndef TRON_HIP_H
#define TRON_HIP_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: this header IS its export list */
#endif
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tron_float2 { float x, y; } tron_float2;

enum {
    TRON_OK = 0,
    TRON_ERR_INVALID = 1,      /* bad argument / dimension */
    TRON_ERR_UNSUPPORTED = 2,  /* valid in the reference's CLI but not implemented (e.g. -i) */
    TRON_ERR_HIP = 3,          /* HIP runtime failure (no device, out of memory, launch error) */
    TRON_ERR_FFT = 4,          /* rocFFT failure */
    TRON_ERR_NOMEM = 5
};

/* Kaiser-Bessel evaluation modes (tron_config.kb_mode) */
enum {
    TRON_KB_EXACT = 0,  /* the reference's formula op for op (src/tron.cu:304-349): IEEE sqrt/div, double Horner, and the
                           reference's summation order: bit-identical interpolation, for audits; about 2.5x slower */
    TRON_KB_FAST  = 1   /* default: the window from a table of quadratic pieces with the reference's exact support (gridding:
                           arc + centre kernels; build_kb_pair_lut) or from an fp32 polynomial in 1-(x/W)^2 (binned gridding,
                           degridding), both fitted at plan creation (error < 5e-7 of the peak), and each kernel's own summation
                           order; reconstructions agree with TRON_KB_EXACT to ~1e-7 relative L2 (bar: 1e-5), same bits every run */
};

/* The run-time configuration: the getopt-settable globals of src/tron.cu:58-87,
   with the defaults of tron_config_default(). */
typedef struct tron_config {
    int   adjoint;         /* -a  flags.adjoint        (src/tron.cu:828)  0 = forward (degrid) */
    int   golden_angle;    /* -G  flags.golden_angle   (src/tron.cu:840) */
    int   koosh;           /* -3  flags.koosh          (src/tron.cu:825)  dims arithmetic only */
    int   verbose;         /* -v  flags.verbose        (src/tron.cu:867) */
    float gridos;          /* -o  gridos = 2           (src/tron.cu:67)  */
    float kernwidth;       /* -k  kernwidth = 2        (src/tron.cu:68)  */
    float data_undersamp;  /* -u  data_undersamp = 1   (src/tron.cu:69)  */
    int   prof_slide;      /* -d  prof_slide = 0       (src/tron.cu:71)  */
    int   skip_angles;     /* -s  skip_angles = 0      (src/tron.cu:72)  */
    int   niter;           /* -i  niter = 0            (src/tron.cu:74)  >0: CGNR iterations (adjoint only) */
    int   blocks;          /* -B  accepted, ignored (launch shapes are chosen per kernel) */
    int   threads;         /* -T  accepted, ignored */
    int   device;          /* -g  HIP device ordinal   (src/tron.cu:838) */
    /* --- extensions with no counterpart in the reference --- */
    int   kb_mode;         /* TRON_KB_FAST (default) or TRON_KB_EXACT */
    int   input_half;      /* adjoint only: k-space is complex-half (2 x IEEE binary16 per sample) */
    int   chunk_slices;    /* slices per internal batch; 0 = choose from the grid size */
    int   pin_host;        /* tron_recon_radial2d[_range]: hipHostRegister the caller's buffers for the call (the
                              reference pins its output with cudaMallocHost, src/tron.cu:967); default 1.  Only a buffer
                              of >= 32 MiB above the program break (a mapping of its own, not the brk heap) is registered;
                              any other, and one that cannot be registered (already pinned by the caller, ...), is copied
                              as pageable memory */
    int   cgnr_consistent; /* CGNR with linear angles: 0 = each operator keeps the reference's own convention (grid
                              src/tron.cu:509, degrid :555 -- not a matched pair), 1 = the forward operator
                              inside the iteration uses the gridding convention.  Golden angle: no effect */
    int   coil_combine;    /* adjoint: 0 = coilcombinesos, root-sum-of-squares (src/tron.cu:255-268,764; default);
                              1 = coilcombinewalsh, adaptive combination (src/tron.cu:222-302; call site commented out at :766) */
    int   walsh_patch;     /* coilcombinewalsh's npatch (src/tron.cu:272: "0 works, 1 good, 3 better", :766); default 1 */
} tron_config;

/* Everything main() derives from the input header and the flags (src/tron.cu:76-79,
   897-962), i.e. the remaining file-static globals plus the output .ra header. */
typedef struct tron_dims {
    int nc, nt;
    int nro, npe1, npe2, npe1work;
    int nx, ny, nz, nxos, nyos, nzos;
    int prof_slide;          /* after the "0 means npe1work" rule (src/tron.cu:920-921) */
    uint64_t out_dims[5];    /* output .ra dims (src/tron.cu:899,930-933,955-958) */
    uint64_t out_bytes;      /* h_outdatasize (src/tron.cu:934,960) */
    uint64_t in_elems;       /* complex elements the input array holds */
} tron_dims;

typedef struct tron_plan tron_plan;

/* Defaults of src/tron.cu:58-87. */
void tron_config_default(tron_config *cfg);

/* main()'s dimension logic, src/tron.cu:905-963.  in_dims = the five dims of the input
   .ra ([nc,nt,nro,npe1,npe2] for the adjoint, [nc,nt,nx,ny,nz] forward).  Fails with
   TRON_ERR_INVALID where the reference asserts (nc odd and != 1, src/tron.cu:963). */
int tron_derive_dims(const tron_config *cfg, const uint64_t in_dims[5], tron_dims *dims);

/* = tron_init(), src/tron.cu:579-606: selects the device, creates streams and rocFFT
   plans, allocates work buffers, uploads the angle / band / deapodisation tables. */
int tron_plan_create(tron_plan **plan, const tron_config *cfg, const tron_dims *dims);

/* = tron_shutdown(), src/tron.cu:608-620 (and destroys the FFT plans the reference leaks). */
int tron_plan_destroy(tron_plan *plan);

/* = recon_radial2d(h_out, h_in), src/tron.cu:726-786: host buffers in, host buffers out,
   all nz slices.  h_in / h_out are the payloads of the input / output .ra files. */
int tron_recon_radial2d(tron_plan *plan, tron_float2 *h_out, const tron_float2 *h_in);

/* The same for slices zfirst .. zfirst+zcount-1 only; results land at the offsets the
   full loop would use (src/tron.cu:739-740,776), so disjoint ranges computed by different
   plans / processes / GPUs assemble the full output. */
int tron_recon_radial2d_range(tron_plan *plan, tron_float2 *h_out, const tron_float2 *h_in,
                              int zfirst, int zcount);

/* The same with BLOCK-relative host pointers (adjoint only): h_in_block points at the first spoke of slice zfirst's
   window (spoke zfirst*prof_slide of the stream, src/tron.cu:738-739), h_out_block at slice zfirst's image.  A process
   that holds only its own share of the stream (one rank per GPU) calls this; the angle index stays
   global (src/tron.cu:630). */
int tron_recon_radial2d_block(tron_plan *plan, tron_float2 *h_out_block, const void *h_in_block,
                              int zfirst, int zcount);

/* = recon_radial2d over several GPUs inside ONE process: the reference's (compiled-out) MULTI_GPU scheme,
   src/tron.cu:582-597,735-736, with contiguous slice blocks instead of its round-robin.  One host worker thread and
   one plan per entry of `devices` (HIP ordinals; NULL = 0..n_devices-1; n_devices <= 0 = every visible device; an
   ordinal may repeat), each reconstructing its block straight into h_out -- no gather, no inter-GPU traffic.
   cfg->device is ignored.  Forward plans (one image) run on the first device. */
int tron_recon_radial2d_multi(const tron_config *cfg, const tron_dims *dims, const int *devices, int n_devices,
                              tron_float2 *h_out, const tron_float2 *h_in);

/* = tron_nufft_adj_radial2d(d_out, d_in, j), src/tron.cu:623-637, batched over slices and
   device resident.  d_in: the whole spoke stream [nc,nt,nro,npe1*npe2] on the device
   (never modified: the density compensation of src/tron.cu:628 is applied on the fly);
   slice z reads spokes z*prof_slide .. +npe1work-1 (src/tron.cu:738-739).
   combine != 0: also runs coilcombinesos (src/tron.cu:764) and writes d_out[nx*ny*(z-zfirst) + id];
   combine == 0: writes the deapodised coil images d_out[nchan*nx*ny*(z-zfirst) + nchan*id + c].
   Asynchronous on the plan's stream; call tron_plan_sync() before reading d_out. */
int tron_nufft_adj_radial2d(tron_plan *plan, void *d_out, const void *d_in,
                            int zfirst, int zcount, int combine);

/* = tron_cgnr_radial2d(d_out, d_in, j, niter), src/tron.cu:665-720: cfg.niter iterations of CGNR (Knopp et al. 2007,
   Alg. 1, density-weighted) per slice, device resident, same buffers and layouts as tron_nufft_adj_radial2d.  The
   reference marks its version "NOT WORKING CORRECTLY YET" (:670); this is the algorithm it cites with the operators it
   wires in -- squared norms in the step sizes, image vectors of the adjoint's output size, the gridding scale divided
   out, the slice's own angle indices in the forward operator (DESIGN.md lists the repairs).  The host entry points
   (tron_recon_radial2d*) take this path when cfg.niter > 0, as recon_radial2d does (:754-755). */
int tron_cgnr_radial2d(tron_plan *plan, void *d_out, const void *d_in, int zfirst, int zcount, int combine);

/* = tron_nufft_radial2d(d_out, d_in, j), src/tron.cu:639-649, for `nimg` images stored
   back to back: d_in[nchan*nx*ny*k + nchan*id + c] -> d_out[nchan*nro*npe*k + nchan*(ro+nro*pe) + c]. */
int tron_nufft_radial2d(tron_plan *plan, void *d_out, const void *d_in, int nimg);

/* = the precompensate kernel, src/tron.cu:405-416: Ram-Lak density compensation of ONE window of
   npe1work spokes, in place, d_nudata[nchan*(nro*pe + ro) + c] *= a*|ro - nro/2| + b.  The
   pipelines above fuse this factor into the gridding kernel's loads and never modify their input;
   this stage entry exists for callers that drive tron_gridradial2d themselves. */
int tron_precompensate(tron_plan *plan, void *d_nudata);

/* = the gridradial2d kernel, src/tron.cu:465-536, for ONE image in the reference's own
   layouts: d_nudata[nchan*(nro*pe + ro) + c] (already density compensated),
   d_udata[nchan*(Y*nxos + X) + c] (centred, not shifted).  skip = the kernel's
   skip_angles argument (= skip_angles + peoffset at src/tron.cu:629-630). */
int tron_gridradial2d(tron_plan *plan, void *d_udata, const void *d_nudata, int skip);

/* = the degridradial2d kernel, src/tron.cu:540-577, ONE image, reference layouts:
   d_udata[nrep*(i*n + j) + c] -> d_nudata[nrep*(ro + nro*pe) + c]. */
int tron_degridradial2d(tron_plan *plan, void *d_nudata, const void *d_udata);

/* Waits for everything the plan has launched. */
int tron_plan_sync(tron_plan *plan);

/* A continuing acquisition on ONE plan: every later call on `plan` grids / degrids with the spoke-angle index starting at
   `skip_angles` -- the reference's `-s` flag, i.e. the kernel argument of src/tron.cu:509, 555 (PHI * float(pe + skip)), which the
   reference can change from call to call at no cost because it evaluates the angles per thread.  Here the angles live in tables
   (the (cos, sin) list, the angle-sorted spoke lists, the gridding kernels' run tables); a plan holds two sets of them, and this
   call builds the idle set for the new angles ON THE DEVICE, asynchronously, on a stream of its own beside whatever the plan has
   queued (host: the (cos, sin) table by libm, a fraction of a millisecond on a few threads).  The next call on the plan waits for
   that build -- not for the queued work -- and reads the new set.  Results are those of a plan created with this skip_angles,
   bit for bit.  Nothing that does not depend on the angles (code objects, Kaiser-Bessel and deapodisation tables, tile orders,
   work buffers) is redone.  Linear angles do not depend on skip_angles: a no-op for golden_angle = 0.
   tron_plan_retarget_times: host seconds the last call took, of which the (cos, sin) table. */
int tron_plan_retarget(tron_plan *plan, int skip_angles);
int tron_plan_retarget_times(const tron_plan *plan, double seconds[2]);
/* Names the gridding kernel(s) the plan's adjoint launches (measurement tooling); a static string. */
const char *tron_plan_grid_kernel_name(const tron_plan *plan);
/* Names the degridding kernel the plan's most recent forward launch ran ("" before the first one): degrid_stream_kernel
   for launches of 16 images and more on grids of whole 32x32 tiles, degrid_tile_kernel otherwise. */
const char *tron_plan_degrid_kernel_name(const tron_plan *plan);

/* Shader clock in MHz at the end of what the plan has queued: a short spin kernel behind that work reads the shader-clock counter
   against the constant 100 MHz counter (s_memtime / s_memrealtime).  The chip holds its clock down by power, so a
   throughput figure is only comparable with another at a similar clock; bench.py reports it beside the sustained rate.
   Synchronises the plan's gridding stream. */
int tron_plan_shader_clock(tron_plan *plan, double *mhz);

/* Wall-clock seconds tron_plan_create spent on this plan (the reference's one published time, src/RUNME4:219, clocks tron_init
   too: src/tron.cu:973-978): [0] total, [1] HIP runtime + code objects + the process's first stream and first large copy (the first
   plan of a process pays for all of them), [2] tables and their upload, [3] of [2]: the angle-dependent tables -- the (cos, sin) list
   (host libm) and, built on the device, the sorted spoke lists, centre windows and run tables: what tron_plan_retarget rebuilds --,
   [4] device work buffers. */
int tron_plan_create_times(const tron_plan *plan, double seconds[5]);

/* Per-stage device timing with hipEvents on the plan's stream (off by default; costs one
   event pair per launch).  stage: 0 grid, 1 fft, 2 post (crop+deapod+SoS), 3 pre
   (pad+deapod), 4 degrid.  Returns accumulated milliseconds and launch count since the
   last reset; synchronises the plan. */
int tron_plan_timing(tron_plan *plan, int enable);
int tron_plan_timing_get(tron_plan *plan, int stage, double *ms, uint64_t *launches);
int tron_plan_timing_reset(tron_plan *plan);

/* Host-side tables the kernels consume, exposed so the arithmetic that must match the
   reference bit for bit can be checked without a GPU.
   trig: (cos, sin) of the angle of stream spoke i, src/tron.cu:509-511 (adjoint) or
         :555-559 (forward); n = number of table entries written.
   band: Rlo | Rhi<<16 per grid point in centred raster order, src/tron.cu:498-502.
   deapod: 1/w per pixel with w as in src/tron.cu:395-400 (n = nx, sigma = gridos for the
         adjoint; n = nxos, sigma = 1 forward). */
int tron_host_trig_table(const tron_config *cfg, const tron_dims *dims, float *cos_sin, size_t n);
int tron_host_band_table(int nxos, float kernwidth, uint32_t *band);
int tron_host_deapod_table(int n, float kernwidth, float sigma, float *inv_weight);
/* CPUs of the NUMA node of a PCI function ("0000:c1:00.0") read from a sysfs tree rooted at sysroot ("/sys"): what
   tron_recon_radial2d_multi binds each per-GPU worker thread to.  Returns the count, 0 if the node is unknown, -1 on bad input. */
int tron_host_numa_cpulist(const char *sysroot, const char *pci_bus_id, int *cpus, int max_cpus);

/* Small device-memory helpers so a host language without HIP bindings can drive the
   device-resident entry points. */
int tron_device_count(int *count);
/* PCI bus id of a device ("0000:c1:00.0", at least 16 bytes): the key of its NUMA node in sysfs, see tron_host_numa_cpulist. */
int tron_device_pci_bus_id(int device, char *buf, int len);
int tron_device_malloc(void **d_ptr, size_t bytes);
int tron_device_free(void *d_ptr);
int tron_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes);
int tron_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes);

const char *tron_last_error(void);
const char *tron_version(void);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif /* TRON_HIP_H */
