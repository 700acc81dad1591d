// `tron` -- command-line driver with the reference's interface (davidssmith/TRON,
// src/tron.cu:790-995): same getopt string, same defaults, same dimension logic, same input /
// output .ra conventions and exit codes, on top of the C ABI of include/tron_hip.h.
//
//   tron [-3aGhv] [-B blocks] [-d prof_slide] [-g gpu] [-i niter] [-k width] [-o gridos]
//        [-r nro] [-s skip_angles] [-T threads] [-u data_undersamp] <infile.ra> [outfile.ra]
//
// Extensions, via ONE environment variable so the flag set stays the reference's:
//   TRON_OPTIONS=kb=fast|exact,gpus=N,combine=walsh|sos,patch=N,cgnr_consistent=1,pin=0|1
//     kb     Kaiser-Bessel evaluation.  fast (default): tabulated / polynomial window, each kernel's own summation order, within 1e-5
//            relative L2 of the reference arithmetic (measured ~1e-7); exact: the reference's expression tree and summation order, bit for bit
//     gpus   one worker thread + plan per GPU (as -g all / -g 0,1,...)
// A complex-half input (eltype 4, elbyte 4) is accepted for -a and gridded from half storage.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rawarray.h"
#include "../../include/tron_hip.h"

static void usage()
{
    fputs("Trajectory-optimized Non-uniform Fast Fourier Transform (MI355X / HIP build)\n"
          "Usage: tron [-3aGhv] [-B blocks] [-d prof_slide] [-g gpu] [-i niter] [-k width] [-o gridos]\n"
          "            [-r nro] [-s skip_angles] [-T threads] [-u data_undersamp] <infile.ra> [outfile.ra]\n"
          "  -3                 3D koosh ball trajectory (dimension bookkeeping only)\n"
          "  -a                 adjoint operation (gridding); default is forward (degridding)\n"
          "  -B blocks          accepted for compatibility, ignored\n"
          "  -d prof_slide      phase encodes to slide between slices (helical / sliding window)\n"
          "  -g n               GPU device to use (default: 0); `-g all` or `-g 0,1,..`: shard the slices over several GPUs\n"
          "  -G                 golden angle radial\n"
          "  -h                 show this help\n"
          "  -i niter           CGNR iterations (adjoint only; 0 = plain gridding)\n"
          "  -k width           half-width of the gridding kernel (default 2)\n"
          "  -o gridos          grid oversampling factor (default 2)\n"
          "  -r nro             number of readout points (taken from the input file)\n"
          "  -s skip_angles     angle index of the first phase encode\n"
          "  -T threads         accepted for compatibility, ignored\n"
          "  -u data_undersamp  input data undersampling factor (default 1)\n"
          "  -v                 verbose output\n",
          stderr);
}

int main(int argc, char *argv[])
{
    tron_config cfg;
    tron_config_default(&cfg);
    bool multi_gpu = false;              // -g all | -g 0,1,... | TRON_OPTIONS gpus=n: one worker thread + plan per device
    std::vector<int> gpu_list;
    int c;
    opterr = 0;
    while ((c = getopt(argc, argv, "3aB:d:g:Ghi:k:o:r:s:T:u:v")) != -1) {   // src/tron.cu:822
        switch (c) {
            case '3': cfg.koosh = 1; break;
            case 'a': cfg.adjoint = 1; break;
            case 'B': cfg.blocks = atoi(optarg); break;
            case 'd': cfg.prof_slide = atoi(optarg); break;
            case 'g':                                                          // src/tron.cu:838: cudaSetDevice(n); here also
                if (strcmp(optarg, "all") == 0) {                              // "all" or a list "0,1,2": one worker per device
                    multi_gpu = true;
                } else if (strchr(optarg, ',')) {
                    multi_gpu = true;
                    for (const char *q = optarg; *q;) {
                        gpu_list.push_back(atoi(q));
                        const char *c = strchr(q, ',');
                        if (!c) break;
                        q = c + 1;
                    }
                } else {
                    cfg.device = atoi(optarg);
                }
                break;
            case 'G': cfg.golden_angle = 1; break;
            case 'h': usage(); return 1;                                       // src/tron.cu:843-845
            case 'i': cfg.niter = atoi(optarg); break;
            case 'k': cfg.kernwidth = atof(optarg); break;
            case 'o': cfg.gridos = atof(optarg); break;
            case 'u': cfg.data_undersamp = atof(optarg); break;
            case 'r': break;                                                   // parsed, then overwritten (src/tron.cu:859,909)
            case 's': cfg.skip_angles = atoi(optarg); break;
            case 'T': cfg.threads = atoi(optarg); break;
            case 'v': cfg.verbose = 1; break;
            default: usage(); return 1;
        }
    }
    if (argc == optind) {                                                      // src/tron.cu:878-881
        usage();
        return 1;
    }
    const char *infile = argv[optind];
    const char *outfile = optind + 1 < argc ? argv[optind + 1] : "img_tron.ra";   // src/tron.cu:877
    int pin_opt = -1;                                                          // TRON_OPTIONS pin=0|1 (default: see the streamed path below)
    // What the reference's getopt string has no letter for (its flags are kept exactly, src/tron.cu:813): TRON_OPTIONS, a comma-separated
    // list -- kb=exact|fast (Kaiser-Bessel mode, default fast), gpus=N (one worker per GPU, as -g all), combine=walsh|sos, patch=N
    // (Walsh patch half-width), cgnr_consistent=1, pin=0|1 (register the host buffers for the copies).  They become tron_config fields; the library itself reads no such variable.
    if (const char *opts = getenv("TRON_OPTIONS")) {
        std::string all(opts);
        size_t pos = 0;
        while (pos <= all.size()) {
            const size_t end = std::min(all.find(',', pos), all.size());
            const std::string tok = all.substr(pos, end - pos);
            const size_t eq = tok.find('=');
            const std::string key = tok.substr(0, eq), val = eq == std::string::npos ? std::string() : tok.substr(eq + 1);
            if (key == "kb") cfg.kb_mode = val == "exact" ? TRON_KB_EXACT : TRON_KB_FAST;
            else if (key == "cgnr_consistent") cfg.cgnr_consistent = atoi(val.c_str()) != 0;
            else if (key == "combine") cfg.coil_combine = val == "walsh" ? 1 : 0;
            else if (key == "patch") cfg.walsh_patch = atoi(val.c_str());
            else if (key == "pin") pin_opt = atoi(val.c_str()) != 0 ? 1 : 0;
            else if (key == "gpus") {
                if (atoi(val.c_str()) > 1 && gpu_list.empty()) {
                    multi_gpu = true;
                    for (int g = 0; g < atoi(val.c_str()); ++g) gpu_list.push_back(g);
                }
            } else if (!key.empty()) {
                fprintf(stderr, "tron: TRON_OPTIONS: unknown option '%s' (kb, gpus, combine, patch, cgnr_consistent, pin)\n", key.c_str());
                return 1;
            }
            pos = end + 1;
        }
    }

#define VPRINT(...) do { if (cfg.verbose) printf(__VA_ARGS__); } while (0)

    VPRINT("Reading %s\n", infile);
    ra_t hdr;
    if (ra_read_header(&hdr, infile) != 0) return 1;
    if (hdr.ndims != 5) {                                                      // assert at src/tron.cu:892
        fprintf(stderr, "tron: %s has %llu dimensions, expected 5 ([nc,nt,nro,npe1,npe2] or [nc,nt,nx,ny,nz])\n", infile, (unsigned long long)hdr.ndims);
        ra_free(&hdr);
        return 1;
    }
    if (hdr.eltype == RA_TYPE_COMPLEX && hdr.elbyte == 4 && cfg.adjoint)
        cfg.input_half = 1;
    else if (!(hdr.eltype == RA_TYPE_COMPLEX && hdr.elbyte == 8)) {
        fprintf(stderr, "tron: %s must hold complex64 data (eltype 4, elbyte 8), found eltype %llu elbyte %llu\n", infile,
                (unsigned long long)hdr.eltype, (unsigned long long)hdr.elbyte);
        ra_free(&hdr);
        return 1;
    }
    tron_dims dims;
    if (tron_derive_dims(&cfg, hdr.dims, &dims) != TRON_OK) {
        fprintf(stderr, "tron: %s\n", tron_last_error());
        ra_free(&hdr);
        return 1;
    }
    if (hdr.size < dims.in_elems * (cfg.input_half ? 4 : 8)) {
        fprintf(stderr, "tron: %s holds %llu bytes, its dimensions need %llu\n", infile, (unsigned long long)hdr.size,
                (unsigned long long)(dims.in_elems * (cfg.input_half ? 4 : 8)));
        ra_free(&hdr);
        return 1;
    }
    uint64_t hdr_dims[5];
    memcpy(hdr_dims, hdr.dims, sizeof(hdr_dims));
    ra_free(&hdr);
    // the FILE must hold what its header promises before anything is allocated by that promise; and an output that IS the input
    // (the reference's read-all-then-write order allows it, src/tron.cu:887-983) must not be truncated while it is still being read
    bool same_file = false;
    {
        struct stat si, so;
        if (stat(infile, &si) != 0) {
            fprintf(stderr, "tron: cannot stat %s\n", infile);
            return 1;
        }
        // (a pipe or a device has no size to check: the reader finds a short stream when it ends)
        const uint64_t need = (6 + 5) * sizeof(uint64_t) + dims.in_elems * (cfg.input_half ? 4 : 8);
        if (S_ISREG(si.st_mode) && (uint64_t)si.st_size < need) {
            fprintf(stderr, "tron: %s is %llu bytes long, its header needs %llu\n", infile, (unsigned long long)si.st_size, (unsigned long long)need);
            return 1;
        }
        same_file = stat(outfile, &so) == 0 && so.st_dev == si.st_dev && so.st_ino == si.st_ino;
    }

    struct timespec t0, t1, tp;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    auto since = [](const struct timespec &a) {
        struct timespec b;
        clock_gettime(CLOCK_MONOTONIC, &b);
        return (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
    };
    const uint64_t in_off = (6 + 5) * sizeof(uint64_t);                        // header of a 5-dimensional array
    const uint64_t in_bytes = dims.in_elems * (cfg.input_half ? 4 : 8);
    ra_t in;
    memset(&in, 0, sizeof(in));
    in.eltype = RA_TYPE_COMPLEX;
    in.elbyte = cfg.input_half ? 4 : 8;
    in.ndims = 5;
    in.size = in_bytes;
    in.dims = static_cast<uint64_t *>(malloc(5 * sizeof(uint64_t)));
    in.data = static_cast<uint8_t *>(malloc(in_bytes ? in_bytes : 1));
    if (!in.dims || !in.data) {
        fprintf(stderr, "unable to allocate memory for data\n");
        return 1;
    }
    memcpy(in.dims, hdr_dims, 5 * sizeof(uint64_t));
    // The payload is read by a helper thread in pieces, in file order, while this thread brings up the GPU (runtime
    // initialisation + plan: work buffers, tables); `ready` = bytes of the payload that are in memory.  The adjoint then
    // runs block by block as soon as a block's spokes are in, and a second helper writes each finished block of images
    // while the next one is computed (src/tron.cu:890, 967-982 read everything, compute everything, write everything).
    std::atomic<uint64_t> ready(0);
    std::atomic<int> read_rc(0);
    double read_s = 0.0;
    std::thread reader([&]() {
        struct timespec a0;
        clock_gettime(CLOCK_MONOTONIC, &a0);
        const uint64_t piece = (uint64_t)32 << 20;
        for (uint64_t done = 0; done < in_bytes;) {
            const uint64_t n = std::min(piece, in_bytes - done);
            const int rc = ra_read_range(infile, in_off, done, n, in.data + done);
            if (rc) { read_rc = rc; break; }
            done += n;
            ready.store(done, std::memory_order_release);
        }
        read_s = since(a0);
    });
    auto wait_for = [&](uint64_t bytes) {
        while (ready.load(std::memory_order_acquire) < bytes && read_rc.load() == 0) usleep(200);
        return read_rc.load() == 0;
    };

    ra_t out;
    memset(&out, 0, sizeof(out));
    out.flags = 0;                                                             // src/tron.cu:897-902
    out.eltype = RA_TYPE_COMPLEX;
    out.elbyte = 8;
    out.ndims = 5;
    out.size = dims.out_bytes;
    out.dims = static_cast<uint64_t *>(malloc(5 * sizeof(uint64_t)));
    out.data = static_cast<uint8_t *>(calloc(dims.out_bytes ? dims.out_bytes : 1, 1));
    // the streamed path (below): blocks of slices; the plan's batches, work buffers and staging are sized for ONE block, and the
    // copies go from / to pageable memory -- registering every block's 40 + 30 MB for a copy of 1 ms each cost more than it
    // returned (whole-body shape, round 6: profiles/round6_wholebody_cli.log; TRON_OPTIONS pin=1 brings it back)
    const bool will_stream = !multi_gpu && cfg.adjoint && dims.nz > 1 && !same_file;
    const int nblocks = std::max(1, std::min(16, dims.nz / 48));
    if (will_stream) {
        if (cfg.chunk_slices <= 0) cfg.chunk_slices = (dims.nz + nblocks - 1) / nblocks;
        cfg.pin_host = pin_opt == 1 ? 1 : 0;
    } else if (pin_opt >= 0) {
        cfg.pin_host = pin_opt;
    }
    // the output's pages are touched by a helper while the GPU comes up (calloc hands out untouched pages: the first download would
    // fault them in one by one, 120 k of them for the whole-body volume)
    std::thread toucher([&]() {
        if (!out.data) return;
        for (uint64_t o = 0; o < dims.out_bytes; o += 4096) reinterpret_cast<volatile uint8_t *>(out.data)[o] = 0;
    });
    tron_plan *plan = nullptr;
    int rc = TRON_OK;
    if (!(out.dims && out.data)) rc = TRON_ERR_NOMEM;
    else if (!multi_gpu) rc = tron_plan_create(&plan, &cfg, &dims);        // multi-GPU: every worker creates its own plan
    toucher.join();
    clock_gettime(CLOCK_MONOTONIC, &tp);
    const double plan_s = (tp.tv_sec - t0.tv_sec) + 1e-9 * (tp.tv_nsec - t0.tv_nsec);
    if (!out.dims || !out.data) {
        reader.join();
        fprintf(stderr, "tron: cannot allocate %llu bytes for the output\n", (unsigned long long)dims.out_bytes);
        return 1;
    }
    memcpy(out.dims, dims.out_dims, 5 * sizeof(uint64_t));
    if (!wait_for(std::min<uint64_t>(in_bytes, 16))) {
        reader.join();
        tron_plan_destroy(plan);
        ra_free(&in);
        ra_free(&out);
        return 1;
    }
    VPRINT("Plan time: %.3f s (the payload is being read meanwhile)\n", plan_s);
    if (!cfg.input_half) {
        const float *f = reinterpret_cast<const float *>(in.data);
        VPRINT("Sanity check: indata[0] = %f + %f i\n", f[0], f[1]);
    }
    VPRINT("indims = {%llu, %llu, %llu, %llu, %llu}\n", (unsigned long long)in.dims[0], (unsigned long long)in.dims[1],
           (unsigned long long)in.dims[2], (unsigned long long)in.dims[3], (unsigned long long)in.dims[4]);
    VPRINT("WARNING: Assuming square Cartesian dimensions for now.\n");

    VPRINT("Running reconstruction ...\n ");
    double write_s = 0.0;
    bool streamed = false;
    int wrc = 0;
    if (rc == TRON_OK && will_stream) {
        // ---- streamed: blocks of slices, each started when its spokes have been read, each written when it is done ----
        streamed = true;
        const uint64_t out_off = ra_data_offset(&out);
        wrc = ra_write_header(&out, outfile);
        const size_t spoke_bytes = (size_t)dims.nro * dims.nc * dims.nt * (cfg.input_half ? 4 : 8);
        const size_t img_bytes = (size_t)dims.nt * dims.nx * dims.ny * sizeof(tron_float2);
        std::atomic<int> blocks_done(0);
        std::atomic<bool> stop(false);
        std::thread writer([&]() {
            struct timespec w0;
            for (int k = 0; k < nblocks && wrc == 0; ++k) {
                while (blocks_done.load(std::memory_order_acquire) <= k && !stop.load()) usleep(200);
                if (blocks_done.load(std::memory_order_acquire) <= k) break;
                clock_gettime(CLOCK_MONOTONIC, &w0);
                const int z0 = (int)((long long)k * dims.nz / nblocks), z1 = (int)((long long)(k + 1) * dims.nz / nblocks);
                wrc = ra_write_range(outfile, out_off, (uint64_t)z0 * img_bytes, (uint64_t)(z1 - z0) * img_bytes, out.data + (size_t)z0 * img_bytes);
                write_s += since(w0);
            }
        });
        for (int k = 0; k < nblocks && rc == TRON_OK; ++k) {
            const int z0 = (int)((long long)k * dims.nz / nblocks), z1 = (int)((long long)(k + 1) * dims.nz / nblocks);
            const uint64_t need = ((uint64_t)(z1 - 1) * dims.prof_slide + dims.npe1work) * spoke_bytes;
            if (need > in_bytes) {     // the library reports the out-of-range window with the reference's wording
                rc = tron_recon_radial2d(plan, reinterpret_cast<tron_float2 *>(out.data), reinterpret_cast<const tron_float2 *>(in.data));
                break;
            }
            if (!wait_for(need)) { rc = TRON_ERR_INVALID; break; }
            rc = tron_recon_radial2d_block(plan, reinterpret_cast<tron_float2 *>(out.data + (size_t)z0 * img_bytes),
                                           in.data + (size_t)z0 * dims.prof_slide * spoke_bytes, z0, z1 - z0);
            if (rc == TRON_OK) blocks_done.store(k + 1, std::memory_order_release);
        }
        stop = true;
        writer.join();
        reader.join();
    } else {
        reader.join();
        if (read_rc.load() != 0) rc = rc == TRON_OK ? TRON_ERR_INVALID : rc;
        if (rc == TRON_OK && multi_gpu)
            rc = tron_recon_radial2d_multi(&cfg, &dims, gpu_list.empty() ? nullptr : gpu_list.data(), (int)gpu_list.size(),
                                           reinterpret_cast<tron_float2 *>(out.data), reinterpret_cast<const tron_float2 *>(in.data));
        else if (rc == TRON_OK)
            rc = tron_recon_radial2d(plan, reinterpret_cast<tron_float2 *>(out.data), reinterpret_cast<const tron_float2 *>(in.data));
    }
    if (read_rc.load() != 0) {
        tron_plan_destroy(plan);
        ra_free(&in);
        ra_free(&out);
        if (streamed) unlink(outfile);            // a header without its images is no result
        return 1;
    }
    if (rc != TRON_OK) {
        fprintf(stderr, "tron: %s\n", tron_last_error());
        tron_plan_destroy(plan);
        ra_free(&in);
        ra_free(&out);
        if (streamed) unlink(outfile);
        return rc == TRON_ERR_UNSUPPORTED ? 2 : 1;
    }
    tron_plan_destroy(plan);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    VPRINT("Read time: %.3f s (beside the plan and the first blocks)\n", read_s);
    VPRINT("Elapsed time: %.2f s\n", (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec));

    VPRINT("Saving result to %s\n", outfile);
    if (streamed) {
        rc = wrc;
        VPRINT("Write time: %.3f s (beside the reconstruction, block by block)\n", write_s);
    } else {
        struct timespec tr0;
        clock_gettime(CLOCK_MONOTONIC, &tr0);
        rc = ra_write(&out, outfile);
        VPRINT("Write time: %.3f s\n", since(tr0));
    }
    VPRINT("Cleaning up.\n");
    ra_free(&in);
    ra_free(&out);
    return rc == 0 ? 0 : 1;
}
