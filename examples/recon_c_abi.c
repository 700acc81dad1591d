/* Minimal C99 client of the C ABI (include/tron_hip.h, include/rawarray.h): what a TRON maintainer's
   main() does after getopt -- read a .ra file, run recon_radial2d's replacement, write a .ra file.
     gcc -std=c99 -Iinclude examples/recon_c_abi.c -Ltron_amd/lib -ltronhip -Wl,-rpath,$PWD/tron_amd/lib -o recon_c_abi
     ./recon_c_abi -a in.ra out.ra        (adjoint, golden angle)      ./recon_c_abi in.ra out.ra   (forward)
     ./recon_c_abi -a -i 3 in.ra out.ra   (CGNR, 3 iterations)         ./recon_c_abi -a -m in.ra out.ra   (every GPU of the node)
     ./recon_c_abi -a -s 4020 in.ra out.ra   (a LATER batch of a continuing golden-angle acquisition on the same plan: the plan is made
                                              for angle index 0, as for the batch before, then moved to index 4020 by tron_plan_retarget
                                              -- what the reference does by passing another skip_angles to its kernels, src/tron.cu:629-630 --
                                              and the file is reconstructed with those angles: the bytes of `tron -a -G -s 4020`) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rawarray.h"
#include "tron_hip.h"

int main(int argc, char **argv)
{
    int adjoint = 0, arg = 1, niter = 0, multi = 0, later = 0, skip = 0;
    while (arg < argc && argv[arg][0] == '-') {
        if (strcmp(argv[arg], "-a") == 0) adjoint = 1;
        else if (strcmp(argv[arg], "-m") == 0) multi = 1;          /* tron_recon_radial2d_multi: one worker + plan per GPU */
        else if (strcmp(argv[arg], "-i") == 0 && arg + 1 < argc) niter = atoi(argv[++arg]);
        else if (strcmp(argv[arg], "-s") == 0 && arg + 1 < argc) { later = 1; skip = atoi(argv[++arg]); }
        else break;
        ++arg;
    }
    if (argc - arg < 2) {
        fprintf(stderr, "usage: %s [-a] [-i niter] [-m] [-s skip_angles] in.ra out.ra\n", argv[0]);
        return 1;
    }
    ra_t in, out;
    if (ra_read(&in, argv[arg]) != 0) return 1;
    if (in.ndims != 5 || in.eltype != RA_TYPE_COMPLEX || in.elbyte != 8) {
        fprintf(stderr, "expected a 5-D complex64 array\n");
        return 1;
    }
    tron_config cfg;
    tron_config_default(&cfg);            /* the reference's defaults, src/tron.cu:58-87 */
    cfg.adjoint = adjoint;
    cfg.golden_angle = 1;
    cfg.niter = niter;                    /* > 0: CGNR, src/tron.cu:754-755 */
    tron_dims dims;
    tron_plan *plan = NULL;
    if (tron_derive_dims(&cfg, in.dims, &dims) != TRON_OK || (!multi && tron_plan_create(&plan, &cfg, &dims) != TRON_OK)) {
        fprintf(stderr, "%s\n", tron_last_error());
        return 1;
    }
    memset(&out, 0, sizeof(out));
    out.eltype = RA_TYPE_COMPLEX;
    out.elbyte = 8;
    out.ndims = 5;
    out.size = dims.out_bytes;
    out.dims = (uint64_t *)malloc(5 * sizeof(uint64_t));
    out.data = (uint8_t *)calloc(dims.out_bytes ? dims.out_bytes : 1, 1);
    memcpy(out.dims, dims.out_dims, 5 * sizeof(uint64_t));
    if (later && !multi) {
        /* the batch before this one (angle index 0) ... */
        if (tron_recon_radial2d(plan, (tron_float2 *)out.data, (const tron_float2 *)in.data) != TRON_OK ||
            /* ... and this one: new angle tables, built on the device beside whatever the plan still has queued */
            tron_plan_retarget(plan, skip) != TRON_OK) {
            fprintf(stderr, "%s\n", tron_last_error());
            return 1;
        }
    }
    if ((multi ? tron_recon_radial2d_multi(&cfg, &dims, NULL, 0, (tron_float2 *)out.data, (const tron_float2 *)in.data)
               : tron_recon_radial2d(plan, (tron_float2 *)out.data, (const tron_float2 *)in.data)) != TRON_OK) {
        fprintf(stderr, "%s\n", tron_last_error());
        return 1;
    }
    tron_plan_destroy(plan);
    if (ra_write(&out, argv[arg + 1]) != 0) return 1;
    printf("%s: %d slice(s) of %d x %d -> %s (libtronhip %s)\n", adjoint ? "adjoint" : "forward", dims.nz, dims.nx, dims.ny,
           argv[arg + 1], tron_version());
    ra_free(&in);
    ra_free(&out);
    return 0;
}
