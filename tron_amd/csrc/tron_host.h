// Host-internal declarations of libtronhip (tables, error reporting).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/tron_hip.h"

namespace tron {

// records a printf-style message as the calling thread's last error and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Tuning / debugging switches of the library are read from the environment ONLY when TRON_TUNING=1 is set
// as well: a library caller's behaviour does not depend on stray TRON_* variables.
const char *tuning_env(const char *name);

float grid_spoke_angle(int pe, int npe, int skip, int golden);
float degrid_spoke_angle(int pe, int npe, int skip, int golden);
size_t trig_table_size(const tron_config &cfg, const tron_dims &d);
void build_trig_table(const tron_config &cfg, const tron_dims &d, float *cos_sin, size_t n);
void build_trig_table_window(int npe, int skip, int golden, float *cos_sin);
void build_band_table(int nxos, float kernwidth, uint32_t *band);
bool scatter_band_is_analytic(int nxos, float kernwidth, const uint32_t *band);   // (u - W)^2 <= X^2 + Y^2 <= (u + W)^2 reproduces the table (tron_grid_scatter.hip)
void build_deapod_table(int n, float kernwidth, float sigma, float *inv_weight);
void build_deapod_table_rect(int rows, int cols, float kernwidth, float sigma, float *inv_weight);
void build_tile_order(int nxos, int tile, std::vector<int> &order);
// run-length classes of the streaming degridding kernel over build_tile_order's centre-first positions: a tile expected to hold
// `est` samples per image takes runs of 2^c images, c the smallest with est * 2^c >= target (c <= 4); end[c] = first position
// past class c (classes are made monotone along the order)
void build_degrid_groups(int nxos, int tile, int npe, int nro, int target, int end[4]);
void build_split_tile_order(int nxos, int tile, int npe, float W, int target, int max_parts,
                            std::vector<int> &order, std::vector<int> &slots);
bool build_centre_relief_order(int nxos, int tile, int npe, float W, int max_parts, int &inner_r0, std::vector<int> &order, std::vector<int> &slots,
                               int target_records = 1400);
float kb_beta(float kernwidth);
double kb_peak(float kernwidth);      // the window's value at 0
// arc gridding kernel: Kaiser-Bessel table over the signed distance from a 2x2 block's first column, two windows per entry
// (tron_hostmath.cpp); kb_pair_lut_scale = pieces per grid unit for a table of `cap` entries (0: this width has none),
// build_kb_pair_lut fills coef[3][cap][2] and returns the entries used
int kb_pair_lut_scale(float kernwidth, int cap);
int build_kb_pair_lut(float kernwidth, int cap, float *coef, float *scale, int *bias, double *err);
// centre kernel: per window and block of `groups` its run of the angle-sorted spoke list (first | count << 16)
void build_centre_windows(const float *phi, size_t nwindows, int npe, const int *groups, int ngroups, float W, uint32_t *out);
// arc gridding kernel: per window of npe spokes (cos_sin + 2 * stride * z), the spokes in ascending line angle (mod pi)
void build_arc_tables(const float *cos_sin, size_t nwindows, size_t stride, int npe, unsigned short *order, float *phi);
double kb_poly_fit(float kernwidth, float *poly, int nterms);
void dcf_constants(int nro, int npe1work, float *a, float *b);
float grid_scale(int nxos, int npe);

}  // namespace tron
