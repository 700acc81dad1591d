// Host-internal declarations of libtronhip (tables, error reporting).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/tron_hip.h"

namespace tron {

// records a printf-style message as the calling thread's last error and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// Tuning / debugging switches of the library are read from the environment ONLY when TRON_TUNING=1 is set
// as well: a library caller's behaviour does not depend on stray TRON_* variables.
const char *tuning_env(const char *name);
bool debug_token(const char *name, std::string *value = nullptr);   // a token of TRON_DEBUG (`sync`, `poison`, `pin_any`, `cold_fault=<path>`)

float grid_spoke_angle(int pe, int npe, int skip, int golden);
float degrid_spoke_angle(int pe, int npe, int skip, int golden);
size_t trig_table_size(const tron_config &cfg, const tron_dims &d);
void build_trig_table(const tron_config &cfg, const tron_dims &d, float *cos_sin, size_t n);
void build_trig_table_mt(const tron_config &cfg, const tron_dims &d, float *cos_sin, size_t n, int max_threads);
float exact_fmodf_pos(float x, float y);      // fmodf for 0 <= x, 0 < y, x / y < 2^28 (the golden-angle wrap), bit for bit
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
// centre kernel: per block of `groups` the angular window of its run of a window's angle-sorted spoke list: out[4 g] = (lo, hi, all | wrap << 1, share)
void build_centre_group_windows(const int *groups, int ngroups, float W, float *out);
double kb_poly_fit(float kernwidth, float *poly, int nterms);
void dcf_constants(int nro, int npe1work, float *a, float *b);
float grid_scale(int nxos, int npe);

}  // namespace tron
