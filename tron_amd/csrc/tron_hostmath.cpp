// Host-side arithmetic of libtronhip that has to agree with the reference BIT FOR BIT:
// the dimension logic of main(), the spoke angles, the per-point radial band and the
// deapodisation weights.  These are evaluated once per plan on the host -- with the same libm
// (sincosf, fmodf, hypotf, sinhf) an IEEE host build of the reference would use -- and uploaded
// as tables, so device transcendental rounding never enters the comparison.
//
// Build with -ffp-contract=off and without fast-math.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tron_hip.h"
#include "tron_host.h"

namespace tron {

static const float kGoldenAngle = 1.9416089796736116f;   // PHI, src/tron.cu:90

// fmodf(x, y) for finite x >= 0, y > 0 with x / y < 2^28: the same VALUE as libm's -- fmod is exact by definition (x - trunc(x / y) y,
// always representable), so any exact evaluation returns the same bits -- at a tenth of glibc's cost (its fmodf subtracts bit by bit: 40 ns
// for the quotients ~ 10^5 the golden-angle index produces, and the (cos, sin) table of a 256-slice plan is 10^5 of them).
// In double: q = trunc(x / y) is within one of the true quotient (x / y carries a relative error of 2^-53, q < 2^28), q y has at most
// 28 + 24 bits and is exact, so is the difference; one correction step lands in [0, y).  tests/test_host.py compares it with fmodf.
float exact_fmodf_pos(float x, float y)
{
    const double xd = x, yd = y;
    double r = xd - floor(xd / yd) * yd;
    if (r < 0.0) r += yd;
    if (r >= yd) r -= yd;
    return (float)r;
}

// src/tron.cu:372-378
static float wrap_angle(float x)
{
    const float two_pi = 2.f * M_PI;
    float y = (x >= 0.f && x < 1.0e9f) ? exact_fmodf_pos(x, two_pi) : fmodf(x, two_pi);
    return y < 0.f ? y + two_pi : y;
}

// Angle of spoke `pe` for the gridding kernel: src/tron.cu:509.  `skip` is the kernel argument,
// i.e. skip_angles + peoffset (src/tron.cu:629-630).
float grid_spoke_angle(int pe, int npe, int skip, int golden)
{
    float t;
    if (golden)
        t = wrap_angle(kGoldenAngle * (float)(pe + skip));
    else
        t = pe * 2.0f * M_PI / (float)npe + M_PI * 0.5f;      // double expression, rounded once
    return t;
}

// Angle of spoke `pe` for the degridding kernel: src/tron.cu:555.
float degrid_spoke_angle(int pe, int npe, int skip, int golden)
{
    float t;
    if (golden)
        t = wrap_angle(kGoldenAngle * (pe + skip));
    else
        t = pe * M_PI / (float)npe;
    return t;
}

// (cos, sin) table for the plan's direction.  Adjoint + golden angle: one entry per spoke of
// the whole stream, because slice z uses angle indices pe + skip_angles + z*prof_slide
// (src/tron.cu:629-630, 738); otherwise one entry per spoke of a window.
size_t trig_table_size(const tron_config &cfg, const tron_dims &d)
{
    if (cfg.adjoint && cfg.golden_angle)
        return (size_t)(d.nz - 1) * d.prof_slide + d.npe1work;
    return (size_t)d.npe1work;
}

void build_trig_table(const tron_config &cfg, const tron_dims &d, float *cos_sin, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        float t = cfg.adjoint ? grid_spoke_angle((int)i, d.npe1work, cfg.skip_angles, cfg.golden_angle)
                              : degrid_spoke_angle((int)i, d.npe1work, cfg.skip_angles, cfg.golden_angle);
        float s, c;
        sincosf(t, &s, &c);                                   // src/tron.cu:511, 559
        cos_sin[2 * i] = c;
        cos_sin[2 * i + 1] = s;
    }
}

// The same table on up to `max_threads` host threads (tron_plan_retarget: 100 k entries between two batches of a 3.5 ms step);
// every entry is computed exactly as above, whichever thread does it.
void build_trig_table_mt(const tron_config &cfg, const tron_dims &d, float *cos_sin, size_t n, int max_threads)
{
    const size_t per_thread = 8192;                           // (~0.2 ms of libm each: below that a thread costs more than it saves)
    int nth = (int)std::min<size_t>((n + per_thread - 1) / per_thread, (size_t)std::max(1, max_threads));
    if (nth <= 1) { build_trig_table(cfg, d, cos_sin, n); return; }
    auto part = [&](int t) {
        const size_t i0 = n * t / nth, i1 = n * (t + 1) / nth;
        for (size_t i = i0; i < i1; ++i) {
            float a = cfg.adjoint ? grid_spoke_angle((int)i, d.npe1work, cfg.skip_angles, cfg.golden_angle)
                                  : degrid_spoke_angle((int)i, d.npe1work, cfg.skip_angles, cfg.golden_angle);
            float sn, cn;
            sincosf(a, &sn, &cn);
            cos_sin[2 * i] = cn;
            cos_sin[2 * i + 1] = sn;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nth; ++t) th.emplace_back(part, t);
    part(0);
    for (auto &x : th) x.join();
}

void build_trig_table_window(int npe, int skip, int golden, float *cos_sin)
{
    for (int i = 0; i < npe; ++i) {
        float s, c;
        sincosf(grid_spoke_angle(i, npe, skip, golden), &s, &c);
        cos_sin[2 * i] = c;
        cos_sin[2 * i + 1] = s;
    }
}

// Rows [0, n) of a table dealt to a few host threads (plan creation: the band and deapodisation tables are 260 k and 65 k libm
// calls at the metric shape, 5 of the 40 ms a plan took in round 5); every entry is computed by the same expression whichever
// thread does it.  Small tables stay on the caller's thread.
template <typename F>
static void for_rows(int n, size_t work_per_row, F body)
{
    static const int hw = std::max(1, std::min(8, (int)std::thread::hardware_concurrency() / 2));
    const int nth = (int)std::min<size_t>((size_t)hw, std::max<size_t>(1, (size_t)n * work_per_row / 32768));
    if (nth <= 1 || n < 2 * nth) { for (int r = 0; r < n; ++r) body(r); return; }
    std::vector<std::thread> th;
    auto part = [&](int t) { for (int r = (int)((long long)n * t / nth); r < (int)((long long)n * (t + 1) / nth); ++r) body(r); };
    for (int t = 1; t < nth; ++t) th.emplace_back(part, t);
    part(0);
    for (auto &x : th) x.join();
}

// Radial band of every Cartesian point, src/tron.cu:498-502: a sample of signed radius r on any
// spoke can contribute to the point only if Rlo <= |r| <= Rhi.
void build_band_table(int nxos, float kernwidth, uint32_t *band)
{
    const int h = nxos / 2;
    // hypotf(+-X, +-Y) is one value (IEEE: the signs do not enter), so only the quadrant |X|, |Y| <= h is evaluated -- a quarter of the
    // libm calls -- and every point looks its (|X|, |Y|) up; the argument ORDER is kept as the reference's (X first), never swapped.
    const int q = h + 1;
    std::vector<uint32_t> quad((size_t)q * q);
    for_rows(q, (size_t)q * 4, [&](int ay) {
        for (int ax = 0; ax < q; ++ax) {
            float R = hypotf((float)ax, (float)ay);
            int Rhi = fminf(floorf(R + kernwidth), nxos / 2 - 1);
            int Rlo = fmaxf(ceilf(R - kernwidth), 0);
            if (Rhi < 0) { Rhi = 0; Rlo = 1; }                // nxos < 2: empty band
            quad[(size_t)ay * q + ax] = (uint32_t)Rlo | ((uint32_t)Rhi << 16);
        }
    });
    for (int yy = 0; yy < nxos; ++yy) {
        const int ay = yy < h ? h - yy : yy - h;
        uint32_t *row = band + (size_t)yy * nxos;
        const uint32_t *qr = quad.data() + (size_t)ay * q;
        for (int xx = 0; xx < nxos; ++xx) row[xx] = qr[xx < h ? h - xx : xx - h];
    }
}

// grid_scatter_kernel tests the band as (u - W)^2 <= X^2 + Y^2 <= (u + W)^2 (u - W clamped at 0) instead of reading this table.  For integer u
// that is the same set as Rlo <= u <= Rhi in exact arithmetic; the table is built in fp32 (hypotf, R +- W rounded), so the two are
// compared here for every point and every radius next to its band's ends, in the kernel's own fp32 expressions (all exact: 4 W is an
// integer, the squares stay below 2^24).  False: the plan keeps the arc kernel.
bool scatter_band_is_analytic(int nxos, float kernwidth, const uint32_t *band)
{
    const int h = nxos / 2, rmax = nxos / 2 - 1;
    const float W = kernwidth;
    if (nxos > 2048 || 4.0f * W != floorf(4.0f * W)) return false;
    for (int yy = 0; yy < nxos; ++yy)
        for (int xx = 0; xx < nxos; ++xx) {
            const uint32_t b = band[(size_t)yy * nxos + xx];
            const int lo = (int)(b & 0xffffu), hi = (int)(b >> 16);
            const float X = (float)(xx - h), Y = (float)(yy - h);
            const float n2 = X * X + Y * Y;
            for (int u = std::max(lo - 2, 0); u <= std::min(hi + 2, rmax); ++u) {
                const float um = fmaxf((float)u - W, 0.0f), up = (float)u + W;
                const bool analytic = n2 >= um * um && n2 <= up * up;
                if (analytic != (u >= lo && u <= hi)) return false;
            }
        }
    return true;
}

// src/tron.cu:323-335 (BEATTY_BETA is not defined by the reference Makefile)
float kb_beta(float kernwidth)
{
    return 2.34f * 2.0f * kernwidth;
}

// Fourier transform of the Kaiser-Bessel window, src/tron.cu:351-370
static float kb_hat(float u, float kernwidth)
{
    float J = 2.0f * kernwidth;
    float beta = kb_beta(kernwidth);
    float r = M_PI * J * u;
    float q = r * r - beta * beta;
    float y, z;
    if (q > 0) {
        z = sqrtf(q);
        y = sinf(z) / z;
    } else if (q < 0) {
        z = sqrtf(-q);
        y = sinhf(z) / z;
    } else
        y = 1;
    return y;
}

// 1/w for every pixel, w as deapodkernel computes it (src/tron.cu:393-400, with the fractional
// x coordinate of :395); the kernel's "/= w" is a multiplication by 1.0f/w (float2math.h:23).
void build_deapod_table(int n, float kernwidth, float sigma, float *inv_weight)
{
    // the second factor depends on the column alone (y = id % n - ...): n evaluations instead of n^2; the first keeps the reference's
    // fractional x = id / float(n) - ... (Q7), one evaluation per pixel
    const float scale = 1.f / n / sigma;
    std::vector<float> hy(n);
    for (int col = 0; col < n; ++col) {
        float y = float(col) - (n + 1) / 2;
        hy[col] = kb_hat(y * scale, kernwidth);
    }
    for_rows(n, (size_t)n * 8, [&](int row) {
        for (size_t id = (size_t)row * n; id < (size_t)(row + 1) * n; ++id) {
            float x = id / float(n) - (n + 1) / 2;
            float wgt = kb_hat(x * scale, kernwidth) * hy[id % n];
            inv_weight[id] = 1.0f / (wgt > 0.f ? wgt : 1.f);
        }
    });
}

// The same for a non-square grid (forward plans only; "TODO: implement non-square images", src/tron.cu:945): rows and
// columns each scaled by their own size, the fractional coordinate of :395 kept on the row axis.
void build_deapod_table_rect(int rows, int cols, float kernwidth, float sigma, float *inv_weight)
{
    for (size_t id = 0; id < (size_t)rows * cols; ++id) {
        float x = id / float(cols) - (rows + 1) / 2;
        float y = float(id % cols) - (cols + 1) / 2;
        float wgt = kb_hat(x * (1.f / rows / sigma), kernwidth) * kb_hat(y * (1.f / cols / sigma), kernwidth);
        inv_weight[id] = 1.0f / (wgt > 0.f ? wgt : 1.f);
    }
}

const char *tuning_env(const char *name)
{
    static const bool on = [] { const char *t = getenv("TRON_TUNING"); return t && atoi(t) != 0; }();
    return on ? getenv(name) : nullptr;
}

// TRON_DEBUG (under TRON_TUNING=1): comma-separated debugging / test hooks -- `sync`, `poison`, `cold_fault=<path>`.
bool debug_token(const char *name, std::string *value)
{
    const char *e = tuning_env("TRON_DEBUG");
    if (!e) return false;
    const std::string all(e), key(name);
    size_t pos = 0;
    while (pos <= all.size()) {
        const size_t end = std::min(all.find(',', pos), all.size());
        const std::string tok = all.substr(pos, end - pos);
        if (tok == key || tok.compare(0, key.size() + 1, key + "=") == 0) {
            if (value) *value = tok.size() > key.size() ? tok.substr(key.size() + 1) : std::string();
            return true;
        }
        pos = end + 1;
    }
    return false;
}

// Density compensation constants, src/tron.cu:408-409
void dcf_constants(int nro, int npe1work, float *a, float *b)
{
    *a = (2.f - 2.f / float(npe1work)) / float(nro);
    *b = 1.f / float(npe1work);
}

// Output scale of the gridding kernel, src/tron.cu:532
float grid_scale(int nxos, int npe)
{
    return 1.f / nxos / npe;
}

// Modified Bessel function I0 by its power series, in double.
static double bessel_i0_series(double t)
{
    const double q = t * t / 4.0;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 500; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

// Coefficients (highest power first) of a degree-(nterms-1) polynomial in s = 1-(x/W)^2 that
// approximates G(s) = (0.5/W)*I0(beta*sqrt(s)), the un-normalised Kaiser-Bessel window of
// src/tron.cu:338-349, on [0,1]: Chebyshev interpolation, converted to the monomial basis in
// long double.  Returns the largest error against G on a dense grid, relative to G's peak.
double kb_poly_fit(float kernwidth, float *poly, int nterms)
{
    const double beta = kb_beta(kernwidth);
    const double amp = 0.5 / (double)kernwidth;
    const int N = nterms;
    std::vector<long double> c(N, 0.0L);
    for (int k = 0; k < N; ++k) {
        long double acc = 0.0L;
        for (int j = 0; j < N; ++j) {
            const long double th = M_PIl * (j + 0.5L) / N;
            const double sj = (double)((cosl(th) + 1.0L) / 2.0L);
            acc += (long double)(amp * bessel_i0_series(beta * sqrt(sj))) * cosl(k * th);
        }
        c[k] = acc * 2.0L / N;
    }
    c[0] /= 2.0L;
    // T_k(t) as polynomials in t, then t = 2s - 1
    std::vector<std::vector<long double>> T(N, std::vector<long double>(N, 0.0L));
    T[0][0] = 1.0L;
    if (N > 1) T[1][1] = 1.0L;
    for (int k = 2; k < N; ++k)
        for (int m = 0; m < N; ++m)
            T[k][m] = (m > 0 ? 2.0L * T[k - 1][m - 1] : 0.0L) - T[k - 2][m];
    std::vector<long double> pt(N, 0.0L);          // polynomial in t
    for (int k = 0; k < N; ++k)
        for (int m = 0; m < N; ++m) pt[m] += c[k] * T[k][m];
    std::vector<long double> ps(N, 0.0L);          // polynomial in s
    for (int m = 0; m < N; ++m) {
        // (2s-1)^m = sum_j C(m,j) (2s)^j (-1)^(m-j)
        long double binom = 1.0L;
        for (int j = 0; j <= m; ++j) {
            ps[j] += pt[m] * binom * powl(2.0L, j) * (((m - j) & 1) ? -1.0L : 1.0L);
            binom = binom * (m - j) / (j + 1);
        }
    }
    for (int k = 0; k < N; ++k) poly[k] = (float)ps[N - 1 - k];
    double worst = 0.0;
    for (int i = 0; i <= 4096; ++i) {
        const double sv = i / 4096.0;
        double acc = poly[0];
        for (int k = 1; k < N; ++k) acc = acc * sv + (double)poly[k];
        const double want = amp * bessel_i0_series(beta * sqrt(sv));
        worst = fmax(worst, fabs(acc - want));
    }
    return worst / (amp * bessel_i0_series(beta));           // relative to the window's peak: what a weighted sum feels
}

// Table of the un-normalised Kaiser-Bessel window of src/tron.cu:338-349 for the arc gridding kernel, over the SIGNED distance
// d = k - P0 of a sample coordinate from the first column (row) P0 of a thread's 2x2 block.  Table position t = d s; entry
// i = trunc(t) (towards zero), fraction f = t - i in (-1, 1); the entry holds TWO quadratics c0 + f (c1 + f c2): component A the
// window at d (column P0), component B the window at d - 1 (column P0 + 1) -- one position, one fraction and one address serve
// both columns, and the two evaluations are one packed fma chain.  Planar: coef[0 .. cap) = c0 (A, B), coef[cap .. 2 cap) = c1,
// coef[2 cap .. 3 cap) = c2, two floats per entry (cap = plane stride in entries); entry index = i + bias.
//   Exactness of the support.  The reference's window is zero unless |x| < W, strictly (src/tron.cu:341), and jumps there by
//   0.5 / W; trajectories with spokes along an axis put many samples at EXACTLY |x| = W, or an ulp inside.  s is a power of
//   two, so t = d s is exact and |d| < W <=> |t| < W s in fp32, W s an integer.  Truncation makes the pieces (i - 1, i] for
//   i < 0, (-1, 1) for i = 0 and [i, i + 1) for i > 0: every boundary of either window is then the CLOSED end of the first
//   piece outside (A: t = -W s and t = W s; B: t = (1 - W) s < 0 and t = (1 + W) s), which holds zeros, and the last piece inside
//   ends on the smooth continuation.  This needs (W - 1) s >= 1: W > 1.  (A floor-indexed table gets t = -W s wrong, a table
//   with an offset or a scale that is no power of two rounds t near the upper boundaries; both were measured as 8e-6 on a
//   linear-angle case against 2e-7.)  The piece around t = 0 is two units wide (8 x the others' error, one piece of ~ 2 W s).
// Returns the pieces per grid unit s for a table of `cap` entries (0: this width has no such table: the caller falls back to
// the binned kernel); build_kb_pair_lut returns the entries used, *err = largest error relative to the window's peak
// (1.4e-7 at W = 2 outside the centre piece, 1.1e-6 in it).
int kb_pair_lut_scale(float kernwidth, int cap)
{
    const double W = kernwidth;
    if (!(W > 1.0) || cap < 16) return 0;
    for (int s = 256; s >= 8; s >>= 1) {
        const double ws = W * s;
        if ((2.0 * W + 1.0) * s + 4.0 > cap || fabs(ws - nearbyint(ws)) > 1e-9 * ws || (W - 1.0) * s < 1.0) continue;
        return s;
    }
    return 0;
}

double kb_peak(float kernwidth)      // the un-normalised window at 0: 0.5 I0(beta) / W (src/tron.cu:338-349)
{
    return 0.5 / (double)kernwidth * bessel_i0_series(kb_beta(kernwidth));
}

int build_kb_pair_lut(float kernwidth, int cap, float *coef, float *scale, int *bias, double *err)
{
    const int s = kb_pair_lut_scale(kernwidth, cap);
    if (s == 0) return 0;
    const double W = kernwidth, beta = kb_beta(kernwidth), amp = 0.5 / W;
    const int ws = (int)nearbyint(W * s);
    const int b = ws + 2, imin = -ws - 1, imax = ws + s + 1;
    auto win = [&](double x) {                                       // smooth continuation of the window
        const double r = x / W;
        return amp * bessel_i0_series(beta * sqrt(std::max(0.0, 1.0 - r * r)));
    };
    double worst = 0.0;
    for (int i = 0; i < 3 * 2 * cap; ++i) coef[i] = 0.f;
    for (int i = imin; i <= imax; ++i)
        for (int w = 0; w < 2; ++w) {
            // window w is F(t) = win(t / s - w) on (lo, hi) = ((w - W) s, (w + W) s); piece i lies inside or outside as a whole
            const int lo = w * s - ws, hi = w * s + ws;
            const int p0 = i < 0 ? i - 1 : (i == 0 ? -1 : i), p1 = i <= 0 ? (i == 0 ? 1 : i) : i + 1;      // the piece's ends
            if (p0 < lo || p1 > hi) continue;
            auto F = [&](double f) { return win((i + f) / s - w); };
            double c0 = F(0.0), c1, c2;
            if (i == 0) {
                c2 = 0.5 * (F(1.0) + F(-1.0)) - c0;
                c1 = 0.5 * (F(1.0) - F(-1.0));
            } else {
                const double sg = i > 0 ? 1.0 : -1.0;                // nodes 0, sg / 2, sg
                const double ym = F(0.5 * sg), y1 = F(sg);
                c2 = 2.0 * (y1 - 2.0 * ym + c0);
                c1 = sg * (y1 - c0 - c2);
            }
            for (int t = -7; t < 8; ++t) {
                const double ff = t / 8.0;
                if ((i > 0 && ff < 0) || (i < 0 && ff > 0)) continue;
                worst = std::max(worst, fabs(c0 + ff * (c1 + ff * c2) - F(ff)) / win(0.0));
            }
            const int e = i + b;
            coef[2 * e + w] = (float)c0;
            coef[2 * (cap + e) + w] = (float)c1;
            coef[2 * (2 * cap + e) + w] = (float)c2;
        }
    *scale = (float)s;
    *bias = b;
    if (err) *err = worst;
    return imax + b + 1;
}

// Centre kernel (tron_grid_centre.hip): the angular window of every 2x2 block (col | row << 8 of the origin-centred 32 x 32 square) --
// the arc kernel's rule: a spoke of line angle phi reaches the block's footprint only if its line passes within
// (W + 1/2)(|cos phi| + |sin phi|) of the block centre (tron_grid_arc.hip).  out[4 g] = (lo, hi, all | wrap << 1, expected share of a
// window's spokes): the block's run of a window's angle-sorted list is [lower_bound(lo), upper_bound(hi)), circular when `wrap`, the whole
// list when `all`; traj_centre_windows_kernel (tron_traj_dev.hip) does the searches per window on the device.  Geometry only: no angle enters.
void build_centre_group_windows(const int *groups, int ngroups, float W, float *out)
{
    const float pi = 3.14159265358979f;
    for (int g = 0; g < ngroups; ++g) {
        const float Xc = 2.f * (groups[g] & 255) - 16.f + 0.5f, Yc = 2.f * (groups[g] >> 8) - 16.f + 0.5f;
        const float R = sqrtf(Xc * Xc + Yc * Yc);
        const float sd0 = (W + 0.52f) * 1.41421356f / R;
        const float wcs = fminf(1.41421356f, (fabsf(Xc) + fabsf(Yc)) / R + 1.5f * sd0);
        const float sd = (W + 0.52f) * wcs / R;
        const float D = sd < 0.999f ? asinf(sd) + 2e-3f : 4.0f;
        const bool all = !(D < 0.5f * pi - 1e-3f);
        float T = atan2f(Yc, Xc);
        T -= floorf(T / pi) * pi;
        const float tlo = T - D, thi = T + D;
        const bool wrap = tlo < 0.f || thi >= pi;
        out[4 * g] = tlo < 0.f ? tlo + pi : tlo;
        out[4 * g + 1] = thi >= pi ? thi - pi : thi;
        out[4 * g + 2] = (float)((all ? 1 : 0) | (wrap ? 2 : 0));
        out[4 * g + 3] = all ? 1.0f : fminf(1.0f, 2.0f * D / pi);
    }
}

// Tiles sorted by distance from the k-space centre: radial sampling density falls as 1/r, so
// the central tiles are the expensive ones and are dispatched first.
void build_tile_order(int nxos, int tile, std::vector<int> &order)
{
    const int tpr = (nxos + tile - 1) / tile;
    struct T { float d; int id; };
    std::vector<T> t;
    for (int ty = 0; ty < tpr; ++ty)
        for (int tx = 0; tx < tpr; ++tx) {
            float cx = tx * tile + tile * 0.5f - nxos / 2, cy = ty * tile + tile * 0.5f - nxos / 2;
            t.push_back({cx * cx + cy * cy, ty * tpr + tx});
        }
    for (size_t i = 1; i < t.size(); ++i) {                  // insertion sort keeps ties in raster order
        T v = t[i];
        size_t j = i;
        while (j > 0 && t[j - 1].d > v.d) { t[j] = t[j - 1]; --j; }
        t[j] = v;
    }
    order.resize(t.size());
    for (size_t i = 0; i < t.size(); ++i) order[i] = t[i].id;
}

void build_degrid_groups(int nxos, int tile, int npe, int nro, int target, int end[4])
{
    std::vector<int> order;
    build_tile_order(nxos, tile, order);
    const int tpr = (nxos + tile - 1) / tile;
    int cls = 0;
    for (int c = 0; c < 4; ++c) end[c] = (int)order.size();
    for (size_t pos = 0; pos < order.size(); ++pos) {
        const int id = order[pos];
        const int x0 = (id % tpr) * tile - nxos / 2, y0 = (id / tpr) * tile - nxos / 2;
        double dens = 0.0;                                   // radial sampling density npe / (pi r) per unit area, nro / nxos samples per cell
        for (int y = y0; y < y0 + tile; ++y)
            for (int x = x0; x < x0 + tile; ++x) dens += 1.0 / std::max(0.5, sqrt((x + 0.5) * (x + 0.5) + (y + 0.5) * (y + 0.5)));
        const double est = npe / M_PI * dens * nro / std::max(nxos, 1);
        int need = 0;
        while (need < 4 && est * (1 << need) < target) ++need;
        while (cls < need) end[cls++] = (int)pos;            // classes only grow along the order
    }
}

// Tile list of the binned gridding kernel for SMALL launches: a tile expected to hold more than `target` sample records
// per image is dealt to several workgroups over disjoint spoke ranges (entry = tile | part << 16 | parts << 20 | slot << 24),
// so that its serial chain no longer bounds the launch.  Expected records of a tile = integral of the radial sampling
// density npe / (pi r) over tile + halo.  slots[s] = tile | parts << 20 of split tile s.
void build_split_tile_order(int nxos, int tile, int npe, float W, int target, int max_parts,
                            std::vector<int> &order, std::vector<int> &slots)
{
    std::vector<int> plain;
    build_tile_order(nxos, tile, plain);
    const int tpr = (nxos + tile - 1) / tile;
    const int halo = (int)ceilf(W);
    order.clear();
    slots.clear();
    for (int id : plain) {
        const int x0 = (id % tpr) * tile - nxos / 2, y0 = (id / tpr) * tile - nxos / 2;
        double dens = 0.0;
        for (int y = y0 - halo; y < y0 + tile + halo; ++y)
            for (int x = x0 - halo; x < x0 + tile + halo; ++x)
                dens += 1.0 / std::max(0.5, sqrt((double)x * x + (double)y * y));
        const double est = npe / M_PI * dens;
        int parts = (int)(est / target + 0.5);
        parts = std::max(1, std::min(std::min(parts, max_parts), std::max(1, npe)));
        if (parts > 1 && slots.size() < 255) {
            const int slot = (int)slots.size();
            slots.push_back(id | (parts << 20));
            for (int g = 0; g < parts; ++g) order.push_back(id | (g << 16) | (parts << 20) | (slot << 24));
        } else {
            order.push_back(id);
        }
    }
}

// Centre relief (binned kernel, GridParams::inner_r0): the plain tile order plus `parts` entries of the origin-centred
// inner tile (id = tile count), which come first -- they are ordinary-sized workgroups, and the reduce pass waits for them.
// slots[0] describes the inner tile for grid_reduce_parts_kernel.  Only for grids whose centre is a corner of four tiles.
bool build_centre_relief_order(int nxos, int tile, int npe, float W, int max_parts, int &inner_r0, std::vector<int> &order, std::vector<int> &slots,
                               int target_records)
{
    order.clear();
    slots.clear();
    // an inner sample's footprint (floor(k) - ceil(W) + 1 .. floor(k) + ceil(W)) must stay inside the inner tile
    inner_r0 = tile / 2 - (int)ceilf(W);
    if (nxos < 4 * tile || (nxos / 2) % tile != 0 || inner_r0 < 8 || nxos / 2 - 1 < inner_r0) return false;
    std::vector<int> plain;
    build_tile_order(nxos, tile, plain);
    const int ntiles = (int)plain.size();
    if (ntiles >= 0xffff) return false;
    const int records = npe * (2 * inner_r0 - 1);                 // inner samples per image
    const int parts = std::max(1, std::min(std::min(max_parts, 15), (records + target_records / 2) / target_records));   // about target_records per workgroup
    slots.push_back(ntiles | (parts << 20));
    for (int g = 0; g < parts; ++g) order.push_back(ntiles | (g << 16) | (parts << 20));
    order.insert(order.end(), plain.begin(), plain.end());
    return true;
}

}  // namespace tron

using namespace tron;

extern "C" void tron_config_default(tron_config *cfg)
{
    memset(cfg, 0, sizeof(*cfg));
    cfg->gridos = 2.f;            // src/tron.cu:67
    cfg->kernwidth = 2.f;         // src/tron.cu:68
    cfg->data_undersamp = 1.f;    // src/tron.cu:69
    cfg->blocks = 4096;           // src/tron.cu:59
    cfg->threads = 128;           // src/tron.cu:58
    cfg->kb_mode = TRON_KB_FAST;
    cfg->walsh_patch = 1;         // src/tron.cu:766
    cfg->pin_host = 1;            // the reference pins its output (cudaMallocHost, src/tron.cu:967)
}

// main()'s dimension logic.  Adjoint: src/tron.cu:905-935; forward: :936-961; the int <- float
// conversions truncate exactly where the reference's implicit conversions do.
extern "C" int tron_derive_dims(const tron_config *cfg, const uint64_t in_dims[5], tron_dims *d)
{
    if (!cfg || !in_dims || !d) return tron::fail(TRON_ERR_INVALID, "tron_derive_dims: null argument");
    memset(d, 0, sizeof(*d));
    for (int i = 0; i < 5; ++i)
        if (in_dims[i] == 0 || in_dims[i] > 0x7fffffffull)
            return tron::fail(TRON_ERR_INVALID, "tron_derive_dims: input dimension %d = %llu out of range", i, (unsigned long long)in_dims[i]);
    int prof_slide = cfg->prof_slide;
    d->nc = (int)in_dims[0];
    d->nt = (int)in_dims[1];
    d->out_dims[0] = 1;                                               // src/tron.cu:899
    if (cfg->adjoint) {
        d->nro = (int)in_dims[2];
        d->npe1 = (int)in_dims[3];
        d->npe2 = (int)in_dims[4];
        d->nx = d->nro / 2;
        d->ny = d->nro / 2;
        d->nxos = d->nx * cfg->gridos;
        d->nyos = d->ny * cfg->gridos;
        if (d->npe1 <= d->nro * cfg->data_undersamp)                 // src/tron.cu:916
            d->npe1work = d->npe1;
        else
            d->npe1work = d->nro * cfg->data_undersamp;
        if (prof_slide == 0) prof_slide = d->npe1work;                // src/tron.cu:920
        if (d->npe1work <= 0 || prof_slide <= 0)
            return tron::fail(TRON_ERR_INVALID, "tron_derive_dims: no spokes per image (npe1work=%d, prof_slide=%d)", d->npe1work, prof_slide);
        if (cfg->koosh) {
            d->nz = d->nro / 2;
            d->nzos = d->nz * cfg->gridos;
        } else {
            d->nz = 1 + (d->npe1 - d->npe1work) / prof_slide;         // src/tron.cu:926
            d->nzos = 1;
        }
        d->out_dims[1] = d->nt;
        d->out_dims[2] = d->nx;
        d->out_dims[3] = d->ny;
        d->out_dims[4] = d->nz;
    } else {
        d->nx = (int)in_dims[2];
        d->ny = (int)in_dims[3];
        d->nz = (int)in_dims[4];
        d->nxos = d->nx * cfg->gridos;
        d->nyos = d->ny * cfg->gridos;
        d->nro = cfg->gridos * d->nx;                                 // src/tron.cu:945
        d->npe1work = cfg->data_undersamp * d->nro;
        d->npe1 = d->npe1work;
        if (cfg->koosh) {
            d->npe2 = d->nz;
            d->nzos = d->nz;
        } else {
            d->npe2 = 1;
            d->nzos = 1;
        }
        d->out_dims[1] = d->nt;
        d->out_dims[2] = d->nro;
        d->out_dims[3] = d->npe1;
        d->out_dims[4] = d->npe2;
    }
    d->prof_slide = prof_slide;
    // h_outdatasize (src/tron.cu:934,960) and the input element count, with checked products: five dims of up to
    // 2^31-1 each wrap 64 bits, and a wrapped count would pass every later size check (crafted .ra header)
    {
        const uint64_t lead = cfg->adjoint ? 1 : (uint64_t)d->nc;
        uint64_t ob = sizeof(tron_float2), ie = 1;
        const uint64_t of[5] = {lead, (uint64_t)d->out_dims[1], (uint64_t)d->out_dims[2], (uint64_t)d->out_dims[3], (uint64_t)d->out_dims[4]};
        bool bad = false;
        for (int i = 0; i < 5; ++i) {
            bad = bad || __builtin_mul_overflow(ob, of[i], &ob);
            bad = bad || __builtin_mul_overflow(ie, in_dims[i], &ie);
        }
        if (bad || ie > (UINT64_MAX >> 4) || ob > (UINT64_MAX >> 1))
            return tron::fail(TRON_ERR_INVALID, "tron_derive_dims: dimensions overflow the addressable size");
        d->out_bytes = ob;
        d->in_elems = ie;
    }
    if (!(d->nc % 2 == 0 || d->nc == 1))                              // assert at src/tron.cu:963
        return tron::fail(TRON_ERR_INVALID, "only one or an even number of coils is supported (nc=%d), as in the reference", d->nc);
    if (d->nx <= 0 || d->nxos <= 0 || d->nro <= 0 || d->npe1work <= 0 || d->nz <= 0)
        return tron::fail(TRON_ERR_INVALID, "degenerate dimensions (nx=%d nxos=%d nro=%d npe=%d nz=%d)", d->nx, d->nxos, d->nro, d->npe1work, d->nz);
    return TRON_OK;
}

extern "C" int tron_host_trig_table(const tron_config *cfg, const tron_dims *dims, float *cos_sin, size_t n)
{
    if (!cfg || !dims || !cos_sin) return tron::fail(TRON_ERR_INVALID, "tron_host_trig_table: null argument");
    if (n > trig_table_size(*cfg, *dims)) return tron::fail(TRON_ERR_INVALID, "tron_host_trig_table: table has only %zu entries", trig_table_size(*cfg, *dims));
    build_trig_table_mt(*cfg, *dims, cos_sin, n, 8);          // (the plan's own builder: tables of more than 8 192 entries on several threads)
    return TRON_OK;
}

// CPUs of the NUMA node a PCI function sits on, from a sysfs tree: <sysroot>/bus/pci/devices/<bus id>/numa_node names the
// node, <sysroot>/devices/system/node/node<N>/cpulist its CPUs ("0-31,64-95").  Returns how many CPUs were written, 0 when
// the node is unknown (-1 in numa_node: single-node hosts, VMs), -1 on malformed input.  (tron_recon_radial2d_multi pins each
// per-GPU worker thread with it; a separate entry point so that the parsing is testable against a fake tree.)
extern "C" int tron_host_numa_cpulist(const char *sysroot, const char *pci_bus_id, int *cpus, int max_cpus)
{
    if (!sysroot || !pci_bus_id || !cpus || max_cpus < 1) return -1;
    std::string id(pci_bus_id);
    for (char &ch : id) ch = (char)tolower((unsigned char)ch);     // sysfs spells the bus id in lower case
    auto slurp = [](const std::string &path, std::string &out) {
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return false;
        char buf[4096];
        const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
        fclose(f);
        buf[n] = 0;
        out = buf;
        return true;
    };
    std::string text;
    if (!slurp(std::string(sysroot) + "/bus/pci/devices/" + id + "/numa_node", text)) return 0;
    const int node = atoi(text.c_str());
    if (node < 0) return 0;
    if (!slurp(std::string(sysroot) + "/devices/system/node/node" + std::to_string(node) + "/cpulist", text)) return 0;
    int n = 0;
    const char *q = text.c_str();
    while (*q) {
        while (*q == ',' || *q == ' ' || *q == '\n') ++q;
        if (!*q) break;
        if (*q < '0' || *q > '9') return -1;
        char *end = nullptr;
        long a = strtol(q, &end, 10), b = a;
        q = end;
        if (*q == '-') { b = strtol(q + 1, &end, 10); if (end == q + 1) return -1; q = end; }
        if (b < a) return -1;
        for (long c = a; c <= b && n < max_cpus; ++c) cpus[n++] = (int)c;
    }
    return n;
}

extern "C" int tron_host_band_table(int nxos, float kernwidth, uint32_t *band)
{
    if (nxos < 2 || nxos > 16384 || !band) return tron::fail(TRON_ERR_INVALID, "tron_host_band_table: bad argument");
    build_band_table(nxos, kernwidth, band);
    return TRON_OK;
}

extern "C" int tron_host_deapod_table(int n, float kernwidth, float sigma, float *inv_weight)
{
    if (n < 1 || !inv_weight) return tron::fail(TRON_ERR_INVALID, "tron_host_deapod_table: bad argument");
    build_deapod_table(n, kernwidth, sigma, inv_weight);
    return TRON_OK;
}
