/*
 * irt_nufft.cpp -- C++/OpenMP restatement of the CPU NUFFT the reference measures itself against: Fessler's IRT
 * `nufft_init('minmax:kb')` + `nufft_adj`, bundled with the reference as MATLAB under contrib/irt.
 *
 * TEST INFRASTRUCTURE / REPORTED CPU BASELINE ONLY (bench.py's irt_baseline, tests/test_irt.py).  Nothing under
 * tron_amd/ links, loads or calls it.
 *
 * Follows, for even J (the reference uses Jd = [4 4], Kd = 2 Nd, n_shift = Nd/2, src/RUNME4_others_grid_slcmt.m:112-130):
 *   contrib/irt/nufft_init.m:153-157      'minmax:kb' -> nufft_alpha_kb_fit per dimension
 *   contrib/irt/nufft_alpha_kb_fit.m:1-33 least-squares fit of L+1 cosine coefficients to the KB scaling factors
 *   contrib/irt/kaiser_bessel_ft.m:95-98  FT of the Kaiser-Bessel window (m = 0, d = 1)
 *   contrib/irt/private/kaiser,m=0.mat    'best' alpha/J (J = 4: 2.31)
 *   contrib/irt/nufft_scale.m:30-49       scaling factors sn
 *   contrib/irt/private/nufft_T.m:60-94, nufft_r.m:31-47, nufft_offset.m:14-19, nufft_diric.m (sinc form)
 *   contrib/irt/nufft_init.m:221-278      interpolation coefficients, linear phase, the sparse matrix (16 nnz per sample)
 *   contrib/irt/nufft_adj.m:50-77         p' * X -> prod(Kd) * ifft2 -> crop [1:N1, 1:N2] -> .* conj(sn)
 * Double precision throughout, as MATLAB runs it (RUNME4:73 casts the data to double).  MATLAB built-ins (besseli,
 * `\`, inv, sparse, ifftn) are replaced by a series I0, a QR least squares, Gauss-Jordan, a scatter loop and a radix-2 FFT.
 * Parity: unpinned by the reference (no numbers for IRT in its tree); pinned here against the numpy restatement
 * oracle/irt_nufft.py, which tests/test_irt.py checks against a brute-force DTFT.
 *
 * As in RUNME4:112-130 the operator is re-initialised for every slice (the trajectory rotates with the sliding window)
 * and the density weights are multiplied in by the caller.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <complex>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef std::complex<double> cd;

namespace {

const double kBestAlphaOverJ[17] = {0, 0, 2.5, 2.27, 2.31, 2.34, 2.32, 2.32, 2.35, 2.34, 2.34, 2.35, 2.34, 2.35, 2.35, 2.35, 2.33};

double bessel_i0(double t)
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

double kaiser_bessel_ft(double u, int J, double alpha)      // kaiser_bessel_ft.m:95-98
{
    const double q = (M_PI * J * u) * (M_PI * J * u) - alpha * alpha;
    const double z = sqrt(fabs(q));
    const double s = q > 0 ? sin(z) / z : (q < 0 ? sinh(z) / z : 1.0);
    return J / bessel_i0(alpha) * s;
}

double sinc(double x) { return x == 0.0 ? 1.0 : sin(M_PI * x) / (M_PI * x); }
double diric(double k, int N, int K) { return sinc(k / ((double)K / N)); }        // nufft_diric.m, sinc form

// least squares by Householder QR (columns = cosines: well conditioned)
void lstsq(std::vector<double> A, int m, int n, std::vector<double> b, std::vector<double> &x)
{
    for (int k = 0; k < n; ++k) {
        double nrm = 0;
        for (int i = k; i < m; ++i) nrm += A[i * n + k] * A[i * n + k];
        nrm = sqrt(nrm);
        const double alpha = A[k * n + k] > 0 ? -nrm : nrm;
        std::vector<double> v(m, 0.0);
        for (int i = k; i < m; ++i) v[i] = A[i * n + k];
        v[k] -= alpha;
        double vv = 0;
        for (int i = k; i < m; ++i) vv += v[i] * v[i];
        if (vv == 0) continue;
        for (int j = k; j < n; ++j) {
            double d = 0;
            for (int i = k; i < m; ++i) d += v[i] * A[i * n + j];
            d = 2 * d / vv;
            for (int i = k; i < m; ++i) A[i * n + j] -= d * v[i];
        }
        double d = 0;
        for (int i = k; i < m; ++i) d += v[i] * b[i];
        d = 2 * d / vv;
        for (int i = k; i < m; ++i) b[i] -= d * v[i];
    }
    x.assign(n, 0.0);
    for (int k = n - 1; k >= 0; --k) {
        double s = b[k];
        for (int j = k + 1; j < n; ++j) s -= A[k * n + j] * x[j];
        x[k] = s / A[k * n + k];
    }
}

struct Dim {
    int N, J, K;
    std::vector<double> alpha;      // alpha[0..L]
    std::vector<cd> sn;             // N
    double Tinv[16 * 16];           // J x J
};

void init_dim(Dim &d, int N, int J, int K)
{
    d.N = N; d.J = J; d.K = K;
    const int L = N > 40 ? 13 : (int)ceil(N / 3.0);                       // nufft_alpha_kb_fit.m
    std::vector<double> X((size_t)N * (L + 1)), sk(N), coef;
    const double gam = 2 * M_PI / K;
    for (int i = 0; i < N; ++i) {
        const double n = i - (N - 1) / 2.0;
        sk[i] = 1.0 / kaiser_bessel_ft(n / K, J, kBestAlphaOverJ[J] * J);
        for (int l = 0; l <= L; ++l) X[(size_t)i * (L + 1) + l] = cos(gam * n * l);
    }
    lstsq(X, N, L + 1, sk, coef);
    d.alpha.assign(L + 1, 0.0);
    d.alpha[0] = coef[0];
    for (int l = 1; l <= L; ++l) d.alpha[l] = coef[l] / 2;
    d.sn.assign(N, cd(0, 0));                                             // nufft_scale.m:30-49, beta = 1
    for (int i = 0; i < N; ++i) {
        const double n = i - (N - 1) / 2.0;
        cd s(0, 0);
        for (int l = -L; l <= L; ++l) s += d.alpha[abs(l)] * std::exp(cd(0, gam * n * l));
        d.sn[i] = s;
    }
    double c[16 * 16];                                                    // nufft_T.m:60-94
    for (int a = 0; a < J; ++a)
        for (int b = 0; b < J; ++b) {
            double s = 0;
            for (int l1 = -L; l1 <= L; ++l1)
                for (int l2 = -L; l2 <= L; ++l2)
                    s += d.alpha[abs(l1)] * d.alpha[abs(l2)] * diric((b + 1) - (a + 1) + (l1 - l2), N, K);
            c[a * J + b] = s;
        }
    // Gauss-Jordan inverse
    double aug[16][32];
    for (int i = 0; i < J; ++i)
        for (int j = 0; j < J; ++j) { aug[i][j] = c[i * J + j]; aug[i][J + j] = i == j ? 1.0 : 0.0; }
    for (int col = 0; col < J; ++col) {
        int piv = col;
        for (int r = col + 1; r < J; ++r) if (fabs(aug[r][col]) > fabs(aug[piv][col])) piv = r;
        for (int j = 0; j < 2 * J; ++j) std::swap(aug[col][j], aug[piv][j]);
        const double inv = 1.0 / aug[col][col];
        for (int j = 0; j < 2 * J; ++j) aug[col][j] *= inv;
        for (int r = 0; r < J; ++r)
            if (r != col) {
                const double f = aug[r][col];
                for (int j = 0; j < 2 * J; ++j) aug[r][j] -= f * aug[col][j];
            }
    }
    for (int i = 0; i < J; ++i)
        for (int j = 0; j < J; ++j) d.Tinv[i * J + j] = aug[i][J + j];
}

// interpolation coefficients of one sample along one dimension: u[j] (J values) and the 0-based grid indices k[j]
inline void sample_dim(const Dim &d, double om, cd *u, int *k)
{
    const int J = d.J, K = d.K, N = d.N, L = (int)d.alpha.size() - 1;
    const double gam = 2 * M_PI / K;
    const double koff = floor(om / gam) - J / 2.0;                        // nufft_offset.m (even J)
    const double dk = om / gam - koff;
    double r[16];
    for (int j = 0; j < J; ++j) {                                         // nufft_r.m:31-47
        const double arg = -(j + 1) + dk;
        double s = 0;
        for (int l = -L; l <= L; ++l) s += d.alpha[abs(l)] * diric(arg + l, N, K);
        r[j] = s;
    }
    for (int j = 0; j < J; ++j) {
        double c = 0;
        for (int i = 0; i < J; ++i) c += d.Tinv[j * J + i] * r[i];
        const double arg = -(j + 1) + dk;
        u[j] = std::exp(cd(0, gam * (N - 1) / 2.0 * arg)) * c;            // nufft_init.m:237-241
        long kk = (long)((j + 1) + koff);
        kk %= K; if (kk < 0) kk += K;
        k[j] = (int)kk;
    }
}

void fft1(cd *x, int n, int sign)          // radix-2, in place, unnormalised
{
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(x[i], x[j]);
    }
    for (int len = 2; len <= n; len <<= 1) {
        const double ang = sign * 2 * M_PI / len;
        const cd wl(cos(ang), sin(ang));
        for (int i = 0; i < n; i += len) {
            cd w(1, 0);
            for (int j = 0; j < len / 2; ++j) {
                const cd a = x[i + j], b = x[i + j + len / 2] * w;
                x[i + j] = a + b;
                x[i + j + len / 2] = a - b;
                w *= wl;
            }
        }
    }
}

struct Operator {           // st = nufft_init(om, [N N], [J J], [K K], [N/2 N/2])
    Dim d[2];
    int M;
    std::vector<int> kk;    // [M][J*J] linear index k1 + K*k2
    std::vector<cd> uu;     // [M][J*J]  = conj(u1 (x) u2) * exp(i om . n_shift)   (nufft_init.m:260-278)
};

void nufft_init(Operator &st, const double *om, int M, int N, int J, int K, int threads)
{
    init_dim(st.d[0], N, J, K);
    st.d[1] = st.d[0];
    st.M = M;
    st.kk.resize((size_t)M * J * J);
    st.uu.resize((size_t)M * J * J);
    const double nshift = N / 2;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int m = 0; m < M; ++m) {
        cd u1[16], u2[16];
        int k1[16], k2[16];
        sample_dim(st.d[0], om[2 * (size_t)m], u1, k1);
        sample_dim(st.d[1], om[2 * (size_t)m + 1], u2, k2);
        const cd phase = std::exp(cd(0, (om[2 * (size_t)m] + om[2 * (size_t)m + 1]) * nshift));
        for (int b = 0; b < J; ++b)
            for (int a = 0; a < J; ++a) {
                st.kk[(size_t)m * J * J + b * J + a] = k1[a] + K * k2[b];
                st.uu[(size_t)m * J * J + b * J + a] = std::conj(u1[a] * u2[b]) * phase;
            }
    }
}

// x[n1 + N*n2] = nufft_adj(X, st)   (nufft_adj.m:50-77)
void nufft_adj(const Operator &st, const cd *X, cd *x, std::vector<cd> &Xk)
{
    const int N = st.d[0].N, K = st.d[0].K, JJ = st.d[0].J * st.d[0].J;
    Xk.assign((size_t)K * K, cd(0, 0));
    for (int m = 0; m < st.M; ++m)                                        // p' * X
        for (int j = 0; j < JJ; ++j) Xk[st.kk[(size_t)m * JJ + j]] += std::conj(st.uu[(size_t)m * JJ + j]) * X[m];
    std::vector<cd> col(K);
    for (int k2 = 0; k2 < K; ++k2) fft1(&Xk[(size_t)k2 * K], K, +1);      // prod(Kd) * ifftn = unnormalised inverse DFT
    for (int k1 = 0; k1 < N; ++k1) {                                      // only the columns that survive the crop
        for (int k2 = 0; k2 < K; ++k2) col[k2] = Xk[(size_t)k2 * K + k1];
        fft1(col.data(), K, +1);
        for (int n2 = 0; n2 < N; ++n2) x[k1 + (size_t)N * n2] = col[n2] * std::conj(st.d[0].sn[k1] * st.d[1].sn[n2]);
    }
}

}  // namespace

extern "C" {

/* One slice: st = nufft_init(om, ...); for each coil x_c = nufft_adj(dcf .* X_c, st).  om: [M][2] radians; X: [nc][M]
   complex (re, im); dcf: [M] or NULL; img: [nc][N*N] complex out (n1 fastest).  Returns 0. */
int irt_adjoint(int N, int J, int K, int M, const double *om, const double *X, const double *dcf, int nc, double *img, int threads)
{
    if (J > 16 || (J & 1) || (K & (K - 1))) return -1;
    Operator st;
    nufft_init(st, om, M, N, J, K, threads > 0 ? threads : 1);
#pragma omp parallel for schedule(dynamic) num_threads(threads > 0 ? threads : 1)
    for (int c = 0; c < nc; ++c) {
        std::vector<cd> Xc(M), Xk;
        for (int m = 0; m < M; ++m) Xc[m] = cd(X[2 * ((size_t)c * M + m)], X[2 * ((size_t)c * M + m) + 1]) * (dcf ? dcf[m] : 1.0);
        nufft_adj(st, Xc.data(), reinterpret_cast<cd *>(img) + (size_t)c * N * N, Xk);
    }
    return 0;
}

/* The comparator as the reference's scripts run it (RUNME4:112-130), timed: `nslices` slices of nro x npe golden-angle
   spokes, nc coils of synthetic data, one nufft_init PER SLICE, density weights |r|, root-sum-of-squares of the coil
   images.  Slices are dealt to `threads` OpenMP threads (each slice: init + nc adjoints, serial).  Returns wall seconds;
   t_init / t_adj receive the summed per-thread seconds of the two parts; checksum guards against dead-code removal. */
double irt_bench_golden(int N, int nro, int npe, int nc, int nslices, int threads, double *t_init, double *t_adj, double *checksum)
{
    const int J = 4, K = 2 * N, M = nro * npe;
    double ti = 0, ta = 0, cs = 0;
#ifdef _OPENMP
    const double w0 = omp_get_wtime();
#else
    const double w0 = 0;
#endif
#pragma omp parallel for schedule(dynamic) num_threads(threads) reduction(+ : ti, ta, cs)
    for (int z = 0; z < nslices; ++z) {
        std::vector<double> om(2 * (size_t)M), dcf(M);
        const float PHI = 1.9416089796736116f;                            /* src/tron.cu:90 */
        for (int pe = 0; pe < npe; ++pe) {
            const double th = fmod((double)(PHI * (float)(pe + z * npe)), 2 * M_PI);     /* src/tron.cu:509 */
            for (int ro = 0; ro < nro; ++ro) {
                const double r = (double)ro / nro - 0.5;
                om[2 * ((size_t)pe * nro + ro)] = 2 * M_PI * r * cos(th);
                om[2 * ((size_t)pe * nro + ro) + 1] = 2 * M_PI * r * sin(th);
                dcf[(size_t)pe * nro + ro] = fabs(r);
            }
        }
#ifdef _OPENMP
        double t0 = omp_get_wtime();
#else
        double t0 = 0;
#endif
        Operator st;
        nufft_init(st, om.data(), M, N, J, K, 1);
#ifdef _OPENMP
        double t1 = omp_get_wtime();
#else
        double t1 = 0;
#endif
        std::vector<cd> X(M), x((size_t)N * N), Xk;
        std::vector<double> sos((size_t)N * N, 0.0);
        unsigned long long s = 0x54524F4Eull + 7919ull * z;
        for (int c = 0; c < nc; ++c) {
            for (int m = 0; m < M; ++m) {
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                const double re = (double)((s >> 11) & 0xfffff) / 524288.0 - 1.0;
                s = s * 6364136223846793005ull + 1442695040888963407ull;
                const double im = (double)((s >> 11) & 0xfffff) / 524288.0 - 1.0;
                X[m] = cd(re, im) * dcf[m];
            }
            nufft_adj(st, X.data(), x.data(), Xk);
            for (size_t i = 0; i < (size_t)N * N; ++i) sos[i] += std::norm(x[i]);
        }
        double acc = 0;
        for (size_t i = 0; i < (size_t)N * N; ++i) acc += sqrt(sos[i]);
#ifdef _OPENMP
        double t2 = omp_get_wtime();
#else
        double t2 = 0;
#endif
        ti += t1 - t0; ta += t2 - t1; cs += acc;
    }
#ifdef _OPENMP
    const double w1 = omp_get_wtime();
#else
    const double w1 = 0;
#endif
    if (t_init) *t_init = ti;
    if (t_adj) *t_adj = ta;
    if (checksum) *checksum = cs;
    return w1 - w0;
}

}  // extern "C"
