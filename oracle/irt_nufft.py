"""CPU restatement (numpy, double precision) of the NUFFT the reference compares itself against:
Fessler's IRT `nufft_init('minmax:kb')` / `nufft` / `nufft_adj`, bundled with the reference as MATLAB
under contrib/irt.  TEST INFRASTRUCTURE and CPU comparator only (BASELINE.json config 1);
nothing under tron_amd/ uses it.

Follows, for even J (the reference uses J = 4, K = 2N, n_shift = N/2; src/RUNME2_others_degrid_phantom.m:60-63):
  contrib/irt/nufft_init.m:153-157   'minmax:kb' -> nufft_alpha_kb_fit per dimension
  contrib/irt/nufft_alpha_kb_fit.m   LS fit of L+1 cosine coefficients to the Kaiser-Bessel FT scaling factors
  contrib/irt/nufft1_error.m:110-152 sn_kaiser = 1 / kaiser_bessel_ft(n/K, J, alpha_best, 0, 1)
  contrib/irt/kaiser_bessel_ft.m:95-98  FT of the KB window (m = 0, d = 1)
  contrib/irt/private/kaiser,m=0.mat 'best' alpha/J table (J = 2..16): 2.5 2.27 2.31 2.34 2.32 2.32 2.35 ...
  contrib/irt/nufft_scale.m:30-49    scaling factors sn
  contrib/irt/private/nufft_T.m:60-94, nufft_r.m:31-47, nufft_offset.m:14-19, nufft_diric.m (sinc form)
  contrib/irt/nufft_init.m:221-278   interpolation coefficients, linear phase, sparse matrix
  contrib/irt/nufft.m:142-166        forward:  x.*sn -> zero-padded fft2 -> p * Xk
  contrib/irt/nufft_adj.m:50-77      adjoint:  p' * X -> prod(Kd)*ifft2 -> crop -> .*conj(sn)
MATLAB built-ins (besseli, besselj, `\`, inv, sparse, fftn) are replaced by scipy/numpy equivalents; the
reference pins no numbers for this code ("parity unpinned"), so the restatement is checked against the
NUFFT's definition, a brute-force DTFT (tests/test_irt.py).
"""
import numpy as np
from scipy import sparse, special

# contrib/irt/private/kaiser,m=0.mat: abest.zn for Jlist = 2..16
KB_BEST_ALPHA_OVER_J = {2: 2.5, 3: 2.27, 4: 2.31, 5: 2.34, 6: 2.32, 7: 2.32, 8: 2.35, 9: 2.34, 10: 2.34,
                        11: 2.35, 12: 2.34, 13: 2.35, 14: 2.35, 15: 2.35, 16: 2.33}


def kaiser_bessel_ft(u, J, alpha):
    """kaiser_bessel_ft.m:95-98 with kb_m = 0, d = 1:  sqrt(2 pi) (J/2) / I0(alpha) * J_{1/2}(z) / sqrt(z),
    z = sqrt((pi J u)^2 - alpha^2)  (J_{1/2}(z)/sqrt(z) = sqrt(2/pi) sin(z)/z, sinh for imaginary z)."""
    q = (np.pi * J * np.asarray(u, float)) ** 2 - alpha ** 2
    z = np.sqrt(np.abs(q))
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(q > 0, np.sin(z) / z, np.where(q < 0, np.sinh(z) / z, 1.0))
    return J / special.i0(alpha) * s


def alpha_kb_fit(N, J, K):
    """nufft_alpha_kb_fit.m:1-33 (beta = 1)."""
    L = 13 if N > 40 else int(np.ceil(N / 3))
    n = np.arange(N) - (N - 1) / 2
    sn_kaiser = 1.0 / kaiser_bessel_ft(n / K, J, KB_BEST_ALPHA_OVER_J[J] * J)
    gam = 2 * np.pi / K
    X = np.cos(gam * np.outer(n, np.arange(L + 1)))
    coef = np.linalg.lstsq(X, sn_kaiser, rcond=None)[0]
    alphas = np.concatenate([[coef[0]], coef[1:] / 2])
    return alphas, 1.0


def nufft_scale(N, K, alpha, beta):
    """nufft_scale.m:30-49 (real alpha)."""
    n = np.arange(N) - (N - 1) / 2
    L = len(alpha) - 1
    sn = np.zeros(N, complex)
    for l in range(-L, L + 1):
        sn += alpha[abs(l)] * np.exp(1j * (2 * np.pi / K) * n * beta * l)
    return sn


def _diric(k, N, K):
    return np.sinc(np.asarray(k, float) / (K / N))      # nufft_diric.m, sinc form (use_true_diric = 0)


def nufft_T(N, J, K, alpha, beta):
    """private/nufft_T.m:60-94."""
    L = len(alpha) - 1
    j1, j2 = np.meshgrid(np.arange(1, J + 1), np.arange(1, J + 1), indexing="ij")
    cssc = np.zeros((J, J))
    for l1 in range(-L, L + 1):
        for l2 in range(-L, L + 1):
            cssc += alpha[abs(l1)] * alpha[abs(l2)] * _diric(j2 - j1 + beta * (l1 - l2), N, K)
    return np.linalg.inv(cssc)


def nufft_offset(om, J, K):
    """private/nufft_offset.m:14-19."""
    gam = 2 * np.pi / K
    if J % 2:
        return np.round(om / gam) - (J + 1) / 2
    return np.floor(om / gam) - J / 2


def nufft_r(om, N, J, K, alpha, beta):
    """private/nufft_r.m:31-47."""
    gam = 2 * np.pi / K
    dk = om / gam - nufft_offset(om, J, K)
    arg = -np.arange(1, J + 1)[:, None] + dk[None, :]
    L = len(alpha) - 1
    rr = np.zeros_like(arg)
    for l in range(-L, L + 1):
        rr += alpha[abs(l)] * _diric(arg + l * beta, N, K)
    return rr, arg


class Nufft:
    """st = nufft_init(om, Nd, Jd, Kd, n_shift) with the default 'minmax:kb' interpolator (nufft_init.m)."""

    def __init__(self, om, Nd, Jd, Kd, n_shift):
        om = np.asarray(om, float)
        self.Nd, self.Jd, self.Kd = tuple(Nd), tuple(Jd), tuple(Kd)
        M = om.shape[0]
        sn = np.ones(1, complex)
        ud, kd = [], []
        for d in range(2):
            N, J, K = Nd[d], Jd[d], Kd[d]
            alpha, beta = alpha_kb_fit(N, J, K)
            sn = np.outer(sn.ravel(), nufft_scale(N, K, alpha, beta)).ravel() if d else nufft_scale(N, K, alpha, beta)
            T = nufft_T(N, J, K, alpha, beta)
            r, arg = nufft_r(om[:, d], N, J, K, alpha, beta)
            c = T @ r
            phase = np.exp(1j * (2 * np.pi / K) * (N - 1) / 2 * arg)          # nufft_init.m:237-241
            ud.append(phase * c)
            koff = nufft_offset(om[:, d], J, K)
            kd.append(np.mod(np.arange(1, J + 1)[:, None] + koff[None, :], K).astype(np.int64))   # 0-based
        self.sn = sn.reshape(Nd[0], Nd[1])                                     # sn(n1, n2) = sn1(n1) * sn2(n2)
        J1, J2 = Jd
        kk = (kd[0][:, None, :] + Kd[0] * kd[1][None, :, :]).reshape(J1 * J2, M)   # dim 1 fastest, as MATLAB
        uu = (ud[0][:, None, :] * ud[1][None, :, :]).reshape(J1 * J2, M)
        phase = np.exp(1j * (om @ np.asarray(n_shift, float)))                 # nufft_init.m:272
        uu = np.conj(uu) * phase[None, :]
        mm = np.broadcast_to(np.arange(M)[None, :], kk.shape)
        self.p = sparse.csr_matrix((uu.ravel(), (mm.ravel(), kk.ravel())), shape=(M, Kd[0] * Kd[1]))

    def forward(self, x):
        """nufft.m:142-166.  x[n1, n2] -> X[M]."""
        Xk = np.fft.fft2(np.asarray(x, complex) * self.sn, s=self.Kd)          # zero padding at the end
        return self.p @ Xk.reshape(-1, order="F")

    def adjoint(self, X):
        """nufft_adj.m:50-77.  X[M] -> x[n1, n2]."""
        Xk = (self.p.conj().T @ np.asarray(X, complex)).reshape(self.Kd, order="F")
        x = np.prod(self.Kd) * np.fft.ifft2(Xk)
        return x[: self.Nd[0], : self.Nd[1]] * np.conj(self.sn)


def radial_trajectory(nro, npe):
    """src/RUNME2_others_degrid_phantom.m:29-36: linear radial, theta = (pe-1) pi / npe, r = (0:nro-1)/nro - 1/2.
    Returns om [nro*npe, 2] (= 2 pi traj, readout fastest) with column 0 <-> first image index."""
    r = np.arange(nro) / nro - 0.5
    th = np.arange(npe) * np.pi / npe
    kx = r[:, None] * np.cos(th)[None, :]
    ky = r[:, None] * np.sin(th)[None, :]
    return 2 * np.pi * np.stack([kx.reshape(-1, order="F"), ky.reshape(-1, order="F")], axis=1)


def shepp_logan(n):
    """Modified Shepp-Logan phantom (Toft's ellipse table, as MATLAB's phantom()); the reference's
    data/shepplogan.ra is a git-LFS pointer, so a stand-in of the same kind is generated here."""
    e = [(1, .69, .92, 0, 0, 0), (-.8, .6624, .8740, 0, -.0184, 0), (-.2, .1100, .3100, .22, 0, -18),
         (-.2, .1600, .4100, -.22, 0, 18), (.1, .2100, .2500, 0, .35, 0), (.1, .0460, .0460, 0, .1, 0),
         (.1, .0460, .0460, 0, -.1, 0), (.1, .0460, .0230, -.08, -.605, 0), (.1, .0230, .0230, 0, -.606, 0),
         (.1, .0230, .0460, .06, -.605, 0)]
    ax = (np.arange(n) - (n - 1) / 2) / ((n - 1) / 2)
    x, y = np.meshgrid(ax, -ax)
    img = np.zeros((n, n))
    for A, a, b, x0, y0, phi in e:
        ph = np.deg2rad(phi)
        xr = (x - x0) * np.cos(ph) + (y - y0) * np.sin(ph)
        yr = -(x - x0) * np.sin(ph) + (y - y0) * np.cos(ph)
        img[(xr / a) ** 2 + (yr / b) ** 2 <= 1] += A
    return img
