"""CPU oracle for the GPCSD hot path -- TEST INFRASTRUCTURE ONLY.

This is a NumPy restatement (ours, written from the maths) of the algorithm the
reference runs in ``GPCSD1D/GPCSD2D.loglik()/predict()/sample_prior()`` and the
operators they call.  It is NOT the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product path (``gpcsd_amd``) never touches this module and fails
loudly when the HIP library is missing.

Pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md section 4), so the pin is ``tests/golden/*.npz`` -- outputs of the
reference itself, imported in the build container through the forward-only
autograd shim by ``tests/golden/generate_goldens.py``.  ``tests/test_oracle_golden.py``
checks every function below against those fixtures.

All citations are relative to /root/reference/.
Third-party numerics the reference relies on (not vendored there):
``numpy.linalg.eigh`` (LAPACK dsyevd), ``numpy.linalg.cholesky`` (dpotrf),
``scipy.special.roots_legendre``; the oracle calls the same routines.
"""
from __future__ import annotations

import numpy as np
import scipy.special

SE = 0        # squared-exponential temporal kernel  (covariances.py:257-271)
MATERN = 1    # Matern-1/2 (exponential) temporal kernel (covariances.py:291-305)


# --------------------------------------------------------------------------------------
# forward-model weights
# --------------------------------------------------------------------------------------
def b_fwd_1d(r, R):
    """sqrt((r/R)^2 + 1) - sqrt((r/R)^2)   -- src/gpcsd/forward_models.py:9-17."""
    q = np.square(np.asarray(r, dtype=np.float64) / R)
    return np.sqrt(q + 1.0) - np.sqrt(q)


def b_fwd_2d(delta1, delta2, R, eps, w=None):
    """log(R+eps+sqrt((R+eps)^2+w^2)) - log(eps+sqrt(eps^2+w^2)), w = |delta|
    -- src/gpcsd/forward_models.py:42-54."""
    if w is None:
        w = np.sqrt(np.square(delta1) + np.square(delta2))
    w = np.asarray(w, dtype=np.float64)
    return np.log(R + eps + np.sqrt((R + eps) ** 2 + w ** 2)) - np.log(eps + np.sqrt(eps ** 2 + w ** 2))


def gauss_legendre(a, b, n):
    """Gauss-Legendre nodes / weights mapped to [a, b] -- src/gpcsd/covariances.py:22-27, :114-123."""
    gx, gw = scipy.special.roots_legendre(n)
    return 0.5 * (gx + 1.0) * (b - a) + a, 0.5 * (b - a) * gw


def expand_grid(x1, x2):
    """(len(x1)*len(x2), 2) points, x1-major -- src/gpcsd/utility_functions.py:15-23."""
    x1 = np.asarray(x1, dtype=np.float64).reshape(-1)
    x2 = np.asarray(x2, dtype=np.float64).reshape(-1)
    return np.stack([np.repeat(x1, x2.size), np.tile(x2, x1.size)], axis=1)


def mykron(A, B):
    """Materialised Kronecker product -- src/gpcsd/utility_functions.py:35-42."""
    a1, a2 = A.shape
    b1, b2 = B.shape
    return (A[:, None, :, None] * B[None, :, None, :]).reshape(a1 * b1, a2 * b2)


# --------------------------------------------------------------------------------------
# temporal Gram matrices
# --------------------------------------------------------------------------------------
def temporal_gram(kind, t, tprime, ell, sigma2):
    """SE: sigma2*exp(-0.5*d^2/ell^2) (covariances.py:269-270);
    Matern-1/2: sigma2*exp(-|d|/ell) (covariances.py:303-304); d = t - tprime^T."""
    t = np.asarray(t, dtype=np.float64).reshape(-1, 1)
    tp = np.asarray(tprime, dtype=np.float64).reshape(-1, 1)
    d = t - tp.T
    if kind == SE:
        return sigma2 * np.exp(-0.5 * np.square(d) / np.square(ell))
    if kind == MATERN:
        return sigma2 * np.exp(-np.sqrt(np.square(d)) / ell)
    raise ValueError("unknown temporal kernel kind %r" % (kind,))


def temporal_sum(temporal, t, tprime=None):
    """Sum of the component Grams -- gpcsd1d.py:118-120 / gpcsd2d.py:141-143."""
    if tprime is None:
        tprime = t
    K = np.zeros((np.size(t), np.size(tprime)))
    for kind, ell, sigma2 in temporal:
        K = K + temporal_gram(kind, t, tprime, ell, sigma2)
    return K


# --------------------------------------------------------------------------------------
# 1D spatial operators
# --------------------------------------------------------------------------------------
def fwd_weights_1d(x, gl_x, gl_w, R):
    """A = gl_w * b_fwd_1d(gl_x - x, R), (nx, ngl) -- covariances.py:86-88."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, 1)
    return gl_w[None, :] * b_fwd_1d(gl_x[None, :] - x, R)


def ks_csd_1d(x, ell):
    """CSD prior covariance -- covariances.py:50-56."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, 1)
    return np.exp(-0.5 * np.square(x - x.T) / np.square(ell))


def kphig_1d(x, gl_x, gl_w, z, R, ell):
    """CSD-LFP cross covariance (nx, nz) -- covariances.py:58-72."""
    z = np.asarray(z, dtype=np.float64).reshape(-1, 1)
    Kc = np.exp(-0.5 * np.square((gl_x[None, :] - z) / ell)).T      # (ngl, nz)
    return fwd_weights_1d(x, gl_x, gl_w, R) @ Kc


def kphi_1d(x, gl_x, gl_w, R, ell, xp=None):
    """LFP-LFP covariance A Kgl A_xp^T, (nx, nxp) -- covariances.py:74-96."""
    A = fwd_weights_1d(x, gl_x, gl_w, R)
    Kgl = np.exp(-0.5 * np.square((gl_x[:, None] - gl_x[None, :]) / ell))
    Axp = A if xp is None else fwd_weights_1d(xp, gl_x, gl_w, R)
    return (A @ Kgl) @ Axp.T


# --------------------------------------------------------------------------------------
# 2D spatial operators
# --------------------------------------------------------------------------------------
def fwd_weights_2d(x, gl_x1, gl_w1, gl_x2, gl_w2, R, eps):
    """A = gl_w_prod^T * b_fwd_2d(|gl - x|), (nx, ngl1*ngl2) -- covariances.py:125-131, :220-221."""
    x = np.asarray(x, dtype=np.float64)
    g = expand_grid(gl_x1, gl_x2)
    wprod = np.prod(expand_grid(gl_w1, gl_w2), axis=1)
    d1 = g[:, 0][None, :] - x[:, 0][:, None]
    d2 = g[:, 1][None, :] - x[:, 1][:, None]
    return wprod[None, :] * b_fwd_2d(d1, d2, R, eps)


def ks_csd_2d(x, ell1, ell2):
    """CSD prior covariance -- covariances.py:177-186."""
    x = np.asarray(x, dtype=np.float64)
    x1 = x[:, 0][:, None]
    x2 = x[:, 1][:, None]
    return np.exp(-0.5 * np.square((x1 - x1.T) / ell1)) * np.exp(-0.5 * np.square((x2 - x2.T) / ell2))


def kgl_2d(gl_x1, gl_x2, ell1, ell2):
    """SE kernel on the tensor GL grid -- covariances.py:216 (sq-dists from :129-130)."""
    g = expand_grid(gl_x1, gl_x2)
    s1 = np.square(g[:, 0][:, None] - g[:, 0][None, :])
    s2 = np.square(g[:, 1][:, None] - g[:, 1][None, :])
    return np.exp(-0.5 * s1 / (ell1 ** 2)) * np.exp(-0.5 * s2 / (ell2 ** 2))


def kphig_2d(x, gl_x1, gl_w1, gl_x2, gl_w2, z, R, eps, ell1, ell2):
    """CSD-LFP cross covariance (nx, nz) -- covariances.py:188-202."""
    z = np.asarray(z, dtype=np.float64)
    g = expand_grid(gl_x1, gl_x2)
    Kc = (np.exp(-0.5 * np.square((g[:, 0][:, None] - z[:, 0][None, :]) / ell1))
          * np.exp(-0.5 * np.square((g[:, 1][:, None] - z[:, 1][None, :]) / ell2)))
    return fwd_weights_2d(x, gl_x1, gl_w1, gl_x2, gl_w2, R, eps) @ Kc


def kphi_2d(x, gl_x1, gl_w1, gl_x2, gl_w2, R, eps, ell1, ell2, xp=None):
    """LFP-LFP covariance A Kgl A_xp^T -- covariances.py:204-232."""
    A = fwd_weights_2d(x, gl_x1, gl_w1, gl_x2, gl_w2, R, eps)
    Kgl = kgl_2d(gl_x1, gl_x2, ell1, ell2)
    Axp = A if xp is None else fwd_weights_2d(xp, gl_x1, gl_w1, gl_x2, gl_w2, R, eps)
    return (A @ Kgl) @ Axp.T


# --------------------------------------------------------------------------------------
# Kronecker-eigen machinery
# --------------------------------------------------------------------------------------
def eig_D(Ks, Kt, sig2n):
    """Two symmetric eigendecompositions and Dvec -- src/gpcsd/utility_functions.py:44-64.

    D[x*nt + i] = es[x]*et[i] + sig2n (scalar) or + sig2n[x] (list; indexed by
    *eigen-index* x exactly as the reference does -- an approximation that parity
    must reproduce)."""
    nx = Ks.shape[0]
    nt = Kt.shape[0]
    if np.isscalar(sig2n) or np.ndim(sig2n) == 0:
        s2 = float(sig2n) * np.ones(nx * nt)
    else:
        s2 = np.repeat(np.asarray(sig2n, dtype=np.float64), nt)
    et, Qt = _eigh(Kt)
    es, Qs = _eigh(Ks)
    D = np.repeat(es, nt) * np.tile(et, nx) + s2
    return Qs, Qt, D


# LAPACK driver behind eig_D: None = numpy.linalg.eigh (dsyevd, what the reference calls); "evr" / "ev" / "evx" select
# another driver through scipy.linalg.eigh.  Only `driver_spread` below changes it.
EIGH_DRIVER = None


def _eigh(K):
    if EIGH_DRIVER is None:
        return np.linalg.eigh(K)
    import scipy.linalg
    return scipy.linalg.eigh(K, driver=EIGH_DRIVER)


def driver_spread(fn, drivers=("evr", "ev")):
    """max relative deviation of fn() (a float or an array) between numpy's dsyevd and other LAPACK drivers.

    With a per-electrode sig2n list the reference ties noise variance x to EIGEN-RANK x of Ks
    (utility_functions.py:54-63).  Ks is numerically rank deficient (about 35 of 384 eigenvalues are above rounding
    noise at the Neuropixels geometry), so the order of the remaining ones -- and with it the objective -- depends on the
    rounding of the eigensolver: equally correct LAPACK drivers disagree at 1e-6 .. 1e-3.  Tests gate the GPU path
    against this spread instead of a fixed tolerance in that one situation."""
    global EIGH_DRIVER
    base = np.asarray(fn(), dtype=np.float64)
    worst = 0.0
    try:
        for d in drivers:
            EIGH_DRIVER = d
            v = np.asarray(fn(), dtype=np.float64)
            worst = max(worst, float(np.max(np.abs(v - base)) / max(np.max(np.abs(base)), 1e-300)))
    finally:
        EIGH_DRIVER = None
    return worst


def loglik_from_K(lfp, Ks, Kt, sig2n):
    """-0.5*R*sum(log D) - 0.5*sum_r sum (Qs^T Y_r Qt)^2 / D, no 2*pi term
    -- gpcsd1d.py:113-128 / gpcsd2d.py:136-151 (Ks already carries the jitter).
    Trials are contracted with one einsum instead of the reference's strided loop."""
    lfp = np.atleast_3d(lfp)
    nx, nt, R = lfp.shape
    Qs, Qt, D = eig_D(Ks, Kt, sig2n)
    logdet = -0.5 * R * np.sum(np.log(D))
    Y = np.ascontiguousarray(np.moveaxis(lfp, 2, 0))          # (R, nx, nt) contiguous trials
    alpha = np.matmul(np.matmul(Qs.T, Y), Qt)                 # (R, nx, nt)
    quad = -0.5 * np.sum(np.square(alpha) / D.reshape(1, nx, nt))
    return float(logdet + quad)


def loglik_dense_cholesky(lfp, Ks, Kt, sig2n):
    """Independent cross-check (scalar sig2n only): dense K = kron(Ks,Kt) + sig2n*I,
    Cholesky log-det and triangular solves.  Not a reference code path; see
    SURVEY.md 'three facts' item 1."""
    lfp = np.atleast_3d(lfp)
    nx, nt, R = lfp.shape
    K = mykron(Ks, Kt) + float(sig2n) * np.eye(nx * nt)
    L = np.linalg.cholesky(K)
    y = lfp.reshape(nx * nt, R)
    import scipy.linalg
    v = scipy.linalg.solve_triangular(L, y, lower=True)
    return float(-R * np.sum(np.log(np.diag(L))) - 0.5 * np.sum(v * v))


# --------------------------------------------------------------------------------------
# model-level: hyper-parameters -> Ks, Kt -> loglik / predict
# --------------------------------------------------------------------------------------
class Geometry1D:
    """Coordinates + GL rule of a GPCSD1D model (covariances.py:12-27)."""
    dim = 1

    def __init__(self, x, t, a=None, b=None, ngl=100):
        self.x = np.asarray(x, dtype=np.float64).reshape(-1, 1)
        self.t = np.asarray(t, dtype=np.float64).reshape(-1, 1)
        self.a = float(np.min(x)) if a is None else float(a)
        self.b = float(np.max(x)) if b is None else float(b)
        self.ngl = int(ngl)
        self.gl_x, self.gl_w = gauss_legendre(self.a, self.b, self.ngl)


class Geometry2D:
    """Coordinates + tensor GL rule of a GPCSD2D model (covariances.py:99-131)."""
    dim = 2

    def __init__(self, x, t, a1=None, b1=None, a2=None, b2=None, ngl1=20, ngl2=60):
        self.x = np.asarray(x, dtype=np.float64)
        self.t = np.asarray(t, dtype=np.float64).reshape(-1, 1)
        self.a1 = float(np.min(self.x[:, 0])) if a1 is None else float(a1)
        self.b1 = float(np.max(self.x[:, 0])) if b1 is None else float(b1)
        self.a2 = float(np.min(self.x[:, 1])) if a2 is None else float(a2)
        self.b2 = float(np.max(self.x[:, 1])) if b2 is None else float(b2)
        self.ngl1, self.ngl2 = int(ngl1), int(ngl2)
        self.gl_x1, self.gl_w1 = gauss_legendre(self.a1, self.b1, self.ngl1)
        self.gl_x2, self.gl_w2 = gauss_legendre(self.a2, self.b2, self.ngl2)


def make_hparams(R, ell_s, temporal, sig2n, eps=0.0, jitter=0.0):
    """Plain container: ell_s is (ell,) in 1D, (ell1, ell2) in 2D; temporal is a list of
    (kind, ell, sigma2)."""
    return {"R": float(R), "eps": float(eps), "ell_s": tuple(float(e) for e in np.atleast_1d(ell_s)),
            "temporal": [(int(k), float(l), float(s)) for k, l, s in temporal],
            "sig2n": sig2n if np.ndim(sig2n) else float(sig2n), "jitter": float(jitter)}


def spatial_kphi(geom, hp, xp=None):
    if geom.dim == 1:
        return kphi_1d(geom.x, geom.gl_x, geom.gl_w, hp["R"], hp["ell_s"][0], xp=xp)
    return kphi_2d(geom.x, geom.gl_x1, geom.gl_w1, geom.gl_x2, geom.gl_w2, hp["R"], hp["eps"],
                   hp["ell_s"][0], hp["ell_s"][1], xp=xp)


def spatial_kphig(geom, hp, z):
    if geom.dim == 1:
        return kphig_1d(geom.x, geom.gl_x, geom.gl_w, z, hp["R"], hp["ell_s"][0])
    return kphig_2d(geom.x, geom.gl_x1, geom.gl_w1, geom.gl_x2, geom.gl_w2, z, hp["R"], hp["eps"],
                    hp["ell_s"][0], hp["ell_s"][1])


def spatial_ks_csd(geom, hp):
    if geom.dim == 1:
        return ks_csd_1d(geom.x, hp["ell_s"][0])
    return ks_csd_2d(geom.x, hp["ell_s"][0], hp["ell_s"][1])


def loglik(geom, hp, lfp):
    """GPCSD{1,2}D.loglik(): Ks = Kphi + JITTER*I (gpcsd1d.py:117, gpcsd2d.py:140)."""
    nx = geom.x.shape[0]
    Ks = spatial_kphi(geom, hp) + hp["jitter"] * np.eye(nx)
    Kt = temporal_sum(hp["temporal"], geom.t)
    return loglik_from_K(lfp, Ks, Kt, hp["sig2n"])


def loglik_tridiagonal(geom, hp, lfp):
    """The same log-likelihood WITHOUT the temporal eigenvectors (what gpcsd_amd's staged path evaluates): with
    Ks = Qs diag(es) Qs^T (comp_eig_D's spatial half, gpcsd1d.py:113-116) and only Kt = Q T Q^T, T tridiagonal (Householder),
    (Ks (x) Kt + sig2n I) is in the basis Qs (x) Q the block-diagonal set of shifted tridiagonal matrices es[x] T + sig2n I.
    sum log D = sum of the logs of their LDL^T pivots, the quadratic form (gpcsd1d.py:124-127) one forward substitution per
    (x, trial) row of Qs^T Y Q.  Scalar noise only (a list ties the noise to the eigen-rank, utility_functions.py:54-63).
    Checker code: a NumPy/SciPy cross-check of the algebra, independent of the device kernels."""
    import scipy.linalg as sla
    lfp = np.atleast_3d(lfp)
    nx, nt, R = lfp.shape
    assert np.ndim(hp["sig2n"]) == 0
    Ks = spatial_kphi(geom, hp) + hp["jitter"] * np.eye(nx)
    Kt = temporal_sum(hp["temporal"], geom.t)
    es, Qs = np.linalg.eigh(Ks)
    T, Q = sla.hessenberg(Kt, calc_q=True)            # symmetric input: T is tridiagonal up to rounding
    d, e = np.diag(T).copy(), np.diag(T, -1).copy()
    sig2 = float(hp["sig2n"])
    W = np.einsum("xa,xtr,tb->arb", Qs, lfp, Q)      # (x', r, t')
    sumlog, quad = 0.0, 0.0
    for a in range(nx):
        lam = es[a]
        piv = np.empty(nt)
        y = W[a].copy()                               # (R, nt): forward substitution L z = y in place
        piv[0] = lam * d[0] + sig2
        for k in range(1, nt):
            l = lam * e[k - 1] / piv[k - 1]
            piv[k] = lam * d[k] + sig2 - l * lam * e[k - 1]
            y[:, k] -= l * y[:, k - 1]
        sumlog += np.sum(np.log(piv))
        quad += np.sum(y * y / piv[None, :])
    return -0.5 * R * sumlog - 0.5 * quad


def predict(geom, hp, lfp, z, tstar, type="csd"):
    """Posterior mean, structured form of gpcsd1d.py:248-293 / gpcsd2d.py:289-334.

    InvY_r = Qs [ (Qs^T Y_r Qt) / D ] Qt^T               (== invmat @ yvec, :262-265)
    out_c[z, j, r] = sum_{x,i} Kcross[x, z] * Kt*_c[j, i] * InvY_r[x, i]    (:276-285)
    where Kt*_c = cov_c.compute_Kt(tstar) has shape (ntstar, nt) and the reference's
    reshape requires ntstar == nt (it raises ValueError otherwise; reproduced).
    No jitter on Ks here (gpcsd1d.py:258).  Returns dict with csd_list/csd/lfp_list/lfp."""
    lfp = np.atleast_3d(lfp)
    nx, nt, R = lfp.shape
    z = np.asarray(z, dtype=np.float64)
    tstar = np.asarray(tstar, dtype=np.float64).reshape(-1, 1)
    nz, ntstar = z.shape[0], tstar.shape[0]
    if ntstar != nt:
        raise ValueError("cannot reshape: predict requires len(t) == len(self.t) (gpcsd1d.py:279)")
    Ks = spatial_kphi(geom, hp)
    Kt = temporal_sum(hp["temporal"], geom.t)
    Qs, Qt, D = eig_D(Ks, Kt, hp["sig2n"])
    Y = np.ascontiguousarray(np.moveaxis(lfp, 2, 0))                      # (R, nx, nt)
    B = np.matmul(np.matmul(Qs.T, Y), Qt) / D.reshape(1, nx, nt)
    InvY = np.matmul(np.matmul(Qs, B), Qt.T)                              # (R, nx, nt)
    out = {}
    cross = {}
    if type in ("both", "csd"):
        cross["csd"] = spatial_kphig(geom, hp, z)
    if type in ("both", "lfp"):
        cross["lfp"] = spatial_kphi(geom, hp, xp=z)
    for name, Kc in cross.items():
        lst = []
        tot = np.zeros((nz, ntstar, R))
        S = np.matmul(Kc.T, InvY)                                         # (R, nz, nt)
        for kind, ell, sigma2 in hp["temporal"]:
            Ktstar = temporal_gram(kind, tstar, geom.t, ell, sigma2)      # (ntstar, nt)
            # mykron(Kc, Ktstar).T @ invy: row (x,i) of mykron is Kc[x,z]*Ktstar[i_row, j_col];
            # the reference indexes Ktstar's FIRST axis with the training index (quirk)
            comp = np.matmul(S, Ktstar)                                   # (R, nz, ntstar)
            comp = np.ascontiguousarray(np.moveaxis(comp, 0, 2))          # (nz, ntstar, R)
            lst.append(comp)
            tot = tot + comp
        out[name + "_list"] = lst
        out[name] = tot
    return out


def sample_prior_from_normals(geom, hp, normals, which="csd", jitter=None):
    """Ls Z_r Lt^T with host-supplied standard normals (nx, nt, R)
    -- gpcsd1d.py:295-309 / gpcsd2d.py:336-360 (RNG stream parity not required)."""
    nx = geom.x.shape[0]
    jit = hp["jitter"] if jitter is None else jitter
    Ks = spatial_ks_csd(geom, hp) if which == "csd" else spatial_kphi(geom, hp)
    Ls = np.linalg.cholesky(Ks + jit * np.eye(nx))
    Lt = np.linalg.cholesky(temporal_sum(hp["temporal"], geom.t))
    Z = np.moveaxis(np.asarray(normals, dtype=np.float64), 2, 0)
    out = np.matmul(np.matmul(Ls, Z), Lt.T)
    return np.ascontiguousarray(np.moveaxis(out, 0, 2)), Ls, Lt


# --------------------------------------------------------------------------------------
# priors + objective of fit()
# --------------------------------------------------------------------------------------
def invgamma_lpdf(x, alpha, beta):
    """priors.py:23-28."""
    return -np.inf if x <= 0 else -(alpha + 1.0) * np.log(x) - beta / x


def invgamma_params(l, u):
    """priors.py:30-32."""
    alpha = 2.0 + 9.0 * np.square((l + u) / (u - l))
    return alpha, 0.5 * (alpha - 1.0) * (l + u)


def halfnormal_lpdf(x, sd):
    """priors.py:46-51."""
    return -np.inf if x <= 0 else -0.5 * np.square(x / sd)


def hparams_from_tparams(tp, dim, kinds, n_sig, eps=0.0, jitter=0.0, R_fixed=None):
    """Log-parameter vector -> hyper-parameters, order of gpcsd1d.py:161-174 / gpcsd2d.py:196-211:
    [log(R/100), log(ell_s/100) x dim, (log ell_t, log sigma2_t) per component, log sig2n (x n_sig)]."""
    tp = np.asarray(tp, dtype=np.float64)
    R = np.exp(tp[0]) * 100.0 if R_fixed is None else R_fixed
    ell_s = np.exp(tp[1:1 + dim]) * 100.0
    p = 1 + dim
    temporal = []
    for k in kinds:
        temporal.append((k, np.exp(tp[p]), np.exp(tp[p + 1])))
        p += 2
    sig = float(np.exp(tp[p])) if n_sig == 1 else np.exp(tp[p:p + n_sig])
    return make_hparams(R, ell_s, temporal, sig, eps=eps, jitter=jitter)


def loglik_grad_fd(geom, lfp, tp, kinds, n_sig, eps=0.0, jitter=0.0, h=1e-6):
    """Central finite differences of loglik w.r.t. the log-parameter vector (no executable
    reference gradient exists: autograd is not installed)."""
    tp = np.asarray(tp, dtype=np.float64)
    g = np.zeros_like(tp)
    for i in range(tp.size):
        e = np.zeros_like(tp)
        e[i] = h
        fp = loglik(geom, hparams_from_tparams(tp + e, geom.dim, kinds, n_sig, eps, jitter), lfp)
        fm = loglik(geom, hparams_from_tparams(tp - e, geom.dim, kinds, n_sig, eps, jitter), lfp)
        g[i] = (fp - fm) / (2 * h)
    return g


def loglik_and_grad(geom, lfp, tp, kinds, n_sig, eps=0.0, jitter=0.0, R_fixed=None):
    """loglik and its gradient w.r.t. the log-parameter vector of fit() (gpcsd1d.py:161-174 / gpcsd2d.py:196-211) in closed
    form: what reverse mode through the reference's own forward pass yields (gpcsd1d.py:211 `jac=grad(obj_fun)`: matmuls,
    elementwise kernels and the eigh VJP  Kbar = Q (diag(wbar) + (Q^T Qbar) o F) Q^T, F_xy = 1 / (w_y - w_x)), simplified by hand.
    No executable reference gradient exists here (autograd is not installed); this restatement is pinned by central
    differences of `loglik` -- itself pinned by the reference's outputs -- in tests/test_oracle_golden.py.

    With B_r = alpha_r / D, alpha_r = Qs^T Y_r Qt, D_xi = es_x et_i + s2_x (s2 indexed by EIGEN-index x, utility_functions.py:54-63):
      dL/dD = -R/2 / D + 1/2 sum_r B_r^2;   dL/ds2_x = sum_i dL/dD_xi;
      Ghat_s[x,y] = 1/2 sum_{r,i} B[x,i] et_i B[y,i] - R/2 delta_xy sum_i et_i / D_xi
                    - 1/2 (s2_x - s2_y) / (es_y - es_x) sum_{r,i} B[x,i] B[y,i]      (x != y; zero for a scalar s2),
      Ghat_t[i,j] = 1/2 sum_{r,x} B[x,i] es_x B[x,j] - R/2 delta_ij sum_x es_x / D_xi;   Gs = Qs Ghat_s Qs^T, Gt = Qt Ghat_t Qt^T;
      dL/dtheta_t = <Gt, dKt/dtheta>,  dL/dell_s = <A^T Gs A, dKgl/dell>,  dL/dR = 2 <Gs A Kgl, dA/dR>.
    Returns (loglik, gradient): d loglik / d tp_k = natural derivative x natural value (every parameter is an exponential of
    its tp entry); the R entry is 0 when R_fixed is given (gpcsd1d.py:161-162)."""
    tp = np.asarray(tp, dtype=np.float64)
    hp = hparams_from_tparams(tp, geom.dim, kinds, n_sig, eps=eps, jitter=jitter, R_fixed=R_fixed)
    lfp = np.atleast_3d(lfp)
    nx, nt, R = lfp.shape
    Rv, ell_s = hp["R"], hp["ell_s"]
    if geom.dim == 1:
        A = fwd_weights_1d(geom.x, geom.gl_x, geom.gl_w, Rv)
        dx = geom.gl_x[:, None] - geom.gl_x[None, :]
        Kgl = np.exp(-0.5 * np.square(dx / ell_s[0]))
        dKgl = [Kgl * np.square(dx) / ell_s[0] ** 3]
        r = geom.gl_x[None, :] - geom.x
        q = np.square(r / Rv)
        dA = geom.gl_w[None, :] * (np.sqrt(q) - q / np.sqrt(q + 1.0)) / Rv
    else:
        A = fwd_weights_2d(geom.x, geom.gl_x1, geom.gl_w1, geom.gl_x2, geom.gl_w2, Rv, hp["eps"])
        g = expand_grid(geom.gl_x1, geom.gl_x2)
        s1 = np.square(g[:, 0][:, None] - g[:, 0][None, :])
        s2 = np.square(g[:, 1][:, None] - g[:, 1][None, :])
        Kgl = np.exp(-0.5 * s1 / ell_s[0] ** 2) * np.exp(-0.5 * s2 / ell_s[1] ** 2)
        dKgl = [Kgl * s1 / ell_s[0] ** 3, Kgl * s2 / ell_s[1] ** 3]
        wprod = np.prod(expand_grid(geom.gl_w1, geom.gl_w2), axis=1)
        w2 = np.square(g[:, 0][None, :] - geom.x[:, 0][:, None]) + np.square(g[:, 1][None, :] - geom.x[:, 1][:, None])
        dA = wprod[None, :] / np.sqrt((Rv + hp["eps"]) ** 2 + w2)
    T = A @ Kgl
    Ks = T @ A.T + hp["jitter"] * np.eye(nx)
    Kt = temporal_sum(hp["temporal"], geom.t)
    et, Qt = _eigh(Kt)                                    # eig_D (utility_functions.py:44-64) with the spectra kept
    es, Qs = _eigh(Ks)
    D = es[:, None] * et[None, :] + (np.asarray(hp["sig2n"], dtype=np.float64)[:, None] if n_sig > 1 else float(hp["sig2n"]))
    Y = np.ascontiguousarray(np.moveaxis(lfp, 2, 0))
    alpha = np.matmul(np.matmul(Qs.T, Y), Qt)
    Bm = alpha / D[None]
    ll = float(-0.5 * R * np.sum(np.log(D)) - 0.5 * np.sum(alpha * Bm))
    P = -0.5 * R / D + 0.5 * np.sum(np.square(Bm), axis=0)
    BmT = np.transpose(Bm, (0, 2, 1))
    Ghs = 0.5 * np.sum(np.matmul(Bm * et[None, None, :], BmT), axis=0) - 0.5 * R * np.diag(np.sum(et[None, :] / D, axis=1))
    Ght = 0.5 * np.sum(np.matmul(BmT, Bm * es[None, :, None]), axis=0) - 0.5 * R * np.diag(np.sum(es[:, None] / D, axis=0))
    if n_sig > 1:
        sv = np.asarray(hp["sig2n"], dtype=np.float64)
        S = np.sum(np.matmul(Bm, BmT), axis=0)
        with np.errstate(divide="ignore", invalid="ignore"):
            F = (sv[:, None] - sv[None, :]) / (es[None, :] - es[:, None])
        F[np.arange(nx), np.arange(nx)] = 0.0
        Ghs = Ghs - 0.5 * F * S
        dsig = np.sum(P, axis=1)
    else:
        dsig = np.array([np.sum(P)])
    Gs = Qs @ Ghs @ Qs.T
    Gt = Qt @ Ght @ Qt.T
    grad = np.zeros_like(tp)
    grad[0] = 0.0 if R_fixed is not None else 2.0 * np.sum((Gs @ T) * dA) * Rv
    M = A.T @ (Gs @ A)
    for k in range(geom.dim):
        grad[1 + k] = np.sum(M * dKgl[k]) * ell_s[k]
    p = 1 + geom.dim
    tt = geom.t.reshape(-1, 1)
    d = tt - tt.T
    for kind, ell, sigma2 in hp["temporal"]:
        Kc = temporal_gram(kind, tt, tt, ell, sigma2)
        dK_dell = Kc * np.square(d) / ell ** 3 if kind == SE else Kc * np.abs(d) / ell ** 2
        grad[p] = np.sum(Gt * dK_dell) * ell
        grad[p + 1] = np.sum(Gt * Kc)                     # dK/dsigma2 * sigma2 = K_c
        p += 2
    grad[p:p + n_sig] = dsig * np.atleast_1d(hp["sig2n"])
    return ll, grad


def invgamma_dlpdf(x, alpha, beta):
    """d/dx of invgamma_lpdf."""
    return -(alpha + 1.0) / x + beta / (x * x)


def halfnormal_dlpdf(x, sd):
    """d/dx of halfnormal_lpdf."""
    return -x / (sd * sd)


# --------------------------------------------------------------------------------------
# downstream per-trial consumer of (Qs, Qt, Dvec)
# --------------------------------------------------------------------------------------
def whitened_quad(Qs, Qt, Dvec, resid):
    """sum((Qs^T resid_b Qt)^2 / Dvec) per trial b of resid (nx, nt, nb) -- the projection step of loglik
    (gpcsd1d.py:124-127) as auditory_lfp/fit_mean_function.py:317-318 reuses it."""
    resid = np.atleast_3d(resid)
    nx, nt, nb = resid.shape
    alpha = np.matmul(np.matmul(Qs.T, np.moveaxis(resid, 2, 0)), Qt)
    return np.sum(np.square(alpha) / np.asarray(Dvec).reshape(1, nx, nt), axis=(1, 2))


def shift_objective(Qs, Qt, Dvec, lfp_trial, mu_lfp, t, tau, mutau=0.0, sigtau=10.0):
    """Negative log posterior of the per-component time shifts tau of one trial --
    auditory_lfp/fit_mean_function.py:311-321: mean = background mu_lfp[:, :, 0] + sum_i mu_i(t + tau_i) (linear
    interpolation with extrapolation, scipy interp1d as at :308), residual whitened by the cached decomposition,
    Gaussian prior N(mutau, sigtau^2) on each shift."""
    import scipy.interpolate
    tt = np.asarray(t, dtype=np.float64).reshape(-1)
    tau = np.asarray(tau, dtype=np.float64).reshape(-1)
    mu = np.array(mu_lfp[:, :, 0], dtype=np.float64, copy=True)
    for i in range(1, mu_lfp.shape[2]):
        f = scipy.interpolate.interp1d(tt, mu_lfp[:, :, i], axis=1, fill_value="extrapolate")
        mu += f(tt + tau[i - 1])
    quad = whitened_quad(Qs, Qt, Dvec, (lfp_trial - mu)[:, :, None])[0]
    return 0.5 * quad + 0.5 * np.sum(np.square((tau - mutau) / sigtau))


# --------------------------------------------------------------------------------------
# trapezoid forward simulators
# --------------------------------------------------------------------------------------
def fwd_model_1d(arr, x, z, R, varsigma=1.0):
    """R/(2 varsigma) * trapz_x( b_fwd_1d(z_i - x, R) * arr[:, t] ) -- forward_models.py:20-39."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    z = np.asarray(z, dtype=np.float64).reshape(-1)
    B = b_fwd_1d(z[:, None] - x[None, :], R)                    # (nz, nx)
    dx = np.diff(x)
    wts = np.zeros_like(x)
    wts[:-1] += 0.5 * dx
    wts[1:] += 0.5 * dx
    return R / (2.0 * varsigma) * (B * wts[None, :]) @ np.asarray(arr, dtype=np.float64)


def fwd_model_2d(arr, x1, x2, z, R, eps, varsigma=1.0):
    """Double trapezoid of b_fwd_2d(z_i - grid) * arr[:, :, t] -- forward_models.py:57-81."""
    x1 = np.asarray(x1, dtype=np.float64).reshape(-1)
    x2 = np.asarray(x2, dtype=np.float64).reshape(-1)
    z = np.asarray(z, dtype=np.float64)
    arr = np.asarray(arr, dtype=np.float64)

    def tw(v):
        d = np.diff(v)
        w = np.zeros_like(v)
        w[:-1] += 0.5 * d
        w[1:] += 0.5 * d
        return w
    w1, w2 = tw(x1), tw(x2)
    d1 = z[:, 0][:, None, None] - x1[None, :, None]
    d2 = z[:, 1][:, None, None] - x2[None, None, :]
    W = b_fwd_2d(d1, d2, R, eps) * w1[None, :, None] * w2[None, None, :]   # (nz, nx1, nx2)
    return np.tensordot(W, arr, axes=([1, 2], [0, 1]))


def trad_csd_1d(lfp):
    """predictcsd_trad_1d (predict_csd.py:3-16): minus the second difference over electrodes, zero at the two ends."""
    lfp = np.asarray(lfp, dtype=np.float64)
    csd = np.zeros_like(lfp)
    csd[1:-1] = lfp[2:] + lfp[:-2] - 2.0 * lfp[1:-1]
    return -csd


def trad_csd_2d(lfp):
    """predictcsd_trad_2d (predict_csd.py:19-31): column-wise on gridded data (nx1, nx2, nt, ntrials), NaN on the end columns."""
    lfp = np.asarray(lfp, dtype=np.float64)
    csd = np.full(lfp.shape, np.nan)
    csd[:, 1:-1] = lfp[:, 2:] + lfp[:, :-2] - 2.0 * lfp[:, 1:-1]
    return -csd
