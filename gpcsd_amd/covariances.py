"""Spatial (forward-model x squared-exponential) and temporal (SE / Matern-1/2) covariance operators.

Class names, constructor signatures, method names and the `params` dict layout
({'value','prior','min','max'} per hyper-parameter) follow src/gpcsd/covariances.py of the reference so model
code that mutates `obj.params['ell']['value']` keeps working.  Every Gram matrix is assembled on the GPU
(libgpcsd_hip.so): pairwise kernel evaluation in gram.hip, the A Kgl A^T contractions on the fp64 MFMA GEMM.
"""
import numpy as np
import scipy.special

from . import _hip
from .priors import GPCSDInvGammaPrior, GPCSDHalfNormalPrior
from .utility_functions import expand_grid, reduce_grid


def _gauss_legendre(a, b, n):
    """Gauss-Legendre rule on [-1,1] mapped to [a,b] (reference: covariances.py:22-27)."""
    nodes, weights = scipy.special.roots_legendre(int(n))
    half = 0.5 * (b - a)
    return 0.5 * (nodes + 1.0) * (b - a) + a, half * weights


def _span(v):
    v = np.sort(np.asarray(v, dtype=np.float64).reshape(-1))
    return float(np.min(np.diff(v))), float(v[-1] - v[0]), float(v[0]), float(v[-1])


# ----------------------------------------------------------------------------------------------- 1D spatial
class GPCSD1DSpatialCov:
    """Electrode positions + Gauss-Legendre rule on [a, b] (reference: covariances.py:12-27)."""

    def __init__(self, x, a, b, ngl):
        self.x = x
        self.a = np.min(x) if a is None else a
        self.b = np.max(x) if b is None else b
        self.ngl = ngl
        self.gl_x, self.gl_w = _gauss_legendre(self.a, self.b, ngl)


class GPCSD1DSpatialCovSE(GPCSD1DSpatialCov):
    """Squared-exponential CSD covariance seen through the 1D forward model (reference: covariances.py:29-96)."""

    def __init__(self, x, ell_prior=None, a=None, b=None, ngl=100):
        super().__init__(x, a, b, ngl)
        xs = np.asarray(self.x, dtype=np.float64).squeeze()
        dmin, width = float(np.min(np.diff(xs))), float(np.max(xs) - np.min(xs))
        if ell_prior is None:
            ell_prior = GPCSDInvGammaPrior()
            ell_prior.set_params(1.2 * dmin, 0.8 * width)
        self.params = {"ell": {"value": ell_prior.sample(), "prior": ell_prior, "min": 0.5 * dmin, "max": width}}

    def compute_Ks(self):
        """CSD spatial correlation exp(-(x-x')^2 / (2 ell^2)) at the electrodes."""
        return _hip.default_context().ks_csd_1d(self.x, self.params["ell"]["value"])

    def compKphig_1d(self, z, R):
        """Cross-covariance between LFP at the electrodes and CSD at z: (nx, nz)."""
        return _hip.default_context().kphig_1d(self.x, self.gl_x, self.gl_w, z, R, self.params["ell"]["value"])

    def compKphi_1d(self, R, xp=None):
        """LFP-LFP covariance between the electrodes and xp (default: the electrodes): (nx, nxp)."""
        return _hip.default_context().kphi_1d(self.x, self.gl_x, self.gl_w, R, self.params["ell"]["value"], xp=xp)


# ----------------------------------------------------------------------------------------------- 2D spatial
class GPCSD2DSpatialCov:
    """Electrode positions (n, 2) + tensor Gauss-Legendre rule (reference: covariances.py:99-137).

    The reference materialises delta1/delta2/delta_w (nx, G) and two (G, G) squared-distance tables at
    construction; here only the 1D node/weight vectors are kept (the kernels recompute distances on the fly)
    and the big tables are available as lazily computed properties for attribute parity."""

    def __init__(self, x, a1, b1, a2, b2, ngl1, ngl2):
        self.x = x
        self.a1, self.b1, self.a2, self.b2 = a1, b1, a2, b2
        self.ngl1, self.ngl2 = ngl1, ngl2
        self.gl_x1, self.gl_w1 = _gauss_legendre(a1, b1, ngl1)
        self.gl_x2, self.gl_w2 = _gauss_legendre(a2, b2, ngl2)

    def reset_x(self, x_new):
        """Swap the electrode positions, keeping the quadrature grid."""
        self.x = x_new

    # ---- attribute parity (host, on demand) ----
    @property
    def gl_x_grid(self):
        return expand_grid(self.gl_x1, self.gl_x2)

    @property
    def gl_w_prod(self):
        return np.prod(expand_grid(self.gl_w1, self.gl_w2), axis=1, keepdims=True)

    @property
    def delta1(self):
        return self.gl_x_grid[:, 0][None, :] - np.asarray(self.x)[:, 0][:, None]

    @property
    def delta2(self):
        return self.gl_x_grid[:, 1][None, :] - np.asarray(self.x)[:, 1][:, None]

    @property
    def delta_w(self):
        return np.sqrt(np.square(self.delta1) + np.square(self.delta2))

    @property
    def gl_x1_sqdist(self):
        g = self.gl_x_grid[:, 0]
        return np.square(g[:, None] - g[None, :])

    @property
    def gl_x2_sqdist(self):
        g = self.gl_x_grid[:, 1]
        return np.square(g[:, None] - g[None, :])


class GPCSD2DSpatialCovSE(GPCSD2DSpatialCov):
    """Anisotropic squared-exponential CSD covariance seen through the 2D forward model
    (reference: covariances.py:140-232)."""

    def __init__(self, x, ell_prior1=None, ell_prior2=None, a1=None, b1=None, a2=None, b2=None, ngl1=100, ngl2=100):
        super().__init__(x, a1, b1, a2, b2, ngl1, ngl2)
        u1, u2 = reduce_grid(x)
        d1, w1, lo1, hi1 = _span(u1)
        d2, w2, lo2, hi2 = _span(u2)
        if ell_prior1 is None:
            ell_prior1 = GPCSDInvGammaPrior()
            ell_prior1.set_params(2.0 * d1, 2.0 * w1)
        if ell_prior2 is None:
            ell_prior2 = GPCSDInvGammaPrior()
            ell_prior2.set_params(2.0 * d2, w2)
        self.params = {
            "ell1": {"value": ell_prior1.sample(), "prior": ell_prior1, "min": d1, "max": 5.0 * hi1 - lo1},
            "ell2": {"value": ell_prior2.sample(), "prior": ell_prior2, "min": d2, "max": hi2 - lo2},
        }

    def _gl(self):
        return self.gl_x1, self.gl_w1, self.gl_x2, self.gl_w2

    def compute_Ks(self):
        """CSD spatial correlation at the electrodes (product of two 1D SE kernels)."""
        return _hip.default_context().ks_csd_2d(self.x, self.params["ell1"]["value"], self.params["ell2"]["value"])

    def compKphig_2d(self, z, R, eps):
        """Cross-covariance between LFP at the electrodes and CSD at z (nz, 2): (nx, nz)."""
        return _hip.default_context().kphig_2d(self.x, *self._gl(), z, R, eps, self.params["ell1"]["value"],
                                               self.params["ell2"]["value"])

    def compKphi_2d(self, R, eps, xp=None):
        """LFP-LFP covariance between the electrodes and xp (default: the electrodes)."""
        return _hip.default_context().kphi_2d(self.x, *self._gl(), R, eps, self.params["ell1"]["value"],
                                              self.params["ell2"]["value"], xp=xp)


# ----------------------------------------------------------------------------------------------- temporal
class GPCSDTemporalCov:
    def __init__(self, t):
        self.t = t


class _StationaryTemporalCov(GPCSDTemporalCov):
    """Shared constructor of the two stationary kernels (ell ~ InvGamma from the sampling grid, sigma2 ~ HalfNormal)."""
    kind = None
    _sigma2_min = 1e-8

    def __init__(self, t, ell_prior=None, sigma2_prior=None):
        super().__init__(t)
        ts = np.asarray(self.t, dtype=np.float64).flatten()
        dt, width = float(np.min(np.diff(ts))), float(np.max(ts) - np.min(ts))
        if ell_prior is None:
            ell_prior = GPCSDInvGammaPrior()
            ell_prior.set_params(1.2 * dt, 0.8 * width)
        if sigma2_prior is None:
            sigma2_prior = GPCSDHalfNormalPrior(1.0)
        # sampling order matters for seeded reproducibility of scripts: ell first, then sigma2
        ell = ell_prior.sample()
        sigma2 = sigma2_prior.sample()
        self.params = {"ell": {"value": ell, "prior": ell_prior, "min": 0.5 * dt, "max": width},
                       "sigma2": {"value": sigma2, "prior": sigma2_prior, "min": self._sigma2_min, "max": np.inf}}

    def compute_Kt(self, t=None, tprime=None):
        """Temporal covariance between t (default self.t) and tprime (default self.t): (len(t), len(tprime))."""
        t = self.t if t is None else t
        tprime = self.t if tprime is None else tprime
        return _hip.default_context().gram_temporal(self.kind, t, tprime, self.params["ell"]["value"],
                                                    self.params["sigma2"]["value"])


    def compute_dKt(self, name):
        """d Kt / d params[name] on this object's own time grid (name: "ell" or "sigma2"): what fit()'s analytic gradient needs
        from a temporal covariance the library does not evaluate itself.  Any GPCSDTemporalCov subclass with a `compute_Kt` may
        be put in a model (the reference traces it with autograd, covariances.py:235-238, gpcsd1d.py:211); one that also offers
        `compute_dKt(name)` keeps the fit at one device evaluation per optimiser step instead of 2p + 1 (central differences).
        Host NumPy: this is the contract's default implementation for the two stationary kernels, used when they sit next to a
        user-defined component (alone, the library differentiates them on the device)."""
        ts = np.asarray(self.t, dtype=np.float64).reshape(-1)
        d = ts[:, None] - ts[None, :]
        ell, s2 = float(self.params["ell"]["value"]), float(self.params["sigma2"]["value"])
        if self.kind == _hip.KIND_SE:
            k = np.exp(-0.5 * np.square(d) / (ell * ell))
            dk = k * np.square(d) / (ell * ell * ell)
        else:
            ad = np.abs(d)
            k = np.exp(-ad / ell)
            dk = k * ad / (ell * ell)
        if name == "ell":
            return s2 * dk
        if name == "sigma2":
            return k
        raise KeyError(name)


class GPCSDTemporalCovSE(_StationaryTemporalCov):
    """sigma2 * exp(-(t-t')^2 / (2 ell^2))   (reference: covariances.py:239-271)."""
    kind = _hip.KIND_SE
    _sigma2_min = 1e-8


class GPCSDTemporalCovMatern(_StationaryTemporalCov):
    """sigma2 * exp(-|t-t'| / ell)   (Matern-1/2; reference: covariances.py:274-305; sigma2 lower bound 0)."""
    kind = _hip.KIND_MATERN
    _sigma2_min = 0
