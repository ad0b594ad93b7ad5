"""GPCSD1D -- 1D (laminar probe) Gaussian-process CSD model on the GPU.

Public surface of src/gpcsd/gpcsd1d.py (constructor :21, loglik :113, fit :130, predict :248, sample_prior :295,
extract/restore_model_params :84-102, update_lfp :104-111, JITTER :17); the arithmetic lives in libgpcsd_hip.so
(see model_base.GPCSDModel)."""
import numpy as np

from . import _hip
from .covariances import GPCSD1DSpatialCovSE, GPCSDTemporalCovSE, GPCSDTemporalCovMatern
from .model_base import GPCSDModel
from .priors import GPCSDInvGammaPrior, GPCSDHalfNormalPrior

np.seterr(all="ignore")      # the reference silences floating-point warnings module-wide (gpcsd1d.py:7)

JITTER = 1e-8


def _noise_param(sig2n_prior, default_sd, upper):
    """sig2n dict for a scalar prior, a per-electrode list of priors, or the default half-normal."""
    if sig2n_prior is None:
        sig2n_prior = GPCSDHalfNormalPrior(default_sd)
    if isinstance(sig2n_prior, list):
        n = len(sig2n_prior)
        return {"value": np.array([p.sample() for p in sig2n_prior]), "prior": sig2n_prior, "min": [1e-8] * n,
                "max": [upper] * n}
    return {"value": sig2n_prior.sample(), "prior": sig2n_prior, "min": 1e-8, "max": upper}


class GPCSD1D(GPCSDModel):
    dim = 1
    JITTER = JITTER
    _spatial_names = ("ell",)

    def __init__(self, lfp, x, t, a=None, b=None, ngl=100, spatial_cov=None, temporal_cov_list=None, R_prior=None,
                 sig2n_prior=None):
        """
        :param lfp: (n_spatial, n_time, n_trials) LFP, ideally scaled to unit standard deviation
        :param x: (n_spatial, 1) electrode depths in microns
        :param t: (n_time, 1) sample times in milliseconds
        :param a, b: integration limits of the forward model (default: min/max of x)
        :param ngl: number of Gauss-Legendre nodes
        :param spatial_cov: GPCSD1DSpatialCovSE (default built from x, a, b, ngl)
        :param temporal_cov_list: temporal covariances (default: one SE + one Matern)
        :param R_prior, sig2n_prior: priors (sig2n_prior may be a list with one prior per electrode)
        """
        self.lfp = np.atleast_3d(lfp)
        self.x = x
        self.t = t
        self.a = np.min(x) if a is None else a
        self.b = np.max(x) if b is None else b
        self.ngl = ngl
        self.spatial_cov = spatial_cov if spatial_cov is not None else GPCSD1DSpatialCovSE(x, a=self.a, b=self.b, ngl=ngl)
        self.temporal_cov_list = (temporal_cov_list if temporal_cov_list is not None
                                  else [GPCSDTemporalCovSE(t), GPCSDTemporalCovMatern(t)])
        xs = np.asarray(self.x, dtype=np.float64).squeeze()
        dmin, width = np.min(np.diff(xs)), np.max(xs) - np.min(xs)
        if R_prior is None:
            R_prior = GPCSDInvGammaPrior()
            R_prior.set_params(dmin, 0.5 * width)
        self.R = {"value": R_prior.sample(), "prior": R_prior, "min": 0.5 * dmin, "max": 0.8 * width}
        self.sig2n = _noise_param(sig2n_prior, 0.1, 0.5)

    def __str__(self):
        s = self._describe(["Integration bounds: (%d, %d)\n" % (self.a, self.b),
                            "Integration number points: %d\n" % self.ngl])
        s += "Spatial covariance ell prior: %s\n" % str(self.spatial_cov.params["ell"]["prior"])
        s += "Spatial covariance ell value %0.4g\n" % self.spatial_cov.params["ell"]["value"]
        return s + self._describe_temporal()

    def update_lfp(self, new_lfp, t, x=None):
        """Swap data (and optionally electrode positions); like the reference, new_lfp is stored as given."""
        if x is not None:
            self.x = x
            self.spatial_cov.x = x
        self.t = t
        for tc in self.temporal_cov_list:
            tc.t = t
        self.lfp = new_lfp

    def fit(self, n_restarts=10, method="L-BFGS-B", fix_R=False, verbose=False,
            options={"maxiter": 1000, "disp": False, "gtol": 1e-5, "ftol": 1e7 * np.finfo(float).eps}, starts=None, workers=1, batch=None):
        """Multi-restart MAP estimate of the hyper-parameters (`starts`: optional explicit log-parameter starts;
        `workers`: restarts run concurrently on this GPU, each on its own context / HIP stream; `batch`: that many
        restarts advance in lock-step, every objective + gradient evaluation of theirs served by one batched device call --
        default: all of them (within a memory budget); every restart walks the trajectory it walks on its own, bit for bit;
        batch=1: one restart after the other, like the reference's loop)."""
        return self._fit(n_restarts, method, fix_R, verbose, options, starts=starts, workers=workers, batch=batch)

    def sample_prior(self, ntrials):
        """Draw CSD trials from the prior: chol(Ks_csd + JITTER I) Z chol(Kt)^T with Z ~ N(0,1) drawn trial by trial
        from numpy's global RNG (same stream consumption as the reference)."""
        nt, nx = self.t.shape[0], self.x.shape[0]
        normals = np.stack([np.random.normal(0, 1, (nx, nt)) for _ in range(ntrials)], axis=2)
        return self._sample_prior_from_normals(normals, _hip.PRED_CSD)
