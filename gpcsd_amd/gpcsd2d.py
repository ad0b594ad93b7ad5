"""GPCSD2D -- 2D (Neuropixels-style array) Gaussian-process CSD model on the GPU.

Public surface of src/gpcsd/gpcsd2d.py (constructor :20-21, loglik :136, fit :153, predict :289, sample_prior :336,
extract/restore_model_params :103-125, update_lfp :127-134, JITTER :16)."""
import numpy as np

from . import _hip
from .covariances import GPCSD2DSpatialCovSE, GPCSDTemporalCovSE, GPCSDTemporalCovMatern
from .gpcsd1d import _noise_param
from .model_base import GPCSDModel
from .priors import GPCSDInvGammaPrior
from .utility_functions import reduce_grid

np.seterr(all="ignore")

JITTER = 1e-7


class GPCSD2D(GPCSDModel):
    dim = 2
    JITTER = JITTER
    _spatial_names = ("ell1", "ell2")

    def __init__(self, lfp, x, t, a1=None, b1=None, a2=None, b2=None, ngl1=20, ngl2=60, spatial_cov=None,
                 temporal_cov_list=None, R_prior=None, sig2n_prior=None, eps=None):
        """
        :param lfp: (n_spatial, n_time, n_trials) LFP, ideally scaled to unit standard deviation
        :param x: (n_spatial, 2) electrode coordinates in microns
        :param t: (n_time, 1) sample times in milliseconds
        :param a1, b1, a2, b2: integration limits per spatial dimension (default: coordinate min/max)
        :param ngl1, ngl2: Gauss-Legendre orders per dimension
        :param eps: zero-charge gap in front of the array (default 5 x the smallest electrode spacing)
        """
        self.lfp = np.atleast_3d(lfp)
        self.x = x
        self.t = t
        self.a1 = np.min(x[:, 0]) if a1 is None else a1
        self.b1 = np.max(x[:, 0]) if b1 is None else b1
        self.a2 = np.min(x[:, 1]) if a2 is None else a2
        self.b2 = np.max(x[:, 1]) if b2 is None else b2
        self.ngl1, self.ngl2 = ngl1, ngl2
        if spatial_cov is None:
            spatial_cov = GPCSD2DSpatialCovSE(self.x, a1=self.a1, b1=self.b1, a2=self.a2, b2=self.b2, ngl1=ngl1, ngl2=ngl2)
        self.spatial_cov = spatial_cov
        self.temporal_cov_list = (temporal_cov_list if temporal_cov_list is not None
                                  else [GPCSDTemporalCovSE(t), GPCSDTemporalCovMatern(t)])
        u1, u2 = reduce_grid(x)
        min_dx = np.min([np.min(np.diff(u1.squeeze())), np.min(np.diff(u2.squeeze()))])
        max_dx = np.max([self.b1 - self.a1, self.b2 - self.a2])
        if R_prior is None:
            R_prior = GPCSDInvGammaPrior()
            R_prior.set_params(min_dx, 0.5 * max_dx)
        self.R = {"value": R_prior.sample(), "prior": R_prior, "min": 0.5 * min_dx, "max": 0.8 * max_dx}
        self.eps = 5 * min_dx if eps is None else eps
        self.sig2n = _noise_param(sig2n_prior, 1.0, 10.0)

    def __str__(self):
        s = self._describe(["Integration bounds: (%d, %d), (%d, %d)\n" % (self.a1, self.b1, self.a2, self.b2),
                            "Integration number points: %d, %d\n" % (self.ngl1, self.ngl2)])
        for d in (1, 2):
            p = self.spatial_cov.params["ell%d" % d]
            s += "Spatial covariance ell prior (dim %d): %s\n" % (d, str(p["prior"]))
            s += "Spatial covariance ell value (dim %d) %0.4g\n" % (d, p["value"])
        return s + self._describe_temporal()

    def update_lfp(self, new_lfp, t, x=None):
        if x is not None:
            self.x = x
            self.spatial_cov.reset_x(x)
        self.t = t
        for tc in self.temporal_cov_list:
            tc.t = t
        self.lfp = np.atleast_3d(new_lfp)

    def _safe_loglik(self):
        # the 2D objective maps a failed factorisation to -inf instead of aborting the restart (gpcsd2d.py:215-218)
        try:
            return self.loglik()
        except np.linalg.LinAlgError:
            return -np.inf

    def fit(self, n_restarts=10, method="L-BFGS-B", fix_R=False, verbose=False, profile=False,
            options={"maxiter": 500, "disp": False, "gtol": 1e-5, "ftol": 1e7 * np.finfo(float).eps}, starts=None, workers=1, batch=None):
        """Multi-restart MAP estimate.  profile=True times one objective and one gradient evaluation per GPU kernel
        (the reference cProfiles them) and returns the table without optimising."""
        if profile:
            tp0 = self._sample_start(fix_R)
            ctx = self._sync_device()
            ctx.prof_reset()
            ctx.prof_enable(True)
            self._objective(tp0, fix_R)
            obj = ctx.prof_all()
            ctx.prof_reset()
            self._objective_grad(tp0, fix_R)
            grad = ctx.prof_all()
            ctx.prof_enable(False)
            return {"objective": obj, "gradient": grad}
        return self._fit(n_restarts, method, fix_R, verbose, options, starts=starts, workers=workers, batch=batch)

    def sample_prior(self, ntrials, type="csd", seed=1):
        """Prior draws of CSD and/or LFP; returns (csd, lfp) with NaN for the part not requested."""
        np.random.seed(seed)
        nt, nx = self.t.shape[0], self.x.shape[0]
        normals = np.random.normal(0, 1, (nx, nt, ntrials))
        csd = np.nan * np.zeros((nx, nt, ntrials))
        lfp = np.nan * np.zeros((nx, nt, ntrials))
        if type in ("csd", "both"):
            csd = self._sample_prior_from_normals(normals, _hip.PRED_CSD)
        if type in ("lfp", "both"):
            lfp = self._sample_prior_from_normals(normals, _hip.PRED_LFP)
        return csd, lfp
