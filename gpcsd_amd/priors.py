"""Hyper-parameter priors (host scalar math, <= 10 flops per evaluation -- deliberately not on the GPU).

Mirrors src/gpcsd/priors.py of the reference: `GPCSDInvGammaPrior` (lpdf :23-28, set_params :30-32,
sample :34-35) and `GPCSDHalfNormalPrior` (lpdf :46-51, sample :53-54).  `dlpdf` (derivative of lpdf) is an
addition used by the analytic gradient that replaces the reference's autograd tape; `lpdf_many` / `dlpdf_many` are the same
expressions on arrays (all restarts of a lock-step fit at once).
"""
import numpy as np
from scipy import stats as _stats


class GPCSDPrior:
    """Base class; concrete priors provide lpdf / dlpdf / sample."""

    def lpdf(self, x):
        raise NotImplementedError

    def dlpdf(self, x):
        raise NotImplementedError

    def sample(self):
        raise NotImplementedError


class GPCSDInvGammaPrior(GPCSDPrior):
    """Inverse-gamma(alpha, beta), unnormalised log density -(alpha+1) log x - beta/x."""

    def __init__(self, alpha=1, beta=1):
        super().__init__()
        self.alpha = alpha
        self.beta = beta

    def __str__(self):
        return "InvGamma(%0.2f, %0.2f)" % (self.alpha, self.beta)

    def lpdf(self, x):
        if x <= 0:
            return -np.inf
        return -(self.alpha + 1.0) * np.log(x) - self.beta / x

    def dlpdf(self, x):
        return -(self.alpha + 1.0) / x + self.beta / (x * x)

    # the same two expressions elementwise on an array of values (the lock-step fit evaluates all restarts' priors at once)
    def lpdf_many(self, x):
        x = np.asarray(x, dtype=np.float64)
        with np.errstate(all="ignore"):
            return np.where(x <= 0, -np.inf, -(self.alpha + 1.0) * np.log(x) - self.beta / x)

    def dlpdf_many(self, x):
        x = np.asarray(x, dtype=np.float64)
        with np.errstate(all="ignore"):
            return -(self.alpha + 1.0) / x + self.beta / (x * x)

    def set_params(self, l, u):
        """Shape/scale so that most of the mass lies in [l, u] (same rule as the reference)."""
        ratio = (l + u) / (u - l)
        self.alpha = 2 + 9 * np.square(ratio)
        self.beta = 0.5 * (self.alpha - 1) * (l + u)

    def sample(self):
        return _stats.invgamma.rvs(self.alpha, scale=self.beta)


class GPCSDHalfNormalPrior(GPCSDPrior):
    """Half-normal(sd), unnormalised log density -x^2 / (2 sd^2)."""

    def __init__(self, sd=1):
        super().__init__()
        self.sd = sd

    def __str__(self):
        return "HalfNormal(%0.2f)" % (self.sd)

    def lpdf(self, x):
        if x <= 0:
            return -np.inf
        return -0.5 * np.square(x / self.sd)

    def dlpdf(self, x):
        return -x / (self.sd * self.sd)

    def lpdf_many(self, x):
        x = np.asarray(x, dtype=np.float64)
        with np.errstate(all="ignore"):
            return np.where(x <= 0, -np.inf, -0.5 * np.square(x / self.sd))

    def dlpdf_many(self, x):
        x = np.asarray(x, dtype=np.float64)
        return -x / (self.sd * self.sd)

    def sample(self):
        return _stats.halfnorm.rvs(scale=self.sd)
