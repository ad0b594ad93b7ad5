"""Traditional second-difference CSD estimators: the comparison baseline of the reference's simulation studies and of its
auditory analysis (sim_from_gp_1D.py:88, sim_from_gp_2D.py:141, fit_gpcsd_baseline.py:106,147).

Same functions, arguments and results as src/gpcsd/predict_csd.py:3-31 (bit for bit: the same three-term sum, negated),
evaluated by one HBM-bound HIP kernel (gpcsd_trad_csd).  Like every operator of the package they need the HIP library and a GPU."""
import numpy as np

from . import _hip


def predictcsd_trad_1d(lfp):
    """-(lfp[x+1] + lfp[x-1] - 2 lfp[x]) for interior electrodes, zero at the two ends; lfp (nx, nt, ntrials)."""
    lfp = np.asarray(lfp, dtype=np.float64)
    if lfp.ndim != 3:
        raise ValueError("lfp must have shape (nx, nt, ntrials)")
    return _hip.default_context().trad_csd(lfp, 0, False)


def predictcsd_trad_2d(lfp):
    """Column-wise second difference on gridded data (nx1, nx2, nt, ntrials); NaN on the first/last column."""
    lfp = np.asarray(lfp, dtype=np.float64)
    if lfp.ndim != 4:
        raise ValueError("lfp must have shape (nx1, nx2, nt, ntrials)")
    return _hip.default_context().trad_csd(lfp, 1, True)
