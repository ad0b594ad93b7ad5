"""Traditional second-difference CSD estimators (comparison baselines, not part of the GP hot path).

Same results as src/gpcsd/predict_csd.py:3-31, written as array slices."""
import numpy as np


def predictcsd_trad_1d(lfp):
    """-(lfp[x+1] + lfp[x-1] - 2 lfp[x]) for interior electrodes, zero at the two ends; lfp (nx, nt, ntrials)."""
    lfp = np.asarray(lfp, dtype=np.float64)
    csd = np.zeros_like(lfp)
    csd[1:-1] = lfp[2:] + lfp[:-2] - 2.0 * lfp[1:-1]
    return -csd


def predictcsd_trad_2d(lfp):
    """Column-wise second difference on gridded data (nx1, nx2, nt, ntrials); NaN on the first/last column."""
    lfp = np.asarray(lfp, dtype=np.float64)
    csd = np.full(lfp.shape, np.nan)
    csd[:, 1:-1] = lfp[:, 2:] + lfp[:, :-2] - 2.0 * lfp[:, 1:-1]
    return -csd
