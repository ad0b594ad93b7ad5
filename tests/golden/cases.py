"""Shared case definitions for the golden fixtures (inputs only -- no reference code).

Used by ``generate_goldens.py`` (which feeds them to the reference, imported from
/root/reference in the build container) and by the tests (which feed the same
inputs to the oracle and to the HIP path).  Inputs that are large are regenerated
from a legacy ``RandomState`` seed (bit-stable across NumPy versions) and verified
against a stored checksum.
"""
import numpy as np

SE, MATERN = 0, 1


def neuropixels_xy(nchan):
    """384-channel Neuropixels checkerboard geometry (formula of neuropixels/extract_data.py:36-42)."""
    c = np.arange(nchan)
    xs = np.array([16.0, 48.0, 0.0, 32.0])[c % 4]
    ys = np.floor(c / 2) * 20.0
    return np.stack([xs, ys], axis=1)


def grid_xy(n1, n2, lo1, hi1, lo2, hi2):
    x1 = np.linspace(lo1, hi1, n1)
    x2 = np.linspace(lo2, hi2, n2)
    return np.stack([np.repeat(x1, n2), np.tile(x2, n1)], axis=1)


def synth_lfp(seed, nx, nt, R):
    return np.random.RandomState(seed).standard_normal((nx, nt, R))


# ---- model-level cases --------------------------------------------------------------
# each: dim, x, t, GL setup, hyper-parameters (R, eps, ell_s, temporal [(kind, ell, sigma2)], sig2n), lfp seed/shape
def model_cases():
    cases = {}
    x24 = np.linspace(0, 2300, 24)[:, None]
    temporal_1d = [(SE, 20.0, 0.5), (MATERN, 5.0, 0.7)]
    # BASELINE cfg1: 1D 24 x 100 x 1
    cases["cfg1_1d_24x100x1"] = dict(dim=1, x=x24, t=np.linspace(0, 100, 100)[:, None], a=0.0, b=2300.0, ngl=100,
                                     R=100.0, eps=0.0, ell_s=(200.0,), temporal=temporal_1d, sig2n=0.05,
                                     seed=11, R_trials=1)
    # scaled-down cfg2: 1D 24 x 500 x 8
    cases["cfg2s_1d_24x500x8"] = dict(dim=1, x=x24, t=np.arange(500.0)[:, None], a=0.0, b=2300.0, ngl=100,
                                      R=100.0, eps=0.0, ell_s=(200.0,), temporal=temporal_1d, sig2n=0.05,
                                      seed=12, R_trials=8, predict_light=True)
    # wider integration bounds, different hyper-parameters, SE only
    cases["1d_wide_24x60x3"] = dict(dim=1, x=x24, t=np.linspace(0, 59, 60)[:, None], a=-200.0, b=2600.0, ngl=60,
                                    R=150.0, eps=0.0, ell_s=(320.0,), temporal=[(SE, 8.0, 1.3)], sig2n=0.2,
                                    seed=13, R_trials=3)
    # per-electrode noise list (auditory_lfp/fit_gpcsd_baseline.py:85-89 usage); short ell keeps eigenvalues separated
    cases["1d_siglist_12x40x4"] = dict(dim=1, x=np.linspace(0, 1100, 12)[:, None], t=np.linspace(0, 39, 40)[:, None],
                                       a=0.0, b=1100.0, ngl=50, R=80.0, eps=0.0, ell_s=(60.0,),
                                       temporal=[(SE, 6.0, 0.8), (MATERN, 3.0, 0.4)],
                                       sig2n=np.linspace(0.02, 0.3, 12), seed=14, R_trials=4)
    # odd sizes (not multiples of 16) to exercise tile edges
    cases["1d_odd_17x37x5"] = dict(dim=1, x=np.linspace(0, 1600, 17)[:, None], t=np.linspace(0, 72, 37)[:, None],
                                   a=0.0, b=1600.0, ngl=33, R=120.0, eps=0.0, ell_s=(150.0,),
                                   temporal=[(MATERN, 9.0, 1.1)], sig2n=0.1, seed=15, R_trials=5)
    # 2D small grid (sim_from_gp_2D.py:21-34 shape, reduced)
    xg = grid_xy(4, 12, 0.0, 48.0, 0.0, 440.0)
    cases["2d_grid_48x40x2"] = dict(dim=2, x=xg, t=np.linspace(0, 39, 40)[:, None] * 0.5, ngl1=10, ngl2=24,
                                    R=60.0, eps=20.0, ell_s=(30.0, 100.0), temporal=[(SE, 4.0, None), (MATERN, 2.0, None)],
                                    temporal_sigma2_rel=(0.5, 0.7), sig2n=0.05, seed=21, R_trials=2)
    # 2D neuropixels first 96 channels
    cases["2d_npx_96x120x3"] = dict(dim=2, x=neuropixels_xy(96), t=0.4 * np.arange(120.0)[:, None], ngl1=12, ngl2=30,
                                    R=100.0, eps=80.0, ell_s=(40.0, 150.0), temporal=[(SE, 20.0, None), (MATERN, 5.0, None)],
                                    temporal_sigma2_rel=(0.5, 0.7), sig2n=0.05, seed=22, R_trials=3)
    # scaled-down cfg3: full 384 x 500 geometry, 2 trials (loglik only -- predict is infeasible in the reference)
    cases["cfg3s_2d_384x500x2"] = dict(dim=2, x=neuropixels_xy(384), t=0.4 * np.arange(500.0)[:, None], ngl1=20, ngl2=60,
                                       R=100.0, eps=80.0, ell_s=(40.0, 150.0), temporal=[(SE, 20.0, None), (MATERN, 5.0, None)],
                                       temporal_sigma2_rel=(0.5, 0.7), sig2n=0.05, seed=23, R_trials=2, loglik_only=True)
    return cases


def case_lfp(c):
    return synth_lfp(c["seed"], c["x"].shape[0], c["t"].shape[0], c["R_trials"])
