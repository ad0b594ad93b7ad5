"""Shared test plumbing: golden fixtures -> oracle geometry / hyper-parameters."""
import os

import numpy as np

import cases as C
from oracle import gpcsd_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def load_model_case(name):
    """-> (case dict, golden npz, oracle geometry, oracle hparams WITHOUT jitter, lfp)."""
    c = C.model_cases()[name]
    g = golden("model_" + name)
    lfp = C.case_lfp(c)
    chk = np.array([lfp.sum(), np.abs(lfp).sum()])
    assert np.array_equal(chk, g["lfp_checksum"]), "seeded input regeneration drifted"
    if c["dim"] == 1:
        geom = O.Geometry1D(c["x"], c["t"], a=c["a"], b=c["b"], ngl=c["ngl"])
    else:
        geom = O.Geometry2D(c["x"], c["t"], ngl1=c["ngl1"], ngl2=c["ngl2"])
    temporal = [(k, ell, float(s2)) for (k, ell, _), s2 in zip(c["temporal"], g["temporal_sigma2"])]
    hp = O.make_hparams(c["R"], c["ell_s"], temporal, c["sig2n"], eps=c["eps"], jitter=0.0)
    return c, g, geom, hp, lfp


def with_jitter(hp, jitter):
    h = dict(hp)
    h["jitter"] = float(jitter)
    return h


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
