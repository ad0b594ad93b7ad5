"""Which optimiser driver fit() uses, and that it never changes silently (host logic, no device needed).

fit() steps SciPy's own L-BFGS-B through the private reverse-communication entry point scipy.optimize._lbfgsb.setulb (the
trajectories of the reference's minimize() call, gpcsd1d.py:211 / gpcsd2d.py:250, bit for bit).  This file has no skip marker on
purpose: on the SciPy of this image the probe MUST succeed, and wherever it does not -- or another `method` is asked for --
fit() must say so with a RuntimeWarning naming the cost instead of quietly taking the slower thread rendezvous."""
import warnings

import numpy as np
import pytest
import scipy

from gpcsd_amd import lbfgsb_lockstep as L
from gpcsd_amd.model_base import GPCSDModel


def test_setulb_driver_is_available_on_this_scipy():
    assert L.available(), ("scipy %s: stepping scipy.optimize._lbfgsb.setulb no longer reproduces minimize(method='L-BFGS-B') "
                           "on the probe problem -- fit() would fall back to the threads driver (0.45-0.69 of the rate); update "
                           "gpcsd_amd/lbfgsb_lockstep.py to this SciPy's protocol" % scipy.__version__)


def test_probe_compares_against_minimize_and_rejects_a_changed_protocol(monkeypatch):
    """The probe is a comparison with minimize() on a problem with an active bound, not a finiteness check: a driver whose steps
    differ from SciPy's in any way is rejected."""
    fg, x0, bounds, opts = L._probe_problem()
    ref = scipy.optimize.minimize(fg, x0, jac=True, method="L-BFGS-B", bounds=bounds, options=opts)
    assert ref.nit >= 5 and ref.x[2] == bounds[2][0]                   # several iterations, the third bound active at the optimum
    real = L.minimize_many

    def skewed(batch_fn, starts, bnds, options, width, **kw):            # same optimum to 1e-9, not the same bits
        out, st = real(batch_fn, starts, bnds, options, width, **kw)
        return {k: (v[0] * (1.0 + 1e-15), v[1], v[2]) for k, v in out.items()}, st
    monkeypatch.setattr(L, "minimize_many", skewed)
    monkeypatch.setattr(L, "_probed", None)
    assert L.available() is False
    monkeypatch.setattr(L, "minimize_many", real)
    monkeypatch.setattr(L, "_probed", None)
    assert L.available() is True


class _P:
    params = {}


class _Stub(GPCSDModel):
    dim = 1

    def __init__(self):
        self.R, self.sig2n, self.spatial_cov, self.temporal_cov_list, self.best = {"value": 1.0}, {"value": 0.1}, _P(), [], None

    def _bounds(self):
        return [(-3.0, 3.0), (-3.0, 3.0)]

    def _objective_and_grad(self, tp, fix_R, fd_step=1e-6):
        x, y = tp
        return float((x * x - 1.0) ** 2 + 0.3 * x + (y - 0.5) ** 2), np.array([4.0 * x * (x * x - 1.0) + 0.3, 2.0 * (y - 0.5)])

    def _batch_can_evaluate(self):
        return True

    def _objective_and_grad_batch(self, items, fix_R):
        return {k: self._objective_and_grad(tp, fix_R) for k, tp in items}

    def _local_lfp(self):
        return np.zeros((1, 1, 1))

    def _current_tparams(self):
        return np.zeros(2)

    def _set_from_tparams(self, tp, fix_R):
        self.best = np.array(tp, dtype=np.float64)


STARTS = [np.array([s, -s]) for s in np.linspace(-2.0, 2.0, 5)]


def test_fallback_to_the_threads_driver_warns(monkeypatch):
    m = _Stub()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                   # the default driver on this SciPy: no warning at all
        m._fit(5, "L-BFGS-B", False, False, {"maxiter": 50}, starts=STARTS, batch=3)
    assert m.fit_driver_used_ == "setulb"
    best = m.best.copy()
    # another method: warned, same machinery otherwise
    m = _Stub()
    with pytest.warns(RuntimeWarning, match="not L-BFGS-B"):
        m._fit(5, "TNC", False, False, {"maxfun": 200}, starts=STARTS, batch=3)
    assert m.fit_driver_used_ == "threads"
    # a SciPy whose setulb does not reproduce minimize(): warned, and the optimum is still the sequential loop's
    monkeypatch.setattr(L, "_probed", False)
    m = _Stub()
    with pytest.warns(RuntimeWarning, match="available\\(\\) is False"):
        m._fit(5, "L-BFGS-B", False, False, {"maxiter": 50}, starts=STARTS, batch=3)
    assert m.fit_driver_used_ == "threads" and np.array_equal(m.best, best)
    # chosen explicitly: no warning
    m = _Stub()
    m.fit_driver = "threads"
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m._fit(5, "L-BFGS-B", False, False, {"maxiter": 50}, starts=STARTS, batch=3)
    assert m.fit_driver_used_ == "threads" and np.array_equal(m.best, best)
