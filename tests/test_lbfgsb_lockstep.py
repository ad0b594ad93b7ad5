"""The lock-step L-BFGS-B driver (host logic of fit(batch=k), no device needed): SciPy's own optimiser stepped through its
reverse-communication interface must walk, bit for bit, the trajectory scipy.optimize.minimize walks -- the reference's call
(gpcsd1d.py:211, gpcsd2d.py:250) -- from the same start, with every kind of bound, truncated or converged, and a chain whose
evaluation fails must end alone."""
import numpy as np
import pytest
import scipy.optimize

from gpcsd_amd import lbfgsb_lockstep as L

pytestmark = pytest.mark.skipif(not L.available(), reason="this SciPy does not expose _lbfgsb.setulb in the 1.15 form")

N = 7
_rs = np.random.RandomState(0)
_A = _rs.standard_normal((N, N))
_A = _A @ _A.T + 0.5 * np.eye(N)
BOUNDS = [(-1.0, 2.0), (None, 1.5), (-0.3, None), (None, None), (-2, 2), (-2, 0.1), (0.2, np.inf)]


def fg(x):
    f = 0.5 * x @ _A @ x + np.sum(np.sin(3 * x)) + 0.1 * np.sum(x ** 4)
    return float(f), _A @ x + 3 * np.cos(3 * x) + 0.4 * x ** 3


@pytest.mark.parametrize("maxiter,width", [(200, 16), (7, 5), (1, 40)])
def test_lockstep_driver_is_bitwise_scipy_minimize(maxiter, width):
    starts = [np.random.RandomState(10 + k).uniform(-3, 3, N) for k in range(40)]
    opts = {"maxiter": maxiter, "gtol": 1e-5, "ftol": 1e7 * np.finfo(float).eps, "disp": False}
    orders = []

    def batch_fn(items):
        orders.append([k for k, _ in items])
        return {k: fg(x) for k, x in items}
    out, stats = L.minimize_many(batch_fn, starts, BOUNDS, opts, width=width)
    assert all(o == sorted(o) for o in orders) and max(len(o) for o in orders) == min(width, 40)
    assert stats["batches"] == len(orders) and stats["points"] == sum(len(o) for o in orders)
    for k, s0 in enumerate(starts):
        ref = scipy.optimize.minimize(fg, s0, jac=True, method="L-BFGS-B", bounds=BOUNDS, options=opts)
        f, x, msg = out[k]
        assert f == ref.fun and np.array_equal(x, ref.x) and msg == ref.message, (k, f, ref.fun, msg, ref.message)


def test_lockstep_driver_failed_evaluation_ends_one_chain():
    starts = [np.random.RandomState(50 + k).uniform(-1, 1, N) for k in range(6)]
    opts = {"maxiter": 50}
    calls = {"n": 0}

    def batch_fn(items):
        calls["n"] += 1
        out = {}
        for k, x in items:
            out[k] = np.linalg.LinAlgError("boom") if (k == 3 and calls["n"] >= 3) else fg(x)
        return out
    out, _ = L.minimize_many(batch_fn, starts, BOUNDS, opts, width=4)
    assert isinstance(out[3], np.linalg.LinAlgError)
    for k in (0, 1, 2, 4, 5):
        ref = scipy.optimize.minimize(fg, starts[k], jac=True, method="L-BFGS-B", bounds=BOUNDS, options=opts)
        assert out[k][0] == ref.fun and np.array_equal(out[k][1], ref.x)
    with pytest.raises(KeyError):                 # an exception type the caller did not declare is raised, not swallowed
        L.minimize_many(lambda items: {k: KeyError("x") for k, _ in items}, starts[:2], BOUNDS, opts, width=2)
    with pytest.raises(ValueError):
        L.minimize_many(batch_fn, starts, BOUNDS[:-1], opts, width=2)


def test_fit_drivers_agree_on_the_stub_model():
    """GPCSDModel._fit with fit_driver 'setulb' (default) and 'threads' (round 2) and batch=1 (the reference's loop): same optima."""
    from gpcsd_amd.model_base import GPCSDModel

    class _P:
        params = {}

    class Stub(GPCSDModel):
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
    starts = [np.array([s, -s]) for s in np.linspace(-2.0, 2.0, 9)]
    res = {}
    for name, kw in (("seq", dict(batch=1)), ("setulb", dict(batch=4)), ("threads", dict(batch=4))):
        m = Stub()
        m.fit_driver = "threads" if name == "threads" else "auto"
        m._fit(9, "L-BFGS-B", False, False, {"maxiter": 200}, starts=starts, **kw)
        res[name] = (np.asarray(m.fit_nll_values_), m.best.copy(), getattr(m, "fit_driver_used_", None))
    assert res["setulb"][2] == "setulb" and res["threads"][2] == "threads"
    for name in ("setulb", "threads"):
        assert np.array_equal(res[name][0], res["seq"][0]) and np.array_equal(res[name][1], res["seq"][1])
