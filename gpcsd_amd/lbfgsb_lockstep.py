"""Lock-step L-BFGS-B: many independent minimisations advanced by ONE driver, one batched evaluation per tick.

fit() runs independent L-BFGS-B restarts (reference: gpcsd1d.py:193-220, gpcsd2d.py:230-262, `scipy.optimize.minimize(...,
method='L-BFGS-B', jac=..., bounds=...)`).  On the GPU k evaluations submitted together cost about what one costs
(gpcsd_loglik_grad_batch), so the restarts advance together: every live chain is at the point where L-BFGS-B asks for an
objective + gradient evaluation, ONE batched device call serves them all, and each chain then runs on to its next request.

The optimiser is SciPy's own L-BFGS-B (`scipy.optimize._lbfgsb.setulb`, the compiled Byrd-Lu-Nocedal-Zhu code) driven through
its reverse-communication interface exactly as `scipy.optimize.minimize` drives it -- same workspace, same `task` protocol,
same iteration / evaluation limits, same termination messages -- so a chain walks, bit for bit, the trajectory the reference's
`minimize` call walks from the same start on the same objective values.  What changes is who waits for whom: round 2 ran one
unmodified `minimize` per Python thread and let the threads rendezvous in their objective callbacks (gpcsd_amd/lockstep.py:
two condition-variable hand-offs and a GIL switch per chain and evaluation, about half of cfg5's wall time); here a single
thread steps B optimiser states (about 10 us each) between device calls.  `available()` is False when this SciPy does not have
the interface in the form used here (private module): fit() then falls back to the threaded evaluator.
"""
import numpy as np

try:                                          # private SciPy interface, probed once (see available())
    from scipy.optimize import _lbfgsb as _slb
    from scipy.optimize._lbfgsb_py import status_messages as _STATUS, task_messages as _TASKMSG
except Exception:                             # pragma: no cover - depends on the SciPy build
    _slb = None
    _STATUS, _TASKMSG = {}, {}

_TASK_START, _TASK_NEW_X, _TASK_FG, _TASK_CONVERGENCE, _TASK_STOP = 0, 1, 3, 4, 5


_probed = None                                # available(): None = not probed yet


def _probe_problem():
    """A small bounded problem whose L-BFGS-B run takes several iterations, a line search with more than one evaluation and
    an active bound: what a change of SciPy's `task` codes or argument semantics would disturb."""
    A = np.array([[3.0, 0.6, 0.0], [0.6, 2.0, -0.4], [0.0, -0.4, 1.5]])
    b = np.array([1.0, -2.0, 0.5])

    def fg(x):
        r = A @ x - b
        f = 0.5 * float(r @ r) + 0.1 * float(np.sum(x ** 4))
        return f, A.T @ r + 0.4 * x ** 3
    return fg, np.array([2.0, 1.5, 1.0]), [(-5.0, 5.0), (None, None), (0.3, None)], {"maxiter": 50}


def available():
    """True when SciPy's compiled L-BFGS-B exposes the reverse-communication entry point AND stepping it through this driver
    reproduces `scipy.optimize.minimize(method="L-BFGS-B")` on a probe problem exactly -- x, fun, message and the iteration /
    evaluation counts.  A SciPy that keeps the private name but changes the protocol fails the comparison and fit() falls back
    (with a RuntimeWarning) to the thread rendezvous around unmodified minimize() calls."""
    global _probed
    if _probed is None:
        _probed = False
        if _slb is not None and hasattr(_slb, "setulb"):
            try:
                from scipy.optimize import minimize
                fg, x0, bounds, opts = _probe_problem()
                ref = minimize(fg, x0, jac=True, method="L-BFGS-B", bounds=bounds, options=opts)
                out, stats = minimize_many(lambda items: {k: fg(x) for k, x in items}, [x0], bounds, opts, width=1)
                got = out[0]
                msg = ref.message if isinstance(ref.message, str) else ref.message.decode()
                _probed = (not isinstance(got, Exception) and got[0] == ref.fun and np.array_equal(got[1], ref.x)
                           and got[2] == msg and stats["points"] == ref.nfev)
            except Exception:
                _probed = False
    return _probed


class _Chain:
    """One L-BFGS-B state: the arrays `_minimize_lbfgsb` allocates (scipy/optimize/_lbfgsb_py.py) and its loop, cut at FG."""

    __slots__ = ("key", "x", "f", "g", "wa", "iwa", "task", "ln_task", "lsave", "isave", "dsave", "nit", "nfev", "done", "failed")

    def __init__(self, key, x0, n, m):
        self.key = key
        self.x = np.array(x0, dtype=np.float64)
        self.f = np.array(0.0, dtype=np.int32)          # (as SciPy initialises it: replaced by the first evaluation)
        self.g = np.zeros((n,), dtype=np.float64)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29, dtype=np.float64)
        self.nit = 0
        self.nfev = 0
        self.done = False
        self.failed = None


def minimize_many(batch_fn, starts, bounds, options, width, on_error=(ValueError, np.linalg.LinAlgError)):
    """Minimise from every start in `starts` with L-BFGS-B, at most `width` chains alive at a time.

    batch_fn([(key, x), ...]) -> {key: (f, g) or an Exception instance}: the batched objective + gradient (keys are indices
    into `starts`; the list is sorted by key, so slot b of a batch means the same restart on every rank of a sharded fit).
    bounds: [(lo, hi)] with None / +-inf for an open side, as scipy.optimize.minimize takes them.
    options: SciPy's L-BFGS-B options (maxcor, ftol, gtol, maxfun, maxiter, maxls; disp / iprint are ignored).
    An exception of a type in `on_error` returned for a point ends that chain alone (its result is the exception instance);
    any other exception is raised.  Returns ({index: (fun, x, message) or Exception}, stats) with stats = {"batches",
    "points"}."""
    if _slb is None:
        raise RuntimeError("scipy.optimize._lbfgsb is not importable")
    opts = dict(options or {})
    m = int(opts.get("maxcor", 10))
    ftol = opts.get("ftol", 2.2204460492503131e-09)
    pgtol = opts.get("gtol", 1e-5)
    maxfun = opts.get("maxfun", 15000)
    maxiter = opts.get("maxiter", 15000)
    maxls = int(opts.get("maxls", 20))
    if not maxls > 0:
        raise ValueError("maxls must be positive.")
    factr = ftol / np.finfo(float).eps
    starts = [np.asarray(s0, dtype=np.float64).ravel() for s0 in starts]
    if not starts:
        return {}, {"batches": 0, "points": 0}
    n = starts[0].size
    if len(bounds) != n:
        raise ValueError("length of x0 != length of bounds")
    lo = np.array([-np.inf if b[0] is None else b[0] for b in bounds], dtype=np.float64)
    hi = np.array([np.inf if b[1] is None else b[1] for b in bounds], dtype=np.float64)
    if (lo > hi).any():
        raise ValueError("LBFGSB - one of the lower bounds is greater than an upper bound.")
    nbd = np.zeros(n, np.int32)
    low_bnd = np.zeros(n, np.float64)
    upper_bnd = np.zeros(n, np.float64)
    for i in range(n):                              # (the encoding of _minimize_lbfgsb: 0 free, 1 lower, 2 both, 3 upper)
        has_l, has_u = not np.isinf(lo[i]), not np.isinf(hi[i])
        if has_l:
            low_bnd[i] = lo[i]
        if has_u:
            upper_bnd[i] = hi[i]
        nbd[i] = (1 if has_l and not has_u else 2 if has_l and has_u else 3 if has_u else 0)
    setulb = _slb.setulb

    def advance(ch):
        """Run the chain to its next evaluation request (returns True) or to its end (False): the body of SciPy's loop."""
        task = ch.task
        while True:
            setulb(m, ch.x, low_bnd, upper_bnd, nbd, ch.f, ch.g, factr, pgtol, ch.wa, ch.iwa, task, ch.lsave, ch.isave, ch.dsave,
                   maxls, ch.ln_task)
            t0 = task[0]
            if t0 == _TASK_FG:
                return True
            if t0 == _TASK_NEW_X:
                ch.nit += 1
                if ch.nit >= maxiter:
                    task[0], task[1] = _TASK_STOP, 504
                elif ch.nfev > maxfun:
                    task[0], task[1] = _TASK_STOP, 502
                continue
            ch.done = True
            return False

    out = {}
    pending = list(range(len(starts)))[::-1]            # pop() hands the restarts out in order
    live = []
    width = max(1, min(int(width), len(starts)))
    stats = {"batches": 0, "points": 0}

    def refill():
        while pending and len(live) < width:
            k = pending.pop()
            ch = _Chain(k, np.clip(starts[k], lo, hi), n, m)         # the start is moved into the box, as minimize() does
            if advance(ch):
                live.append(ch)
            else:                                                    # (cannot happen: START always asks for an evaluation)
                out[k] = (float(ch.f), ch.x.copy(), _message(ch))

    refill()
    while live:
        live.sort(key=lambda ch: ch.key)
        res = batch_fn([(ch.key, ch.x.copy()) for ch in live])
        stats["batches"] += 1
        stats["points"] += len(live)
        still = []
        for ch in live:
            r = res[ch.key]
            if isinstance(r, Exception):
                if isinstance(r, on_error):
                    out[ch.key] = r
                    continue
                raise r
            ch.f, ch.g = r[0], np.asarray(r[1], dtype=np.float64)
            ch.nfev += 1
            if advance(ch):
                still.append(ch)
            else:
                out[ch.key] = (ch.f, ch.x.copy(), _message(ch))
        live = still
        refill()
    return out, stats


def _message(ch):
    return _STATUS.get(int(ch.task[0]), "") + ": " + _TASKMSG.get(int(ch.task[1]), "")
