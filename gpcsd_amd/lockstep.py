"""Lock-step evaluation of independent optimiser chains.

fit() runs independent L-BFGS-B restarts (reference: gpcsd1d.py:193-220, gpcsd2d.py:230-262) and the trial-shift fits of
auditory_lfp/fit_mean_function.py:323-328 run one L-BFGS-B per trial.  Each chain asks for one objective evaluation at a
time, and on the GPU one evaluation is a latency-bound chain of small launches -- but k evaluations submitted together cost
about the same latency as one (gpcsd_loglik_grad_batch, gpcsd_whitened_quad).  The chains stay unmodified SciPy optimisers,
each on its own thread; their objective callbacks rendezvous here, and when every live chain has submitted its point ONE
batched evaluation serves them all.
"""
import threading


class LockstepEvaluator:
    def __init__(self, batch_fn):
        """batch_fn(list of (key, x)) -> dict key -> result (a result that is an Exception instance is raised in the chain
        that asked for it)."""
        self._fn = batch_fn
        self._cond = threading.Condition()
        self._pending = {}
        self._results = {}
        self._live = 0
        self.batches = 0            # number of batched evaluations issued
        self.points = 0             # number of points they covered

    def join(self):
        with self._cond:
            self._live += 1

    def leave(self):
        with self._cond:
            self._live -= 1
            self._fire_if_complete()

    def _fire_if_complete(self):
        # called with the lock held; the chains that submitted are all blocked in evaluate(), so running the batch here --
        # on whichever thread completed the rendezvous -- is single-threaded with respect to the model state
        if self._pending and len(self._pending) >= self._live:
            # sorted by key: the SET of keys in a batch is deterministic, the order in which the optimiser threads arrived is
            # not -- and under trial sharding every rank all-reduces the batch element-wise, so slot b must mean the same
            # restart on every rank (and a given restart's position must not depend on thread timing)
            items = sorted(self._pending.items(), key=lambda kv: kv[0])
            self._pending = {}
            try:
                res = self._fn(items)
            except Exception as e:           # a failure of the batch as a whole goes to every chain
                res = {k: e for k, _ in items}
            self.batches += 1
            self.points += len(items)
            self._results.update(res)
            self._cond.notify_all()

    def evaluate(self, key, x):
        with self._cond:
            self._pending[key] = x
            self._fire_if_complete()
            while key not in self._results:
                self._cond.wait()
            r = self._results.pop(key)
        if isinstance(r, Exception):
            raise r
        return r


def run_chains(jobs, chain_fn, batch_fn, width):
    """Run chain_fn(job, evaluate) for every job with at most `width` chains alive; evaluate(x) blocks until the batched
    evaluation that includes x is done.  Returns ({job index: chain result}, evaluator); a chain that raises has the
    exception as its result."""
    import queue
    ev = LockstepEvaluator(batch_fn)
    q = queue.Queue()
    for i, job in enumerate(jobs):
        q.put((i, job))
    out = {}
    nworkers = max(1, min(int(width), len(jobs)))
    for _ in range(nworkers):        # every worker counts as live from the start, so the first batch waits for all of them
        ev.join()

    def worker():
        try:
            while True:
                try:
                    i, job = q.get_nowait()
                except queue.Empty:
                    return
                try:
                    out[i] = chain_fn(job, lambda x, _i=i: ev.evaluate(_i, x))
                except Exception as e:
                    out[i] = e
        finally:
            ev.leave()
    threads = [threading.Thread(target=worker, daemon=True) for _ in range(nworkers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return out, ev
