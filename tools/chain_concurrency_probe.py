"""Do independent eigensolver chains overlap on one GPU?  k contexts (one host thread each) solve the same 250 x 250 problem
in a loop; compare with one context solving k replicas in ONE chain (gpcsd_eigh_batch).   GPCSD_NO_GRAPH=1 for eager launches."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 250
t = np.arange(n, dtype=np.float64)[:, None]
A = np.exp(-0.5 * ((t - t.T) / 9.0) ** 2) + 0.3 * np.exp(-np.abs(t - t.T) / 4.0)
reps = 60
for k in (1, 2, 3, 4):
    ctxs = [_hip.Context(None) for _ in range(k)]
    for c in ctxs:
        for _ in range(4):
            c.eigh(A)

    def run(c):
        for _ in range(reps):
            c.eigh(A)
    ths = [threading.Thread(target=run, args=(c,)) for c in ctxs]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t0
    print("%d contexts x eigh(%d): %.3f ms per solve per context, %.0f solves/s in total" % (k, n, 1e3 * dt / reps, k * reps / dt), flush=True)
    del ctxs
c = _hip.Context(None)
for k in (1, 2, 4, 8):
    stack = np.ascontiguousarray(np.stack([A] * k))
    for _ in range(4):
        c.eigh_batch(stack)
    t0 = time.perf_counter()
    for _ in range(reps):
        c.eigh_batch(stack)
    dt = time.perf_counter() - t0
    print("1 context, eigh_batch of %d: %.3f ms per call, %.0f solves/s" % (k, 1e3 * dt / reps, k * reps / dt), flush=True)
