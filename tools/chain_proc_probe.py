"""One context evaluating loglik at the cfg3 geometry with one resident trial in a loop for a fixed wall time; start several of
these PROCESSES side by side to see whether independent chains overlap across processes (tools/chain_concurrency_probe2.py asks
the same of threads in one process)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
w = bench.workload("cfg3")
m = bench.build_model(w, np.random.RandomState(0).standard_normal((w["nx"], w["nt"], 1)))
ctx = m._sync_device()
ctx.decomposition_cache(False)
hp, keep = m._hparams(m.JITTER)
for _ in range(20):
    ctx.loglik_parts(hp)
t_end = float(sys.argv[1])            # absolute time.time() at which to stop
t_start = float(sys.argv[2])
while time.time() < t_start:
    pass
n = 0
t0 = time.perf_counter()
while time.time() < t_end:
    ctx.loglik_parts(hp)
    n += 1
dt = time.perf_counter() - t0
print("pid %d: %.3f ms per loglik (%d calls)" % (os.getpid(), 1e3 * dt / n, n), flush=True)
