"""Time one loglik+gradient evaluation (the unit of work of fit) at the bench geometry, with the per-kernel profile."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
w = bench.workload(name)
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
for _ in range(3):
    m._loglik_and_grad_natural()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    ll, g = m._loglik_and_grad_natural()
dt = (time.perf_counter() - t0) / n
print("%s: loglik+grad %.3f ms per evaluation (loglik alone: see bench)" % (name, dt * 1e3))
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(3):
    m._loglik_and_grad_natural()
ctx.prof_enable(False)
prof = ctx.prof_all()
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    if v["count"]:
        print("  %-28s %7.3f ms/eval  n=%d  %6.1f TF/s" % (k, v["ms"] / 3, v["count"] // 3, (v["flops"] / max(v["ms"], 1e-9)) / 1e9))
