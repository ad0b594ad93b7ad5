"""Per-scope profile of one batched objective+gradient evaluation (gpcsd_loglik_grad_batch) at a bench geometry:
    python tools/grad_batch_timing.py cfg5 32"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = bench.workload(name)
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
ng = 1 + m.dim + 2 * len(m.temporal_cov_list) + 1
sets = []
for k in range(B):
    np.random.seed(k)
    m._set_from_tparams(m._sample_start(False), False)
    sets.append(m._hparams(m.JITTER))
hps = [h for h, _ in sets]
for _ in range(4):
    ctx.loglik_grad_batch(hps, ng)
t0 = time.perf_counter()
n = 20
for _ in range(n):
    ctx.loglik_grad_batch(hps, ng)
dt = (time.perf_counter() - t0) / n
print("%s B=%d: %.3f ms per batched evaluation = %.0f evals/s" % (name, B, dt * 1e3, B / dt))
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(3):
    ctx.loglik_grad_batch(hps, ng)
ctx.prof_enable(False)
prof = ctx.prof_all()
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    if v["count"]:
        print("  %-28s %7.3f ms/eval  n=%d  %6.1f TF/s" % (k, v["ms"] / 3, v["count"] // 3, (v["flops"] / max(v["ms"], 1e-9)) / 1e9))
