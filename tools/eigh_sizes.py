"""Time gpcsd_eigh (whole symmetric eigen-decomposition, the `np.linalg.eigh` of utility_functions.py:58-59) at sizes past the
single-workgroup tail, with the event-scope split of one call.  usage: python tools/eigh_sizes.py [n ...]"""
import sys, time, json
import numpy as np
sys.path.insert(0, ".")
from gpcsd_amd import _hip

ns = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096]
ctx = _hip.default_context()
rng = np.random.default_rng(0)
for n in ns:
    t = np.linspace(0.0, 1.0, n)
    A = np.exp(-0.5 * ((t[:, None] - t[None, :]) / 0.05) ** 2) + 1e-3 * np.exp(-np.abs(t[:, None] - t[None, :]) / 0.01)
    for _ in range(3):                 # eager (allocates), captured + instantiated as a hipGraph, first replay
        ctx.eigh(A)
    ctx.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        w, V = ctx.eigh(A)
    ms = (time.perf_counter() - t0) / reps * 1e3
    ctx.prof_reset(); ctx.prof_enable(1)
    ctx.eigh(A)
    ctx.prof_enable(0)
    pr = {k: round(v["ms"], 3) for k, v in ctx.prof_all().items() if v["count"] and v["ms"] > 0.02}
    err = np.abs(V @ (w[:, None] * V.T) - A).max()
    wl = np.linalg.eigvalsh(A)
    print(json.dumps({"n": n, "ms_incl_copies": round(ms, 3), "recon_err": float(err), "eval_err": float(np.abs(np.sort(w) - wl).max()), "scopes": pr}), flush=True)
