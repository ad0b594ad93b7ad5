"""Timing of the blocked Cholesky (chol.hip) on the GPU box: ms and TF/s (n^3/3 flops) per order, the scope split, and the
A/B knobs GPCSD_POTRF_LOOKAHEAD=0|1, GPCSD_POTRF_TCFG=0|1|2|3 (tile configuration of the trailing updates).
    python tools/potrf_probe.py [n ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip

ns = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192, 12000]
ctx = _hip.default_context()
out = {"lookahead": os.environ.get("GPCSD_POTRF_LOOKAHEAD", "1"), "tcfg": os.environ.get("GPCSD_POTRF_TCFG", "0"), "runs": []}
for n in ns:
    ms, tf = ctx.potrf_bench(n, reps=3)
    ctx.prof_reset()
    ctx.prof_enable(1)
    ctx.potrf_bench(n, reps=1)
    ctx.prof_enable(0)
    prof = {k: v for k, v in ctx.prof_all().items() if k.startswith("potrf") and v["count"]}
    split = {k: {"ms": v["ms"] / 2.0, "launches": v["count"] // 2, "tflops": (v["flops"] / 2.0) / (v["ms"] / 2.0 * 1e-3) / 1e12 if v["ms"] else None}
             for k, v in prof.items()}       # (the profiled call runs the factorisation twice: warm-up + 1)
    out["runs"].append({"n": n, "ms": ms, "tflops": tf, "frac_of_78.6": tf / 78.6, "scopes": split})
    print("n=%6d  %8.3f ms  %6.2f TF/s  (%.3f of peak)" % (n, ms, tf, tf / 78.6), file=sys.stderr)
    for k, v in sorted(split.items(), key=lambda kv: -kv[1]["ms"]):
        print("      %-24s %8.3f ms  %5d launches  %s" % (k, v["ms"], v["launches"], "%.1f TF/s" % v["tflops"] if v["tflops"] else ""), file=sys.stderr)
out["diag128_phases_us"] = ctx.potrf_diag_probe()
print("diag128 phases (us):", {k: round(v, 2) for k, v in out["diag128_phases_us"].items()}, file=sys.stderr)
print(json.dumps(out))
