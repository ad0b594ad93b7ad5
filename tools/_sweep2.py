import sys, os
sys.path.insert(0, "/root/repo")
from gpcsd_amd import _hip
ctx = _hip.default_context()
shapes = [
    ("fold proj_spatial", 192, 25000, 192, True, False),
    ("fold temporal", 19200, 250, 250, False, False),
    ("fold tstar", 19200, 500, 250, False, False),
    ("fold cross", 192, 25000, 192, False, False),
    ("Ks A Kgl", 384, 1200, 1200, False, False),
    ("unfolded tstar", 19200, 1000, 500, False, False),
]
for name, M, N, K, ta, tb in shapes:
    row = []
    for cfg in (1, 3, 5):
        ms, tf = ctx.gemm_bench(M, N, K, ta, tb, cfg=cfg, reps=10)
        row.append("%d:%6.1fus/%5.1fTF" % (cfg, ms * 1e3, tf))
    print("%-22s %6dx%6dx%5d  " % (name, M, N, K) + "  ".join(row), flush=True)
