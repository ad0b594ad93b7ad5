"""Tile configurations at the FOLDED shapes of a paired step (K = 192 / 250), R = 50 and R = 400 trials (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip
ctx = _hip.default_context()
shapes = []
for R in (50, 400):
    shapes += [("R=%d  W = Us^T Y  " % R, 192, 250 * R * 2, 192, True, False),
               ("R=%d  (W V), quad" % R, 192 * R * 2, 250, 250, False, False),
               ("R=%d  B = Ps a   " % R, 192, 250 * R * 2, 192, False, False),
               ("R=%d  S Pcat^T   " % R, 192 * R * 2, 500, 250, False, True)]
for name, M, N, K, ta, tb in shapes:
    row = []
    for cfg in (1, 2, 3):
        ms, tf = ctx.gemm_bench(M, N, K, ta, tb, cfg=cfg, reps=20)
        row.append("%d:%6.1fus/%5.1fTF" % (cfg, ms * 1e3, tf))
    print("%-20s %6dx%6dx%4d  " % (name, M, N, K) + "  ".join(row), flush=True)
