"""Sweep the fp64 MFMA GEMM tile configurations over the shapes of the GPCSD hot path (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip
ctx = _hip.default_context()
shapes = [  # (name, M, N, K, transA, transB)
    ("proj_spatial  QsT Y", 384, 25000, 384, True, False),
    ("proj_temporal W Qt", 19200, 500, 500, False, False),
    ("pred_back     B QtT", 19200, 500, 500, False, True),
    ("Ks A Kgl", 384, 1200, 1200, False, False),
    ("Ks T At", 384, 384, 1200, False, True),
    ("wy_vz", 64, 500, 500, False, False),
    ("wy_update", 500, 500, 64, False, False),
    ("dc_merge", 500, 500, 500, False, False),
    ("1d proj_spatial", 24, 100000, 24, True, False),
    ("1d proj_temporal", 4800, 500, 500, False, False),
]
for name, M, N, K, ta, tb in shapes:
    row = []
    for cfg in (0, 1, 2, 3, 5):
        try:
            ms, tf = ctx.gemm_bench(M, N, K, ta, tb, cfg=cfg, reps=5)
            row.append("%d:%6.1fus/%5.1fTF" % (cfg, ms * 1e3, tf))
        except Exception as e:
            row.append("%d:ERR" % cfg)
    print("%-22s %6dx%6dx%5d  " % (name, M, N, K) + "  ".join(row), flush=True)
