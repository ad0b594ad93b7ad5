"""Run one GEMM configuration repeatedly (for rocprofv3 --pmc passes): python tools/gemm_one.py M N K ta tb cfg reps"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip
M, N, K, ta, tb, cfg, reps = (int(v) for v in sys.argv[1:8])
ctx = _hip.default_context()
print(ctx.gemm_bench(M, N, K, bool(ta), bool(tb), cfg=cfg, reps=reps))
