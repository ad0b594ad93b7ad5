import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip
ctx = _hip.default_context()
for cfg in (1, 3):
    for M in (8192, 16384, 19200, 24576, 32768, 65536):
        ms, tf = ctx.gemm_bench(M, 512, 512, False, False, cfg=cfg, reps=5)
        print("cfg %d M=%6d N=512 K=512: %7.1f us %5.1f TF" % (cfg, M, ms * 1e3, tf), flush=True)
