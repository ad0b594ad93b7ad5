"""Fixed cost of the GEMM (prologue + epilogue) from the K dependence of the run time: t(K) = a + b K."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip
ctx = _hip.default_context()
for (M, N) in ((19200, 500), (19200, 512), (16384, 512)):
    ts = {}
    for K in (128, 256, 512, 1024):
        ms, tf = ctx.gemm_bench(M, N, K, False, False, cfg=3, reps=10)
        ts[K] = ms * 1e3
    b = (ts[1024] - ts[512]) / 512.0
    a = ts[512] - 512 * b
    print("M=%d N=%d: " % (M, N) + "  ".join("K=%d %.1fus" % (k, v) for k, v in ts.items()) + "   fixed a=%.1f us, per-K b=%.4f us (MFMA-bound b would be %.4f)" % (a, b, 2.0 * M * N / 78.6e6), flush=True)
