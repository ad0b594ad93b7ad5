#!/usr/bin/env python3
"""Device-to-host copy rate of the result path: gpcsd_fetch of a resident buffer into (a) a recycled pinned block
(PinnedPool) and (b) fresh pageable NumPy memory.  Run on the GPU box:  python tools/pcie_probe.py [MiB]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpcsd_amd import _hip

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib * (1 << 20) // 8
ctx = _hip.default_context()
ctx.hbm_copy_peak(n * 8)                                  # leaves a device buffer "peak_b" of n doubles
lib = ctx._lib
for label, pinned in (("pinned pool", True), ("pageable np.empty", False)):
    ts = []
    for rep in range(6):
        out = _hip.pinned_pool.empty((n,)) if pinned else np.empty(n)
        t0 = time.perf_counter()
        ctx._check(lib.gpcsd_fetch(ctx._h, b"peak_b", _hip._ptr(out), out.size))
        ts.append(time.perf_counter() - t0)
        del out
    print("%-20s %4d MiB: best %.2f ms = %.1f GB/s   (all: %s)" % (label, mib, 1e3 * min(ts), n * 8 / min(ts) / 1e9,
                                                                  " ".join("%.1f" % (1e3 * t) for t in ts)))
