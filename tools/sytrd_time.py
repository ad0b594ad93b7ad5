#!/usr/bin/env python3
"""Kernel time of the single-workgroup tridiagonalisation tail for a few orders (HIP events around the launch, through the
library's profiler): python tools/sytrd_time.py [lib.so ...]   -- each library in its own subprocess (GPCSD_LIB_PATH)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("_SYTRD_CHILD"):
    sys.path.insert(0, ROOT)
    from gpcsd_amd import _hip
    ctx = _hip.default_context()
    out = []
    for n in (192, 224, 250):
        t = np.arange(n, dtype=np.float64)[:, None]
        A = np.exp(-0.5 * ((t - t.T) / 9.0) ** 2) + 0.3 * np.exp(-np.abs(t - t.T) / 4.0)
        for _ in range(5):
            ctx.debug_sytrd(A)
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(40):
            ctx.debug_sytrd(A)
        ctx.prof_enable(False)
        p = ctx.prof_get("sytrd_rtail")
        out.append("n=%d %.1f us" % (n, 1e3 * p["ms"] / p["count"]))
    print(os.path.basename(os.environ.get("GPCSD_LIB_PATH", "default")), "  ".join(out), flush=True)
else:
    for lib in sys.argv[1:] or [""]:
        env = dict(os.environ, _SYTRD_CHILD="1")
        if lib:
            env["GPCSD_LIB_PATH"] = os.path.abspath(lib)
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=env)
