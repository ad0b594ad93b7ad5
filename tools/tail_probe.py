"""Duration of the register tail alone (its own device-clock stamps) for whole problems of n rows: the strip phase costs what
T(n) - T(192 + block part) says.   python tools/tail_probe.py [n ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip  # noqa: E402

ctx = _hip.Context()
sizes = [int(a) for a in sys.argv[1:]] or [188, 192, 208, 224, 250, 256]
for n in sizes:
    t = np.arange(n) * 0.4
    K = 0.5 * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 20.0 ** 2) + 0.7 * np.exp(-np.abs(t[:, None] - t[None, :]) / 5.0)
    ctx.prof_enable(2)
    best = 1e9
    for _ in range(6):
        ctx.debug_sytrd(K)
        ms, nwg, fl = ctx.prof_tail_clock(2)
        if ms > 0:
            best = min(best, ms)
    ctx.prof_enable(0)
    print("n=%3d  tail %.1f us" % (n, 1e3 * best), flush=True)
