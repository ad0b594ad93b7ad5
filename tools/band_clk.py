"""Phase clocks of a -DBT_PHASE_CLK build of the band tail (tuning aid): python tools/band_clk.py [n ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpcsd_amd import _hip
ctx = _hip.Context()
for n in [int(a) for a in sys.argv[1:]] or [188]:
    t = np.arange(n) * 0.4
    K = 0.5 * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 20.0 ** 2) + 0.7 * np.exp(-np.abs(t[:, None] - t[None, :]) / 5.0)
    K = K / np.max(np.abs(K))
    ctx.debug_sybrd(K)
    band, V, tau = ctx.debug_sybrd(K)
    ph = [band[4, n - 4], band[4, n - 3], band[4, n - 2], band[4, n - 1], band[3, n - 3], band[3, n - 2]]
    print("n=%d (us): to A %.1f  X %.1f  H/M %.1f  Z %.1f  update %.1f | wave 0 QR %.1f | sum %.1f" % tuple([n] + [0.01 * v for v in ph] + [0.01 * sum(ph[:5])]))
    print("      QR split: park + rows + pending update %.1f | house4 %.1f | rest %.1f" % (0.01 * band[3, n - 1], 0.01 * band[2, n - 2], 0.01 * (ph[5] - band[3, n - 1] - band[2, n - 2])))
