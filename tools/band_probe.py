"""Round 5: the band tail (sytrd_bandtail.hpp) and the banded consumers (band.hip) against NumPy, and the fused calls with the band
form on / off.  python tools/band_probe.py [sizes...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpcsd_amd import _hip  # noqa: E402
import band_prototype as bp  # noqa: E402


def check_sybrd(ctx, A, tag):
    n = A.shape[0]
    band, V, tau = ctx.debug_sybrd(A)
    Q = bp.q_of(V, tau)
    B = bp.band_to_dense(band)
    sc = np.max(np.abs(A))
    err = np.max(np.abs(Q.T @ A @ Q - B)) / sc
    orth = np.max(np.abs(Q.T @ Q - np.eye(n)))
    ev = np.max(np.abs(np.linalg.eigvalsh(B) - np.linalg.eigvalsh(A))) / sc
    print("%-28s n=%3d  |Q^T A Q - B| %.1e  orth %.1e  spectrum %.1e" % (tag, n, err, orth, ev))
    return max(err, orth, ev)


def main():
    ctx = _hip.Context()
    rs = np.random.RandomState(0)
    sizes = [int(a) for a in sys.argv[1:]] or [9, 12, 13, 23, 64, 100, 188, 190, 192, 193, 196, 200, 231, 249, 250]
    worst = 0.0
    for n in sizes:
        t = np.arange(n) * 0.4
        K = 0.5 * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 20.0 ** 2) + 0.7 * np.exp(-np.abs(t[:, None] - t[None, :]) / 5.0)
        G = rs.standard_normal((n, n))
        worst = max(worst, check_sybrd(ctx, K, "SE + Matern Gram"), check_sybrd(ctx, G + G.T, "random symmetric"))
        if n >= 64:
            worst = max(worst, check_sybrd(ctx, np.ones((n, n)) + 1e-3 * np.eye(n), "rank one + small shift"))
    print("worst", worst)
    if os.environ.get("BT_PHASE_CLK") == "1":                  # a -DBT_PHASE_CLK build leaves its phase clocks in the band's unused corner
        for n in (188, 250):
            t = np.arange(n) * 0.4
            K = 0.5 * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 20.0 ** 2) + 0.7 * np.exp(-np.abs(t[:, None] - t[None, :]) / 5.0)
            K = K / np.max(np.abs(K))
            band, V, tau = ctx.debug_sybrd(K)
            ph = [band[4, n - 4], band[4, n - 3], band[4, n - 2], band[4, n - 1], band[3, n - 3], band[3, n - 2]]
            print("n=%d phase clocks (us): to A %.1f  X %.1f  H/M %.1f  Z %.1f  update %.1f | wave 0's own factorisations %.1f" %
                  tuple([n] + [0.01 * v for v in ph]))
    # timing of the tail alone (fenced debug call: upload + kernel + download; take the min of a few)
    for n in (188, 250):
        t = np.arange(n) * 0.4
        K = 0.5 * np.exp(-0.5 * (t[:, None] - t[None, :]) ** 2 / 20.0 ** 2) + 0.7 * np.exp(-np.abs(t[:, None] - t[None, :]) / 5.0)
        for fn, nm in ((ctx.debug_sybrd, "band"), (ctx.debug_sytrd, "tridiagonal")):
            fn(K)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                fn(K)
                best = min(best, time.perf_counter() - t0)
            print("debug call n=%d %-12s %.3f ms (fenced, with transfers)" % (n, nm, 1e3 * best))


if __name__ == "__main__":
    main()
