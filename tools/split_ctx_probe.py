"""Timing probe: loglik on one context, predict on another (same data on both), queued together every step.  Tells whether two
temporal eigen-chains overlap when nothing orders them (they share no stream and no workspace here)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                     # noqa: E402
from gpcsd_amd import _hip                      # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    w = bench.workload("cfg3")
    ms = []
    for i in range(2):
        m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
        lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1000)
        m.update_lfp(lfp, w["t"])
        ctx = m._sync_device()
        ctx.decomposition_cache(False)
        ms.append((m, ctx))
    z = np.ascontiguousarray(w["x"])
    (ma, ca), (mb, cb) = ms
    hp, keep = ma._hparams(ma.JITTER)
    hp0, keep0 = mb._hparams(0.0)

    def step():
        ca.loglik_parts_async(hp)
        cb.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
        return ca.loglik_parts_wait()

    for _ in range(40):
        step()
    ca.synchronize(); cb.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ca.synchronize(); cb.synchronize()
    print("split contexts: %.3f ms per step" % (1e3 * (time.perf_counter() - t0) / steps))
    # each half alone, queued back to back (no fences inside the loop)
    t0 = time.perf_counter()
    for _ in range(steps):
        ca.loglik_parts_async(hp)
        ca.loglik_parts_wait()
    t1 = time.perf_counter()
    for _ in range(steps):
        cb.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
    cb.synchronize()
    t2 = time.perf_counter()
    print("loglik alone %.3f ms, predict alone (queued back to back) %.3f ms" % (1e3 * (t1 - t0) / steps, 1e3 * (t2 - t1) / steps))


if __name__ == "__main__":
    main()
