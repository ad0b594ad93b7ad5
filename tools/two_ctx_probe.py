"""How well do independent eigen-chains of several contexts overlap on one GPU?  k contexts (one process, one host thread), each
queues loglik_async + predict_resident per step, then all are collected.  Prints ms per step-of-all-contexts."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                     # noqa: E402
from gpcsd_amd import _hip                      # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    w = bench.workload("cfg3")
    ms = []
    for i in range(k):
        m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
        lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1000 + i)
        m.update_lfp(lfp, w["t"])
        ctx = m._sync_device()
        ctx.decomposition_cache(False)
        ms.append((m, ctx, m._hparams(m.JITTER), m._hparams(0.0)))
    z = np.ascontiguousarray(w["x"])
    print("contexts distinct:", len({id(c) for _, c, _, _ in ms}))

    def step():
        for m, ctx, (hp, _k), (hp0, _k0) in ms:
            ctx.loglik_parts_async(hp)
            ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
        for m, ctx, _a, _b in ms:
            ctx.loglik_parts_wait()

    for _ in range(40):
        step()
    for _, ctx, _a, _b in ms:
        ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for _, ctx, _a, _b in ms:
        ctx.synchronize()
    dt = time.perf_counter() - t0
    print("%d contexts: %.3f ms per round of %d steps = %.3f ms per step" % (k, 1e3 * dt / steps, k, 1e3 * dt / steps / k))


if __name__ == "__main__":
    main()
