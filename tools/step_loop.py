"""Run a few steady-state steps (loglik -> predict_resident) of a bench workload and nothing else: the process to put under
`rocprofv3 --kernel-trace` when a timeline of the step is wanted (tools/timeline.py reads the trace).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/step_loop.py cfg3 8
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                     # noqa: E402
from gpcsd_amd import _hip                      # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    setup = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    w = bench.workload(name)
    m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
    lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1000)
    m.update_lfp(lfp, w["t"])
    ctx = m._sync_device()
    ctx.decomposition_cache(False)
    z = np.ascontiguousarray(w["x"])
    hp, keep = m._hparams(m.JITTER)
    hp0, keep0 = m._hparams(0.0)

    sync_ll = os.environ.get("STEP_SYNC_LOGLIK") == "1"

    paired = os.environ.get("STEP_PAIRED", "1") == "1"

    def step():
        if paired and not sync_ll:
            ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            if os.environ.get("STEP_PREFETCH") == "1":
                ctx.prefetch_pair(hp, hp0, z, w["t"])
            ctx.loglik_parts_wait()
        elif sync_ll:
            ctx.loglik_parts(hp)
            ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
        else:
            ctx.loglik_parts_async(hp)
            ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            ctx.loglik_parts_wait()

    depth = int(os.environ.get("STEP_DEPTH", "1"))        # 2: queue step k+1 before collecting step k's log-likelihood
    if depth == 2 and paired and not sync_ll:
        def run(n):
            ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
            for _ in range(n - 1):
                ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
                ctx.loglik_parts_wait()
            ctx.loglik_parts_wait()
        run(setup)
        ctx.synchronize()
        t0 = time.perf_counter()
        run(steps)
        ctx.synchronize()
        print("steps %d  %.3f ms/step  (paired, host loop two steps deep)" % (steps, 1e3 * (time.perf_counter() - t0) / steps))
        return
    for _ in range(setup):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("steps %d  %.3f ms/step  (%s loglik)" % (steps, 1e3 * dt / steps, "synchronous" if sync_ll else ("paired" if paired else "asynchronous")))


if __name__ == "__main__":
    main()
