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

    def step():
        ctx.loglik_parts(hp)
        ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)

    for _ in range(setup):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    th = 0.0
    for _ in range(steps):
        a = time.perf_counter()
        ctx.loglik_parts(hp)
        b = time.perf_counter()
        ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
        th += time.perf_counter() - b
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("steps %d  %.3f ms/step  host time inside predict_resident %.3f ms/step" % (steps, 1e3 * dt / steps, 1e3 * th / steps))


if __name__ == "__main__":
    main()
