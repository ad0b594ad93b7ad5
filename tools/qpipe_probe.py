"""Round 5: which calls take the pipelined stage 5 (gpcsd_q_pipeline's counter after each call form)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                     # noqa: E402
from gpcsd_amd import _hip                      # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
w = bench.workload("cfg3")
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
lfp = bench.synth_data(w, m, R, seed=11)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
ctx.decomposition_cache(False)
z = w["x"]
hp, k1 = m._hparams(m.JITTER)
hp0, k0 = m._hparams(0.0)
for rep in range(3):
    n = ctx.q_pipeline()
    ctx.loglik_parts(hp)
    n1 = ctx.q_pipeline()
    ctx.predict_resident(hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
    ctx.synchronize()
    n2 = ctx.q_pipeline()
    ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
    ctx.loglik_parts_wait()
    ctx.synchronize()
    n3 = ctx.q_pipeline()
    print("rep", rep, "loglik_parts +%d, predict_resident +%d, paired +%d" % (n1 - n, n2 - n1, n3 - n2), flush=True)

if os.environ.get("GPCSD_QPIPE_CLK") == "1":
    # the gates of the LAST paired step: when each started to wait, when it passed, against the start of the tail it waited for
    for _ in range(10):
        ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
        ctx.loglik_parts_wait()
    ctx.synchronize()
    raw = ctx.fetch("wy_clk", (64,)).view(np.uint64)
    for par in (0, 1):
        st = [int(v) for v in raw[48 + 8 * par: 56 + 8 * par]]
        print("panel parity %d, workgroup (0, 0): gate passed -> G %.1f | diagonal blocks %.1f | doubling + store %.1f | slab load %.1f | W1, W2 %.1f | "
              "update %.1f | store %.1f us" % tuple([par] + [0.01 * (st[k + 1] - st[k]) for k in range(7)]))
    clk = raw[:48].reshape(-1, 3)
    for p in range(4):
        for y in range(2):
            e, x, t0 = [int(v) for v in clk[p * 4 + y]]
            if e:
                print("panel %d problem %d: gate entered %+8.1f us, passed %+8.1f us after the start of its tail" % (p, y, 0.01 * (e - t0), 0.01 * (x - t0)))
