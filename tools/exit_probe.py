"""Where does the spatial tail stop (rank-revealing early exit)?  Reads e of replica 0 of the symmetric spatial class out of its arena
after a paired cfg3 step, with the spatial side shared (one replica) and not (two)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                     # noqa: E402
from gpcsd_amd import _hip                      # noqa: E402

w = bench.workload("cfg3")
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
lfp = bench.synth_data(w, m, 16, seed=11)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
ctx.decomposition_cache(False)
z = np.ascontiguousarray(w["x"])
hp, k1 = m._hparams(m.JITTER)
hp0, k0 = m._hparams(0.0)
n = 192
even = lambda x: (x + 1) & ~1
o = 0
offs = {}
for name, sz in (("A0", n * n), ("A1", n * n), ("V", (n + 64) * n), ("tau", n + 64 + 2), ("d", n), ("e", n)):
    offs[name] = o
    o += even(sz)
for share in (False, True):
    ctx.pair_share_s(share)
    for _ in range(3):
        ctx.loglik_predict_async(hp, hp0, z, w["t"], _hip.PRED_CSD, want_lists=True)
        ctx.loglik_parts_wait()
    ctx.synchronize()
    for tag in ("p0s", "p0a"):
        a = ctx.fetch("eig_%s_arena" % tag, (offs["e"] + n,))
        e = a[offs["e"]:offs["e"] + n]
        tau = a[offs["tau"]:offs["tau"] + n]
        nz = np.nonzero(e)[0]
        print("shared" if share else "two replicas", tag, "last nonzero e at", nz.max() if nz.size else -1, "nonzero tau:", int(np.count_nonzero(tau)))
