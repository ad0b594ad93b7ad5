#!/usr/bin/env python3
"""Evidence that a batched evaluation shares every launch: run N calls of gpcsd_loglik_grad_batch with B sets under
`rocprofv3 --kernel-trace --stats` and compare the number of kernel dispatches per call for B = 1 and B = 8.
    rocprofv3 --kernel-trace --stats -d out1 -o b1 --output-format csv -- python3 tools/launch_count.py 1
    rocprofv3 --kernel-trace --stats -d out8 -o b8 --output-format csv -- python3 tools/launch_count.py 8
    python3 tools/launch_count.py summarize out1/b1_kernel_stats.csv out8/b8_kernel_stats.csv
"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NCALLS = 30

if sys.argv[1] == "summarize":
    tot = []
    for f in sys.argv[2:4]:
        rows = list(csv.DictReader(open(f)))
        tot.append((sum(int(r["Calls"]) for r in rows), sum(float(r["TotalDurationNs"]) for r in rows)))
    print("kernel dispatches: B=1 %d   B=8 %d   (identical setup + %d calls each)" % (tot[0][0], tot[1][0], NCALLS))
    print("GPU time:          B=1 %.2f ms  B=8 %.2f ms" % (tot[0][1] / 1e6, tot[1][1] / 1e6))
    sys.exit(0)

import numpy as np
import bench
B = int(sys.argv[1])
w = bench.workload("cfg5")
m = bench.build_model(w, np.zeros((w["nx"], w["nt"], 1)))
lfp = bench.synth_data(w, m, w["trials_per_gpu"], seed=1000)
m.update_lfp(lfp, w["t"])
ctx = m._sync_device()
ng = 1 + m.dim + 2 * len(m.temporal_cov_list) + 1
sets = []
for k in range(8):                        # the same eight sets are BUILT either way; only B of them are evaluated per call
    np.random.seed(k)
    m._set_from_tparams(m._sample_start(False), False)
    sets.append(m._hparams(m.JITTER))
hps = [h for h, _ in sets][:B]
for _ in range(NCALLS):
    ctx.loglik_grad_batch(hps, ng)
ctx.synchronize()
