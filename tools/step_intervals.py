#!/usr/bin/env python3
"""Intervals between consecutive launches of the once-per-step kernel over a whole rocprofv3 kernel trace (CSV): the mean per block
of 10 steps and every interval above a threshold, with what ran (and what did not) inside it.
    python tools/step_intervals.py <..._kernel_trace.csv> [threshold_ms]"""
import csv, sys
from collections import Counter
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
marks = [i for i, r in enumerate(rows) if "gemm_pred_unfold_kernel" in r[2]]
print("step launches:", len(marks))
iv = [(rows[marks[k + 1]][0] - rows[marks[k]][0]) / 1e6 for k in range(len(marks) - 1)]
for b in range(0, len(iv), 10):
    blk = iv[b:b + 10]
    print("steps %4d..%4d  mean %.4f  max %.4f" % (b, b + len(blk) - 1, sum(blk) / len(blk), max(blk)))
for k, v in enumerate(iv):
    if v > thr:
        i0, i1 = marks[k], marks[k + 1]
        t0 = rows[i0][0]
        print("\ninterval %d = %.3f ms: kernels inside" % (k, v))
        last_end = rows[i0][1]
        for s, e, n in rows[i0:i1 + 1]:
            gap = (s - last_end) / 1e3
            flag = "   <-- %.0f us with nothing running before this" % gap if gap > 200 else ""
            if flag or v < 5:
                print("  +%9.1f us %8.1f us  %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, n.split("(")[0][-60:], flag))
            last_end = max(last_end, e)
