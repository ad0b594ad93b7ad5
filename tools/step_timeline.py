#!/usr/bin/env python3
"""One steady-state step of `bench.py --only-value` out of a rocprofv3 kernel trace (CSV): per stream, which kernels ran when.

    python tools/step_timeline.py <..._kernel_trace.csv> [--step-kernel unfold_swap_sum_kernel] [--from-end 10]

A step is delimited by two consecutive launches of a kernel that runs exactly once per step (the final relayout pass of
predict); the step `--from-end` steps before the end of the trace is printed: start offset and duration of every kernel that
STARTS inside the window, grouped by stream (rocprofv3's Stream_Id, falling back to Queue_Id), then per stream its busy time
and the span from its first start to its last end.  Because consecutive steps overlap (the next step's chains run beside this
step's predict GEMMs) the window holds the tail of one step's predict and the chains of the next -- exactly one step's worth of
every kernel.  Also prints, over the last 50 steps, the mean interval between the delimiting launches (= ms per step under the
profiler) and the mean busy time per stream per step."""
import argparse
import csv
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("gpcsd::", "")
    return n if len(n) <= 64 else n[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--step-kernel", default="gemm_pred_unfold_kernel,unfold_swap_sum_kernel,swap_last2_sum_kernel",
                    help="kernels that run exactly once per step, first one present in the trace wins")
    ap.add_argument("--from-end", type=int, default=10)
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            stream = r.get("Stream_Id") or r.get("Queue_Id") or "?"
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), stream, r["Kernel_Name"]))
    rows.sort()
    marks = []
    for cand in args.step_kernel.split(","):
        marks = [i for i, r in enumerate(rows) if cand in r[3]]
        if len(marks) >= 3:
            args.step_kernel = cand
            break
    if len(marks) < args.from_end + 2:
        raise SystemExit("trace holds %d steps only" % len(marks))
    # steady-state statistics over the last (up to) 50 steps
    last = marks[-51:] if len(marks) > 51 else marks
    ivals = [(rows[b][1] - rows[a][1]) / 1e6 for a, b in zip(last[:-1], last[1:])]
    t_a, t_b = rows[last[0]][1], rows[last[-1]][1]
    busy = defaultdict(int)
    count = defaultdict(int)
    kbusy = defaultdict(int)
    for s, e, st, name in rows:
        if t_a <= s < t_b:
            busy[st] += e - s
            kbusy[short(name)] += e - s
            count[short(name)] += 1
    nsteps = len(last) - 1
    print("# steady state over the last %d steps of the trace" % nsteps)
    print("ms per step (interval between %s launches): mean %.4f  min %.4f  max %.4f" % (args.step_kernel, sum(ivals) / len(ivals),
                                                                                      min(ivals), max(ivals)))
    for st in sorted(busy):
        print("stream %-4s busy %.4f ms per step" % (st, busy[st] / 1e6 / nsteps))
    print("# kernels per step (launches, ms): top 24 by time")
    for k in sorted(kbusy, key=lambda k: -kbusy[k])[:24]:
        print("%-66s %6.2f launches  %8.4f ms" % (k, count[k] / nsteps, kbusy[k] / 1e6 / nsteps))
    # one step in detail
    i0, i1 = marks[-args.from_end - 2], marks[-args.from_end - 1]
    w0, w1 = rows[i0][1], rows[i1][1]
    print("\n# one step in detail: window = end of one %s launch to the end of the next (%.4f ms)" % (args.step_kernel, (w1 - w0) / 1e6))
    by_stream = defaultdict(list)
    for s, e, st, name in rows:
        if w0 <= s < w1:
            by_stream[st].append((s, e, name))
    for st in sorted(by_stream):
        ks = by_stream[st]
        b = sum(e - s for s, e, _ in ks)
        print("\n## stream %s: %d kernels, busy %.4f ms, first start +%.1f us, last end +%.1f us" % (
            st, len(ks), b / 1e6, (ks[0][0] - w0) / 1e3, (max(e for _, e, _ in ks) - w0) / 1e3))
        for s, e, name in ks:
            print("  +%9.1f us  %8.1f us  %s" % ((s - w0) / 1e3, (e - s) / 1e3, short(name)))


if __name__ == "__main__":
    main()
