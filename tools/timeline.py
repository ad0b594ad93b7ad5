"""Timeline of the last steps in a rocprofv3 kernel trace (CSV): per hardware queue, which kernels ran when.

    python tools/timeline.py gpurun_out/tl/**/*_kernel_trace.csv [ms_window]

Prints, for the last `ms_window` milliseconds of the trace (default 5), one line per kernel: start offset, duration, queue and
name (demangled names shortened), plus per-queue busy time -- enough to see which stream is the critical path of a step and
where it idles."""
import csv
import glob
import sys


def main():
    paths = [p for a in sys.argv[1:] if not a.replace(".", "").isdigit() for p in glob.glob(a, recursive=True)]
    win = [float(a) for a in sys.argv[1:] if a.replace(".", "").isdigit()]
    win_ms = win[0] if win else 5.0
    rows = []
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                             r["Kernel_Name"]))
    rows.sort()
    end = max(r[1] for r in rows)
    t0 = end - int(win_ms * 1e6)
    sel = [r for r in rows if r[0] >= t0]
    queues = sorted({(r[2], r[3]) for r in sel})
    qname = {q: "q%d" % i for i, q in enumerate(queues)}
    busy = {q: 0 for q in queues}
    for s, e, q, st, name in sel:
        short = name.split("(")[0]
        if len(short) > 70:
            short = short[:70]
        busy[(q, st)] += e - s
        print("%9.1f us  %8.1f us  %-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, qname[(q, st)], short))
    for q in queues:
        print("queue %s (id %s, stream %s): busy %.3f ms of %.3f" % (qname[q], q[0], q[1], busy[q] / 1e6, win_ms))


if __name__ == "__main__":
    main()
