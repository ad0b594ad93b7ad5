"""Summarise a rocprofv3 --kernel-trace --pmc run of `bench.py --only-value` for the GEMM launches of a step: per (kernel, grid) --
i.e. per distinct launch of the step -- launches, median duration, and the medians of the counters, with the derived figures
north_star asks for: MFMA-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD)) and effective shader clock
(GRBM_GUI_ACTIVE per XCD / duration).
    python tools/gemm_counters.py <rocprof output dir> <out.json> [top]"""
import collections
import csv
import glob
import json
import sys

d, out = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 4
WANT = ("gemm", "wy_qstage", "wy_apply", "tridiag_solve")          # the MFMA kernels of a step (round 6: stage 5's and the solve as well)
want = lambda name: any(w in name for w in WANT)
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
assert cc, "no counter_collection.csv under " + d
rows = list(csv.DictReader(open(cc[0])))
names = {r["Counter_Name"] for r in rows}
key = lambda r: (r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
disp = collections.defaultdict(set)
for r in rows:
    if not want(r["Kernel_Name"]):
        continue
    acc[key(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    disp[key(r)].add(r.get("Dispatch_Id", ""))
dur = collections.defaultdict(list)
seen = set()
for r in rows:                                     # (the counter rows carry their dispatch's time stamps)
    if want(r["Kernel_Name"]) and r.get("Dispatch_Id") not in seen and r.get("Start_Timestamp"):
        seen.add(r.get("Dispatch_Id"))
        dur[key(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
med = lambda v: sorted(v)[len(v) // 2] if v else None
res = []
for k, cs in acc.items():
    dns = med(dur.get(k, []))
    e = {"kernel": k[0][:90], "grid": k[1], "launches": len(disp[k]), "median_duration_us": dns / 1e3 if dns else None,
         "counters_median": {n: med(v) for n, v in cs.items()}}
    g = e["counters_median"].get("GRBM_GUI_ACTIVE")
    b = e["counters_median"].get("SQ_VALU_MFMA_BUSY_CYCLES")
    if g and dns:
        gx = g / 8.0 if g / dns > 3.0 else g              # summed over the 8 XCDs
        e["effective_clock_ghz"] = gx / dns
        if b:
            # SQ_VALU_MFMA_BUSY_CYCLES: cycles with the MFMA pipe busy, summed over the SIMDs that report (per-SE sampling: the guide's
            # caveat); normalised here by active cycles x 1024 SIMDs, and also reported raw
            e["mfma_busy_share_of_active_cycles"] = b / (gx * 1024.0)
    res.append(e)
res.sort(key=lambda e: -(e["median_duration_us"] or 0) * e["launches"])
json.dump({"source": d, "counters": sorted(names), "gemm_launches": res[:top], "all_gemm_kernels": len(res)}, open(out, "w"), indent=1)
for e in res[:top]:
    print("%-72s grid %-8s x%-4d %8.1f us  clock %s GHz  mfma busy %s" % (e["kernel"][:72], e["grid"], e["launches"], e["median_duration_us"] or 0,
          "%.2f" % e["effective_clock_ghz"] if "effective_clock_ghz" in e else "-",
          "%.2f" % e["mfma_busy_share_of_active_cycles"] if "mfma_busy_share_of_active_cycles" in e else "-"))
