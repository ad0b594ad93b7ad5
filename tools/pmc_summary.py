#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per (kernel, grid): launches and mean counter value per launch.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write --out profiles/r01_pmc_traffic.json

Reads either the CSV output or the rocpd sqlite database (view `counters_collection`) of each pass directory.
FETCH_SIZE / WRITE_SIZE are derived counters in units of 1024 B (their rocprofv3 expression ends in "/1024").
Following /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE is DOUBLED on gfx950 (128-B requests tallied
at 64 B) before it is compared with a byte count; WRITE_SIZE is taken as is.  tools/pmc_calib.py gives a launch with a
known byte count (256 MiB read + 256 MiB written) to check both corrections in this library's own access pattern.
"""
import argparse
import csv
import glob
import json
import os
import re
import sqlite3
from collections import defaultdict


def short(name):
    m = re.match(r"(?:void )?(?:gpcsd::)?([A-Za-z0-9_]+)(<[^(]*>)?", name)
    if not m:
        return name[:60]
    return m.group(1) + (m.group(2) or "")


def load(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                key = (short(row["Kernel_Name"]), int(row["Grid_Size"]), int(row["Workgroup_Size"]))
                a = acc[key][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
        con = sqlite3.connect(f)
        for kn, gs, ws, cn, val in con.execute(
                "select kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection"):
            a = acc[(short(kn), int(gs), int(ws))][cn]
            a[0] += 1
            a[1] += float(val)
        con.close()
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", default=None)
    ap.add_argument("--unit-bytes", type=float, default=1024.0, help="bytes per FETCH_SIZE/WRITE_SIZE unit")
    ap.add_argument("--per-step-kernel", default="gemm_pred_unfold_kernel,unfold_swap_sum_kernel,swap_last2_sum_kernel",
                    help="kernels launched exactly once per bench step (first one present wins): its launch count is the number "
                         "of steps profiled")
    args = ap.parse_args()
    merged = defaultdict(dict)
    for d in args.dirs:
        for key, ctrs in load(d).items():
            for cname, (cnt, tot) in ctrs.items():
                merged[key][cname] = {"launches": cnt, "mean": tot / cnt}
    rows = []
    for (kern, grid, wg), ctrs in merged.items():
        r = {"kernel": kern, "grid": grid, "workgroup": wg}
        for cname, v in ctrs.items():
            r["launches"] = v["launches"]
            r[cname] = v["mean"]
        if "FETCH_SIZE" in r:
            r["hbm_read_bytes_per_launch"] = 2.0 * r["FETCH_SIZE"] * args.unit_bytes     # gfx950 correction (x2)
        if "WRITE_SIZE" in r:
            r["hbm_write_bytes_per_launch"] = r["WRITE_SIZE"] * args.unit_bytes
        if "hbm_read_bytes_per_launch" in r and "hbm_write_bytes_per_launch" in r:
            r["hbm_traffic_bytes_per_launch"] = r["hbm_read_bytes_per_launch"] + r["hbm_write_bytes_per_launch"]
        rows.append(r)
    rows.sort(key=lambda r: -r.get("hbm_traffic_bytes_per_launch", 0.0) * r.get("launches", 1))
    for r in rows[:40]:
        print("%-70s grid %9d x%-5d n=%4d  read %10.3f MB  write %10.3f MB" % (
            r["kernel"][:70], r["grid"], r["workgroup"], r.get("launches", 0),
            r.get("hbm_read_bytes_per_launch", float("nan")) / 1e6, r.get("hbm_write_bytes_per_launch", float("nan")) / 1e6))
    steps = 0
    for cand in args.per_step_kernel.split(","):
        steps = sum(r.get("launches", 0) for r in rows if r["kernel"].startswith(cand))
        if steps:
            break
    total = sum(r.get("hbm_traffic_bytes_per_launch", 0.0) * r.get("launches", 0) for r in rows)
    per_step = total / steps if steps else None
    if per_step:
        print("steps profiled: %d   HBM traffic per step: %.1f MB" % (steps, per_step / 1e6))
    if steps:
        for r in rows:
            r["launches_per_step"] = r.get("launches", 0) / steps
    if args.out:
        with open(args.out, "w") as fh:
            json.dump({"unit_bytes": args.unit_bytes, "fetch_correction": 2.0, "steps_profiled": steps, "steps": steps,
                       "hbm_traffic_bytes_per_step": per_step, "rows": rows}, fh, indent=1)


if __name__ == "__main__":
    main()
