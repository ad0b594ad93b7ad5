"""Committed rocprofv3 summaries (profiles/<round>_*) a bench line quotes: per-step HBM traffic, the dominant kernel's share."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------------- committed profiles
# rocprofv3 summaries of `bench.py --only-value [--workload W]` (tools/profile_r05.sh), one set per workload: a line never
# inherits another workload's numbers (no file for the workload, or a non-default trial count: null).
PROFILE_ROUND = "r06"


def _profile(kind, wl, ext):
    p = os.path.join(ROOT, "profiles", "%s_%s_%s.%s" % (PROFILE_ROUND, kind, wl, ext))
    return p if os.path.exists(p) else None


def pmc_step_traffic(wl):
    """HBM bytes per step from the committed rocprofv3 --pmc passes over `bench.py --only-value` for this workload (FETCH_SIZE
    x2 on gfx950 + WRITE_SIZE, separate passes; tools/pmc_summary.py).  (None, None) if no profile is committed for it."""
    path = _profile("pmc_traffic", wl, "json")
    if path is None:
        return None, None
    with open(path) as fh:
        d = json.load(fh)
    per_step = d.get("hbm_traffic_bytes_per_step")
    top = sorted((r for r in d.get("rows", []) if "hbm_traffic_bytes_per_launch" in r),
                 key=lambda r: -r["hbm_traffic_bytes_per_launch"] * r.get("launches", 1))[:4]
    return per_step, {"source": "profiles/" + os.path.basename(path), "steps_in_profile": d.get("steps"),
                      "largest": [{"kernel": r["kernel"][:60], "bytes_per_launch": r["hbm_traffic_bytes_per_launch"],
                                   "launches_per_step": r.get("launches_per_step")} for r in top]}


def rocprof_kernel(wl, kernel_substr):
    """(share of GPU time, average launch ms, launches, source) of a kernel in the committed `rocprofv3 --kernel-trace --stats`
    summary of `bench.py --only-value` for this workload; Nones if no profile is committed for it."""
    import csv
    path = _profile("kernel_stats", wl, "csv")
    if path is None:
        return None, None, None, None
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if kernel_substr in row.get("Name", ""):
                return (float(row["Percentage"]) / 100.0, float(row["AverageNs"]) * 1e-6, int(row["Calls"]),
                        "profiles/" + os.path.basename(path))
    return None, None, None, "profiles/" + os.path.basename(path)
