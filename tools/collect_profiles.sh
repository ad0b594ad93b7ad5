#!/bin/bash
# Rebuild profiles/r05_*_<wl>.* from what tools/profile_r05.sh left under gpurun_out/prof_r05_<wl>/ (gpurun merges only
# gpurun_out/ back from the GPU box):   bash tools/collect_profiles.sh [cfg3|cfg2]
set -eo pipefail
WL=${1:-cfg3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/prof_r05_$WL
cd "$ROOT"
cp "$O/kt/kt_kernel_stats.csv" "profiles/r05_kernel_stats_$WL.csv"
cp "$O/only_value.json" "profiles/r05_only_value_$WL.json"
cp "$O/kt_bench.json" "profiles/r05_only_value_under_rocprof_$WL.json"
python3 tools/step_timeline.py "$O/kt/kt_kernel_trace.csv" > "profiles/r05_step_timeline_$WL.txt"
python3 tools/pmc_summary.py "$O/pmc_fetch" "$O/pmc_write" --out "profiles/r05_pmc_traffic_$WL.json" | tail -2
