#!/bin/bash
# Rebuild profiles/r03_*_<wl>.* from what tools/profile_r03.sh left under gpurun_out/prof_r03_<wl>/ (gpurun merges only
# gpurun_out/ back from the GPU box):   bash tools/collect_profiles.sh [cfg3|cfg2]
set -eo pipefail
WL=${1:-cfg3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/prof_r03_$WL
cd "$ROOT"
cp "$O/kt/kt_kernel_stats.csv" "profiles/r03_kernel_stats_$WL.csv"
cp "$O/only_value.json" "profiles/r03_only_value_$WL.json"
cp "$O/kt_bench.json" "profiles/r03_only_value_under_rocprof_$WL.json"
python3 tools/step_timeline.py "$O/kt/kt_kernel_trace.csv" > "profiles/r03_step_timeline_$WL.txt"
python3 tools/pmc_summary.py "$O/pmc_fetch" "$O/pmc_write" --out "profiles/r03_pmc_traffic_$WL.json" | tail -2
