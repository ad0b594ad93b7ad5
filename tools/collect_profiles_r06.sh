#!/bin/bash
# profiles/r06_*_<wl>.* from what tools/profile_r06.sh left under gpurun_out/prof_r06_<wl>/:   bash tools/collect_profiles_r06.sh cfg3|cfg2|cfg3fit
set -eo pipefail
WL=${1:-cfg3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
O=$ROOT/gpurun_out/prof_r06_$WL
cd "$ROOT"
cp "$(find "$O/kt" -name '*kernel_stats.csv' | head -1)" "profiles/r06_kernel_stats_$WL.csv"
cp "$O/only_value.json" "profiles/r06_only_value_$WL.json"
cp "$O/kt_bench.json" "profiles/r06_only_value_under_rocprof_$WL.json"
cp "$O/step_timeline.txt" "profiles/r06_step_timeline_$WL.txt"
cp "$O/pmc_traffic.json" "profiles/r06_pmc_traffic_$WL.json"
[ -f "$O/gemm_counters.json" ] && cp "$O/gemm_counters.json" "profiles/r06_gemm_counters_$WL.json"
ls -la profiles/r06_*_$WL.*
