#!/bin/bash
# Profiles of the timed region, reproducible from the repo root on the GPU box:
#     bash tools/profile_r05.sh [cfg3|cfg2] [steps]
# Every pass runs `python3 bench.py --only-value` (setup + warm-up + timed loop, nothing else; program directly behind `--`):
#   1. rocprofv3 --kernel-trace --stats   -> profiles/r05_kernel_stats_<wl>.csv, profiles/r05_step_timeline_<wl>.txt
#   2. rocprofv3 --pmc FETCH_SIZE         (its own pass, as the gfx950 guide prescribes)
#   3. rocprofv3 --pmc WRITE_SIZE         -> profiles/r05_pmc_traffic_<wl>.json (tools/pmc_summary.py)
# and one unprofiled run of the same command for reference (profiles/r05_only_value_<wl>.json).
set -eo pipefail
WL=${1:-cfg3}
STEPS=${2:-100}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_r05_$WL
mkdir -p "$OUT" "$ROOT/profiles"
export TMPDIR=/tmp
cd "$ROOT"
CMD="bench.py --only-value --workload $WL --steps $STEPS --warmup 5"
python3 $CMD > "$OUT/only_value.json"
cp "$OUT/only_value.json" "profiles/r05_only_value_$WL.json"   # (profiles/ on the GPU box is scratch: tools/collect_profiles.sh rebuilds it from gpurun_out/)
echo "[profile] kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $CMD > "$OUT/kt_bench.json"
cp "$(find "$OUT/kt" -name '*kernel_stats.csv' | head -1)" "profiles/r05_kernel_stats_$WL.csv"
cp "$OUT/kt_bench.json" "profiles/r05_only_value_under_rocprof_$WL.json"
python3 tools/step_timeline.py "$(find "$OUT/kt" -name '*kernel_trace.csv' | head -1)" > "profiles/r05_step_timeline_$WL.txt"
# (counter collection serialises kernels: a launch that waits for a running one -- the gated stage of the unannounced pipelined
# chain, taken by the very first step only -- cannot make progress under it; the steady state of the announced loop queues the same
# kernels ungated, so the switch changes nothing that is measured)
export GPCSD_Q_PIPE=0
echo "[profile] pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o f -- python3 $CMD --steps 20 --setup-steps 20 > "$OUT/pmc_fetch.json"
echo "[profile] pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o w -- python3 $CMD --steps 20 --setup-steps 20 > "$OUT/pmc_write.json"
python3 tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" --out "profiles/r05_pmc_traffic_$WL.json" > "$OUT/pmc_summary.txt"
tail -3 "$OUT/pmc_summary.txt"
echo "[profile] done: profiles/r05_*_$WL.*"
