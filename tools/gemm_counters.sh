#!/bin/bash
# MFMA-utilisation counters of the GEMM launches of the bench step (north_star: "rocprof HBM GB/s and MFMA utilisation"), over the
# same command the other committed profiles use:   bash tools/gemm_counters.sh [cfg3|R400]
# -> gpurun_out/gemm_counters_<tag>/ and profiles/r04_gemm_counters_<tag>.json (counters in their own pass; --kernel-trace only).
set -eo pipefail
TAG=${1:-cfg3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/gemm_counters_$TAG
mkdir -p "$OUT" "$ROOT/profiles"
export TMPDIR=/tmp
cd "$ROOT"
EXTRA=""
[ "$TAG" = "R400" ] && EXTRA="--trials-per-gpu 400"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY \
    --output-format csv -d "$OUT/pmc" -o g -- python3 bench.py --only-value --steps 20 --setup-steps 20 --warmup 5 $EXTRA > "$OUT/bench.json"
python3 tools/gemm_counters.py "$OUT/pmc" "profiles/r04_gemm_counters_$TAG.json" 6 | tee "$OUT/summary.txt"
