#!/bin/bash
# Profiles of the timed regions, round 6; reproducible from the repo root on the GPU box:
#     bash tools/profile_r06.sh cfg3|cfg2|cfg3fit [steps]
# Every pass runs `python3 bench.py --only-value` (setup + warm-up + timed loop, nothing else; program directly behind `--`):
#   1. rocprofv3 --kernel-trace --stats          -> kernel_stats, step_timeline
#   2. rocprofv3 --pmc FETCH_SIZE                 (its own pass, as the gfx950 guide prescribes)
#   3. rocprofv3 --pmc WRITE_SIZE                 -> pmc_traffic (tools/pmc_summary.py: FETCH_SIZE x2 on gfx950)
#   4. (cfg3 only) rocprofv3 --kernel-trace --pmc <SQ / GRBM counters>  -> gemm_counters (tools/gemm_counters.py)
# and one unprofiled run of the same command.  Everything lands under gpurun_out/prof_r06_<wl>/ (gpurun brings only that
# directory home); tools/collect_profiles_r06.sh copies the summaries into profiles/r06_*.
set -eo pipefail
WL=${1:-cfg3}
STEPS=${2:-100}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_r06_$WL
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
if [ "$WL" = "cfg3fit" ]; then
    CMD="bench.py --only-value --workload cfg3fit --steps $STEPS --warmup 5"      # (the bench step: all 8 restarts in one lock-step batch)
    STEPK="fwdR_grad_kernel"
else
    CMD="bench.py --only-value --workload $WL --steps $STEPS --warmup 5"
    STEPK="gemm_pred_unfold_kernel,unfold_swap_sum_kernel,swap_last2_sum_kernel"
fi
python3 $CMD > "$OUT/only_value.json"
echo "[profile] kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 $CMD > "$OUT/kt_bench.json"
python3 tools/step_timeline.py "$(find "$OUT/kt" -name '*kernel_trace.csv' | head -1)" --step-kernel "$STEPK" > "$OUT/step_timeline.txt"
# (counter collection serialises kernels: the gated stage of a pipelined chain cannot make progress under it -- its gates would
# give up, the call be repeated unpipelined and the pipeline latched off (gpcsd_q_pipeline_stats); switched off up front instead)
export GPCSD_Q_PIPE=0
echo "[profile] pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o f -- python3 $CMD --steps 20 --setup-steps 20 > "$OUT/pmc_fetch.json"
echo "[profile] pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o w -- python3 $CMD --steps 20 --setup-steps 20 > "$OUT/pmc_write.json"
python3 tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" --per-step-kernel "$STEPK" --out "$OUT/pmc_traffic.json" > "$OUT/pmc_summary.txt"
tail -3 "$OUT/pmc_summary.txt"
if [ "$WL" = "cfg3" ]; then
    echo "[profile] MFMA counters"
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY \
        --output-format csv -d "$OUT/counters" -o g -- python3 $CMD --steps 20 --setup-steps 20 > "$OUT/counters_bench.json"
    python3 tools/gemm_counters.py "$OUT/counters" "$OUT/gemm_counters.json" 12 | tee "$OUT/gemm_counters.txt"
fi
rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/counters"        # (raw per-dispatch CSVs: tens of MB)
find "$OUT/kt" -name '*kernel_trace.csv' -delete
echo "[profile] done: $OUT"
