#!/bin/bash
# round 6, GPU call 1: new parity tests, cfg3fit bench + per-kernel profile, stall probe variants
set -o pipefail
OUT=gpurun_out/r6a
mkdir -p $OUT
step() {  # step <seconds> <log> <cmd...>: a step that times out or is killed ends the call (no further GPU step)
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s) -> $log"
    timeout -k 10 $secs "$@" > $OUT/$log 2>&1
    local rc=$?
    echo "   rc=$rc"
    if [ $rc -ge 124 ]; then echo "step timed out or was killed: stopping"; tail -5 $OUT/$log; exit $rc; fi
    return 0
}
if [ "$1" != "stall" ]; then
step 600 t_new.txt python -m pytest -x -q -m gpu tests/test_q_pipeline.py tests/test_resident_predictions.py -k "not three_models" -p no:cacheprovider
tail -3 $OUT/t_new.txt
step 600 t_fit.txt python -m pytest -x -q -s -m gpu tests/test_hip_fit2d.py -k "cfg3_objective_and_gradient_at_50 or cfg3_fit_20" -p no:cacheprovider
tail -6 $OUT/t_fit.txt
step 300 grad_timing.txt python tools/grad_timing.py cfg3
step 400 bench_cfg3fit.txt python bench.py --workload cfg3fit --steps 40 --warmup 3
tail -c 3000 $OUT/bench_cfg3fit.txt
cp bench_detail.json $OUT/bench_detail_cfg3fit.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
step 300 prof_fit.txt rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fit -o fit -- python3 bench.py --workload cfg3fit --only-value --fit-batch 1 --steps 30 --warmup 3
fi
if [ "$1" = "stall" ] || [ "$1" = "all" ]; then
for v in "base:" "keep:--keep" "close:--close" "noann:--no-announce" "pin:--pin-lfp"; do
    step 200 stall_${v%%:*}.txt python tools/stall_probe.py cfg2 8 ${v#*:}
done
HSA_ENABLE_SDMA=0 step 200 stall_nosdma.txt python tools/stall_probe.py cfg2 8
HSA_NO_SCRATCH_RECLAIM=1 step 200 stall_noscratchreclaim.txt python tools/stall_probe.py cfg2 8
GPCSD_NO_GRAPH=1 step 200 stall_nograph.txt python tools/stall_probe.py cfg2 8
step 200 stall_base2.txt python tools/stall_probe.py cfg2 8
grep -h stalled_loops $OUT/stall_*.txt | cut -c1-400
fi
