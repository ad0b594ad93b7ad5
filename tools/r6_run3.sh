#!/bin/bash
# round 6, GPU call: the q-pipeline / resident tests again, then the stall probe's variants (one variable per run)
set -o pipefail
OUT=gpurun_out/r6c
mkdir -p $OUT
step() {
    local secs=$1 log=$2; shift 2
    echo "== $* (limit ${secs}s) -> $log"
    timeout -k 10 $secs "$@" > $OUT/$log 2>&1
    local rc=$?
    echo "   rc=$rc"
    if [ $rc -ge 124 ]; then echo "step timed out or was killed: stopping"; tail -5 $OUT/$log; exit $rc; fi
    return 0
}
step 600 t_new.txt python -m pytest -x -q -m gpu tests/test_q_pipeline.py tests/test_resident_predictions.py -k "not three_models" -p no:cacheprovider
tail -3 $OUT/t_new.txt
GPCSD_GRAD_BRANCHES=0 step 120 ab_nobranch.txt python tools/grad_timing.py cfg3; head -1 $OUT/ab_nobranch.txt
for v in "base:" "keep:--keep" "close:--close" "noann:--no-announce" "pin:--pin-lfp"; do
    step 200 stall_${v%%:*}.txt python tools/stall_probe.py cfg2 8 ${v#*:}
done
HSA_ENABLE_SDMA=0 step 200 stall_nosdma.txt python tools/stall_probe.py cfg2 8
HSA_NO_SCRATCH_RECLAIM=1 step 200 stall_noscratchreclaim.txt python tools/stall_probe.py cfg2 8
GPCSD_NO_GRAPH=1 step 200 stall_nograph.txt python tools/stall_probe.py cfg2 8
step 200 stall_base2.txt python tools/stall_probe.py cfg2 8
for f in $OUT/stall_*.txt; do echo "$(basename $f): $(tail -1 $f | cut -c1-1500)"; done
