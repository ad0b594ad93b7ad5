#!/bin/bash
# round 6, GPU call: the 64-orbit prediction unfold against the 32-orbit one (parity tests, then the step, then FETCH_SIZE)
set -o pipefail
OUT=gpurun_out/r6u
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
step 500 t_pred.txt python -m pytest -x -q -m gpu tests -k "predict or unfold or recipe" -p no:cacheprovider
tail -3 $OUT/t_pred.txt
for bf in 1 2 1 2; do
    GPCSD_UNFOLD_BF=$bf step 200 step_bf${bf}_$RANDOM.txt python bench.py --only-value --steps 300 --warmup 20
done
for f in $OUT/step_bf*.txt; do echo "$(basename $f): $(tail -1 $f | cut -c1-300)"; done
