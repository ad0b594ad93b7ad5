#!/bin/bash
# round 6, GPU call: the whole GPU suite, smoke() and the driver's bench command on the build as committed
set -o pipefail
OUT=gpurun_out/r6final
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
step 1000 t_gpu.txt python -m pytest -x -q -m gpu tests -p no:cacheprovider
tail -4 $OUT/t_gpu.txt
grep -q " passed" $OUT/t_gpu.txt || exit 1
step 120 smoke.txt python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
tail -1 $OUT/smoke.txt
step 400 bench.txt python bench.py --gpus 1 --steps 20 --warmup 5
tail -1 $OUT/bench.txt | cut -c1-600
