#!/bin/bash
set -o pipefail
OUT=gpurun_out/r6k
mkdir -p $OUT
step() {
    local secs=$1 log=$2; shift 2
    echo "== $* -> $log"
    timeout -k 10 $secs "$@" > $OUT/$log 2>&1
    local rc=$?
    echo "   rc=$rc"
    if [ $rc -ge 124 ]; then echo "step $log timed out or was killed: stopping"; tail -5 $OUT/$log; exit $rc; fi
    return 0
}
step 120 smoke.txt python -c "import __graft_entry__ as g; g.smoke()"; tail -3 $OUT/smoke.txt
step 400 bench_default.txt python bench.py --gpus 1 --steps 20 --warmup 5
tail -1 $OUT/bench_default.txt | cut -c1-3900
cp bench_detail.json $OUT/bench_detail_default.json 2>/dev/null
step 1000 t_all.txt python -m pytest -x -q -m gpu tests -p no:cacheprovider
tail -6 $OUT/t_all.txt
