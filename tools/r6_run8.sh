#!/bin/bash
set -o pipefail
OUT=gpurun_out/r6h
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
step 100 ptr.txt python tools/ptr_attr_probe.py; cat $OUT/ptr.txt | tail -3
step 300 bench_cfg3.txt python bench.py --steps 100 --warmup 3 --no-sub-results --no-cpu-baseline
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r6h/bench_cfg3.txt').read().strip().splitlines()[-1])
c=d['config']; print({k:c[k] for k in ('fenced_loglik_ms','fenced_predict_ms','library_default_ms_per_step','unannounced_ms_per_step','two_steps_in_flight_ms')}, d['ms_per_step'])
PY
step 300 bench_cfg5_b4.txt python bench.py --workload cfg5 --fit-batch 4 --steps 100 --no-cpu-baseline
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r6h/bench_cfg5_b4.txt').read().strip().splitlines()[-1])
print('cfg5 batch 4:', d['value'], 'evals/s', d['ms_per_step'], 'ms per batch')
PY
step 1000 t_all.txt python -m pytest -x -q -m gpu tests -p no:cacheprovider
tail -6 $OUT/t_all.txt
