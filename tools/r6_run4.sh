#!/bin/bash
# round 6, GPU call: stall probe on the bounce-buffer build, then the whole GPU suite
set -o pipefail
OUT=gpurun_out/r6d
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
for k in 1 2 3; do step 200 stall_fixed$k.txt python tools/stall_probe.py cfg2 8; done
step 200 stall_fixed_cfg3.txt python tools/stall_probe.py cfg3 6
step 200 stall_fixed_mixed.txt python tools/two_models_probe.py cfg3,cfg2,cfg2,cfg3,cfg2
for f in $OUT/stall_fixed[123].txt $OUT/stall_fixed_cfg3.txt; do python3 - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "stalled", d["stalled_loops"], [ (l["median_ms"], l["max_ms"]) for l in d["loops"]])
PY
done
tail -2 $OUT/stall_fixed_mixed.txt | cut -c1-1200
step 1000 t_all.txt python -m pytest -x -q -m gpu tests -p no:cacheprovider
tail -8 $OUT/t_all.txt
