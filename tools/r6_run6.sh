#!/bin/bash
set -o pipefail
OUT=gpurun_out/r6f
mkdir -p $OUT
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $OUT/$log 2>&1
    local rc=$?
    if [ $rc -ge 124 ]; then echo "step $log timed out or was killed: stopping"; tail -5 $OUT/$log; exit $rc; fi
    return 0
}
GPCSD_LL_WAIT=spin step 200 probe_spin1.txt python tools/stall_probe.py cfg2 8
GPCSD_LL_WAIT=spin step 200 probe_spin2.txt python tools/stall_probe.py cfg2 8
HSA_ENABLE_INTERRUPT=0 step 200 probe_noirq.txt python tools/stall_probe.py cfg2 8
step 200 probe_base.txt python tools/stall_probe.py cfg2 8
for f in probe_spin1 probe_spin2 probe_noirq probe_base; do python3 - $OUT/$f.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "stalled", d["stalled_loops"], [(l["max_ms"], l["stall_s_after_first_evaluation"], l["untimed_stalls_s_after_first_evaluation"]) for l in d["loops"]])
PY
done
