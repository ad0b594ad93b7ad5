#!/bin/bash
# round 6, GPU call: what triggers the stall (injected events on a model in steady state), and its scope (a second process's heartbeat)
set -o pipefail
OUT=gpurun_out/r6e
mkdir -p $OUT
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $OUT/$log 2>&1
    local rc=$?
    if [ $rc -ge 124 ]; then echo "step $log timed out or was killed: stopping"; tail -5 $OUT/$log; exit $rc; fi
    return 0
}
step 200 probe_base.txt python tools/stall_probe.py cfg2 8
step 200 probe_settle.txt python tools/stall_probe.py cfg2 8 --settle
for ev in none malloc mallocfree manysmall hostalloc hostallocfree numpy newctx graphs; do
    step 120 inj_$ev.txt python tools/stall_inject.py $ev
    echo "$ev: $(grep -h '^{' $OUT/inj_$ev.txt | cut -c1-600)"
done
PROBE_HEARTBEAT=1 step 120 inj_graphs_hb.txt python tools/stall_inject.py graphs
grep -h '^{\|HEARTBEAT' $OUT/inj_graphs_hb.txt | cut -c1-900
GPU_MAX_HW_QUEUES=4 step 200 probe_hwq4.txt python tools/stall_probe.py cfg2 8
for f in probe_base probe_settle probe_hwq4; do python3 - $OUT/$f.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "stalled", d["stalled_loops"], [(l["max_ms"], l["stall_s_after_first_evaluation"], l["untimed_stalls_s_after_first_evaluation"]) for l in d["loops"]])
PY
done
