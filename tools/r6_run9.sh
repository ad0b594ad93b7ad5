#!/bin/bash
set -o pipefail
OUT=gpurun_out/r6i
mkdir -p $OUT
step() {
    local secs=$1 log=$2; shift 2
    timeout -k 10 $secs "$@" > $OUT/$log 2>&1
    local rc=$?
    if [ $rc -ge 124 ]; then echo "step $log timed out or was killed: stopping"; tail -5 $OUT/$log; exit $rc; fi
    return 0
}
ov() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%.4f ms" % d["ms_per_step"])
except Exception as e: print("?", e)
PY
}
for rep in 1 2; do
  for v in "base:" "ks22:GPCSD_KS_CFG=22" "ks33:GPCSD_KS_CFG=33" "ks25:GPCSD_KS_CFG=25" "res8:GPCSD_RESERVE_CUS=8" "res16:GPCSD_RESERVE_CUS=16" "res8ks22:GPCSD_RESERVE_CUS=8 GPCSD_KS_CFG=22"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs timeout -k 10 200 python bench.py --only-value --steps 100 --warmup 5 > $OUT/ov_${name}_$rep.txt 2>&1
    env $envs GPCSD_BENCH_ANNOUNCE=0 GPCSD_BENCH_SHARE_S=0 timeout -k 10 200 python bench.py --only-value --steps 100 --warmup 5 > $OUT/ovd_${name}_$rep.txt 2>&1
    echo "$name rep $rep: announced $(ov $OUT/ov_${name}_$rep.txt)   library default $(ov $OUT/ovd_${name}_$rep.txt)"
  done
done
step 200 t_recipe.txt python -m pytest -x -q -s -m gpu tests/test_recipe.py tests/test_resident_predictions.py -p no:cacheprovider; tail -4 $OUT/t_recipe.txt
step 200 probe_two_ctx.txt python tools/stall_probe.py cfg3 4; GPU_MAX_HW_QUEUES=16 step 200 probe_two_ctx_hwq16.txt python tools/stall_probe.py cfg3 4
for f in probe_two_ctx probe_two_ctx_hwq16; do python3 - $OUT/$f.txt <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], [l["median_ms"] for l in d["loops"]])
PY
done
