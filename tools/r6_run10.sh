#!/bin/bash
set -o pipefail
OUT=gpurun_out/r6j
mkdir -p $OUT
val() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%.0f evals/s (%.3f ms/step; fit %.0f)" % (d["value"], d["ms_per_step"], d["config"].get("fit_evals_per_sec") or 0))
except Exception as e: print("?", e)
PY
}
for wl in cfg5 aud24; do
  for v in "ch256:GPCSD_GRAD_CH=256" "ch512:GPCSD_GRAD_CH=512" "ch512nobr:GPCSD_GRAD_CH=512 GPCSD_GRAD_BRANCHES=0" "ch256nobr:GPCSD_GRAD_CH=256 GPCSD_GRAD_BRANCHES=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs timeout -k 10 200 python bench.py --workload $wl --steps 40 --no-cpu-baseline > $OUT/${wl}_$name.txt 2>&1
    echo "$wl $name: $(val $OUT/${wl}_$name.txt)"
  done
done
