#!/bin/bash
# round 6, GPU call: alternating runs of the timed `value` loop under sets of environment switches.
#   tools/r6_ab_multi.sh "name1:VAR=a VAR2=b" "name2:" ...      (workload from $WL, default cfg3)
set -o pipefail
WL=${WL:-cfg3}
OUT=gpurun_out/abm
mkdir -p $OUT
for i in 1 2 3; do
  for spec in "$@"; do
    name=${spec%%:*}; envs=${spec#*:}
    env $envs timeout -k 10 200 python bench.py --workload $WL --only-value --steps 300 --warmup 20 > $OUT/${name}_$i.txt 2>&1 || { tail -5 $OUT/${name}_$i.txt; exit 1; }
    echo "$name run $i: $(tail -1 $OUT/${name}_$i.txt | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_step"],4), round(d["value"]), d.get("parity_rel_err_loglik_vs_oracle"), d.get("parity_rel_err_predict_vs_oracle"))')"
  done
done
