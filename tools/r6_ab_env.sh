#!/bin/bash
# round 6, GPU call: A/B of one environment switch on the timed `value` loop, alternating runs.   tools/r6_ab_env.sh VAR val_a val_b [workload]
set -o pipefail
VAR=$1; A=$2; B=$3; WL=${4:-cfg3}
OUT=gpurun_out/ab_$VAR
mkdir -p $OUT
for i in 1 2 3; do
  for v in "$A" "$B"; do
    if [ "$v" = "unset" ]; then
        timeout -k 10 200 python bench.py --workload $WL --only-value --steps 300 --warmup 20 > $OUT/run_${v}_$i.txt 2>&1 || exit 1
    else
        env $VAR=$v timeout -k 10 200 python bench.py --workload $WL --only-value --steps 300 --warmup 20 > $OUT/run_${v}_$i.txt 2>&1 || exit 1
    fi
    echo "$VAR=$v run $i: $(tail -1 $OUT/run_${v}_$i.txt | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"])')"
  done
done
