#!/bin/bash
OUT=gpurun_out/r6l; mkdir -p $OUT
for cfg in 0 3 2; do
  GPCSD_GRAD_MID_CFG=$cfg timeout -k 10 200 python bench.py --workload cfg3fit --steps 40 --warmup 3 --no-cpu-baseline > $OUT/fit_mid$cfg.txt 2>&1
  python3 - $OUT/fit_mid$cfg.txt $cfg <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print("MID_CFG", sys.argv[2], "B=8 ms/step %.3f (%.0f evals/s)  single %.3f ms  batch4 %.0f/s  fit %.0f/s" % (d['ms_per_step'], d['value'], c['single_eval_ms'], c['batch4_evals_per_sec'], c['fit_evals_per_sec']))
PY
done
