#!/bin/bash
OUT=gpurun_out/r6m; mkdir -p $OUT
for wl in cfg5 aud24 npx69fit; do for cfg in 3 0; do
  GPCSD_GRAD_MID_CFG=$cfg timeout -k 10 200 python bench.py --workload $wl --steps 40 --warmup 3 --no-cpu-baseline > $OUT/${wl}_mid$cfg.txt 2>&1
  python3 - $OUT/${wl}_mid$cfg.txt $wl $cfg <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
print(sys.argv[2], "MID_CFG", sys.argv[3], "%.3f ms/step (%.0f evals/s) fit %.0f/s" % (d['ms_per_step'], d['value'], c['fit_evals_per_sec']))
PY
done; done
