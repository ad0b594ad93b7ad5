#!/bin/bash
# round 6, GPU call: gradient GEMM A/B (chunk size, tile configuration), then the new tests and the cfg3fit bench
set -o pipefail
OUT=gpurun_out/r6b
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
for ch in 512 256 128; do for gs in 3 5; do for gt in 3 5; do
    GPCSD_GRAD_CH=$ch GPCSD_GRAD_GS_CFG=$gs GPCSD_GRAD_GT_CFG=$gt step 120 ab_${ch}_${gs}_${gt}.txt python tools/grad_timing.py cfg3
    echo "CH=$ch GS=$gs GT=$gt: $(head -1 $OUT/ab_${ch}_${gs}_${gt}.txt) $(grep -h 'gemm_grad_Gt\|gemm_grad_Gs' $OUT/ab_${ch}_${gs}_${gt}.txt | tr -s ' ' | tr '\n' ';')"
done; done; done
GPCSD_GRAD_BRANCHES=0 step 120 ab_nobranch.txt python tools/grad_timing.py cfg3; head -1 $OUT/ab_nobranch.txt
GPCSD_GRAD_KRON=0 step 120 ab_nokron.txt python tools/grad_timing.py cfg3; head -1 $OUT/ab_nokron.txt
step 600 t_new.txt python -m pytest -x -q -m gpu tests/test_q_pipeline.py tests/test_resident_predictions.py -k "not three_models" -p no:cacheprovider
tail -3 $OUT/t_new.txt
step 400 bench_cfg3fit.txt python bench.py --workload cfg3fit --steps 40 --warmup 3
tail -c 2500 $OUT/bench_cfg3fit.txt
cp bench_detail.json $OUT/bench_detail_cfg3fit.json 2>/dev/null
