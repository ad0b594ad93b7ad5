#!/bin/bash
# Counters of the 64x64 / BK8 GEMM at the folded critical-path shape and its R = 400 counterpart (run through gpurun):
# effective shader clock (GRBM_GUI_ACTIVE / duration), MFMA pipe busy, wave wait / issue-stall split.
OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for M in 19200 153600; do
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
    -d $OUT/m$M --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/gemm_one.py $M 250 250 0 0 2 20 > $OUT/m$M.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for M in (19200, 153600):
    f = glob.glob("$OUT/m%d/**/*counter_collection.csv" % M, recursive=True)
    if not f: print("no counters for", M); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gemm_f64_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    kt = glob.glob("$OUT/m%d/**/*kernel_trace.csv" % M, recursive=True)
    durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0])) if "gemm_f64_kernel" in r["Kernel_Name"]]
    d = sorted(durs)[len(durs) // 2]
    print("M=%d: median duration %.1f us (%.1f TF/s under the profiler)" % (M, d / 1e3, 2.0 * M * 250 * 250 / d / 1e3))
    for k, v in acc.items():
        v = sorted(v)
        print("   %-28s median %.4g" % (k, v[len(v) // 2]))
    if "GRBM_GUI_ACTIVE" in acc:
        g = sorted(acc["GRBM_GUI_ACTIVE"])[len(acc["GRBM_GUI_ACTIVE"]) // 2]
        print("   effective clock: %.2f GHz (GRBM_GUI_ACTIVE / duration; per-XCD sums divide by 8 if > 3)" % (g / d))
PY
