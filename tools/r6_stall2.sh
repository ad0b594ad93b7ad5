#!/bin/bash
# round 6, GPU call: two host-memory-management variables for the stall (huge-page collapse, automatic NUMA balancing)
set -o pipefail
OUT=gpurun_out/r6s2
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
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag /proc/sys/kernel/numa_balancing > $OUT/host_mm.txt 2>&1
grep -c . /sys/devices/system/node/online >> $OUT/host_mm.txt 2>&1; cat /sys/devices/system/node/online >> $OUT/host_mm.txt 2>&1
getconf CLK_TCK >> $OUT/host_mm.txt; grep "CONFIG_HZ" /boot/config-$(uname -r) >> $OUT/host_mm.txt 2>&1
cat $OUT/host_mm.txt
for v in "base:" "thp:--thp-off" "pol:--mempolicy" "both:--thp-off --mempolicy" "base2:"; do
    step 200 stall_${v%%:*}.txt python tools/stall_probe.py cfg2 8 ${v#*:}
done
for f in $OUT/stall_*.txt; do echo "$(basename $f): $(tail -1 $f | cut -c1-900)"; done
