#!/bin/bash
# Soak, fuzz and eigensolver stress on the final build of round 6 -> gpurun_out/r06_soak_fuzz_stress.txt (copied to profiles/)
set -o pipefail
OUT=gpurun_out/r06_soak_fuzz_stress.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout -k 10 400 "$@" >> $OUT 2>&1; local rc=$?; echo "(exit $rc)" >> $OUT; if [ $rc -ge 124 ]; then echo "timed out: $*"; exit $rc; fi; }
run python tools/soak_paired.py cfg3 600 tri 16
run python tools/soak_paired.py cfg3 600 x 6
run python tools/soak_paired.py cfg2 600 tri 24
run python tools/fuzz_models.py 60 5
FUZZ_GRAD=1 run python tools/fuzz_models.py 120 7
FUZZ_BIG=1 run python tools/fuzz_models.py 24 11
run python tools/eigh_stress.py
grep -v "^Librccl\|amdgpu.ids\|^$" $OUT | tail -40
