#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_run6.log
: > $L
for env in "DFH_CHOL_LR=1" "DFH_CHOL_LR_NORMAL_PRIO=1"; do
  echo "== n=16384 $env" >> $L
  env $env timeout 300 python tools/time_chol.py 16384 >> $L 2>&1
done
bash tools/r3_trace.sh lrnp 16384 DFH_CHOL_LR_NORMAL_PRIO=1 >> $L 2>&1
grep -v "^W2026\|^E2026" $L | tail -60
