#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_run7.log
: > $L
for env in "DFH_CHOL_LR=0 DFH_CHOL_SERIAL_MIN_REM=100000" "DFH_CHOL_LR=0 DFH_CHOL_SERIAL_MIN_REM=6144" "DFH_CHOL_LR=0 DFH_CHOL_SERIAL_MIN_REM=3072" "DFH_CHOL_LR=0 DFH_CHOL_SERIAL_MIN_REM=0" "DFH_CHOL_LR=1 DFH_CHOL_SERIAL_MIN_REM=100000" "DFH_CHOL_LR=1 DFH_CHOL_SERIAL_MIN_REM=3072"; do
  echo "== n=16384 $env" >> $L
  env $env timeout 300 python tools/time_chol.py 16384 >> $L 2>&1
done
for n in 4096 8192; do for env in "DFH_CHOL_SERIAL_MIN_REM=100000" "DFH_CHOL_SERIAL_MIN_REM=3072" "DFH_CHOL_SERIAL_MIN_REM=0"; do echo "== n=$n $env" >> $L; env $env timeout 300 python tools/time_chol.py $n >> $L 2>&1; done; done
grep -v "^W2026\|^E2026" $L | tail -60
