#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_run8.log
: > $L
timeout 900 python -m pytest tests/test_gpu_chol_paths.py -q -k "recursive or defaults" >> $L 2>&1
for n in 16384 12288 8192; do
for env in "DFH_CHOL_REC_MIN=0" "DFH_CHOL_REC_MIN=8192" "DFH_CHOL_REC_MIN=6144" "DFH_CHOL_REC_MIN=3072"; do
  echo "== n=$n $env" >> $L
  env $env timeout 300 python tools/time_chol.py $n >> $L 2>&1
done; done
grep -v "^W2026\|^E2026" $L | tail -60
