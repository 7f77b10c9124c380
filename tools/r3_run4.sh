#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_run4.log
: > $L
for env in "DFH_CHOL_LR=0" "DFH_CHOL_LR=1" "DFH_CHOL_LR_MIN_REM=9728" "DFH_CHOL_LR_MIN_REM=5632"; do
  echo "== n=16384 $env" >> $L
  env $env timeout 300 python tools/time_chol.py 16384 >> $L 2>&1
done
for n in 4096 8192; do echo "== n=$n" >> $L; timeout 300 python tools/time_chol.py $n >> $L 2>&1; done
timeout 600 python tools/syrk_exp.py >> $L 2>&1
bash tools/r3_trace.sh lr1 16384 DFH_CHOL_LR=1 >> $L 2>&1
timeout 900 python -m pytest tests/test_gpu_chol_paths.py -q >> $L 2>&1
grep -v "^W2026\|^E2026" $L | tail -60
