#!/bin/bash
# round 3, GPU run 1: the resident look-ahead factorisation -- correctness of the new paths, then timings
mkdir -p gpurun_out
L=gpurun_out/r3_run1.log
: > $L
timeout 900 python -m pytest tests/test_gpu_chol_paths.py -x -q -k "defaults or resident or forced or no-handoffs" >> $L 2>&1
echo "pytest rc=$?" >> $L
for n in 4096 8192 16384; do
  for env in "DFH_CHOL_LR=0" "DFH_CHOL_LR=1" "DFH_CHOL_LR_MIN_REM=5632" "DFH_CHOL_LR_MIN_REM=9728" "DFH_CHOL_LR_MIN_REM=3584"; do
    if [ $n -lt 8192 ] && [ "$env" != "DFH_CHOL_LR=0" ] && [ "$env" != "DFH_CHOL_LR_MIN_REM=3584" ]; then continue; fi
    echo "== n=$n $env" >> $L
    env $env timeout 300 python tools/time_chol.py $n >> $L 2>&1
  done
done
tail -40 $L
