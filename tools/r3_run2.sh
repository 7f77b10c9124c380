#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_run2.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_chol_paths.py -q -k "defaults or resident or forced or no-handoffs" >> $L 2>&1
echo "pytest rc=$?" >> $L
for env in "DFH_CHOL_LR=0" "DFH_CHOL_LR=1"; do
  echo "== n=16384 $env" >> $L
  env $env timeout 300 python tools/time_chol.py 16384 >> $L 2>&1
done
bash tools/r3_trace.sh lr1 16384 DFH_CHOL_LR=1 >> $L 2>&1
bash tools/r3_trace.sh lr0 16384 DFH_CHOL_LR=0 >> $L 2>&1
tail -30 $L
