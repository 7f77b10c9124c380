#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_run10.log
: > $L
timeout 1200 python -m pytest tests/test_gpu_chol_paths.py tests/test_gpu_mgpu.py -q -x >> $L 2>&1
for n in 2048 4096 8192 16384; do echo "== n=$n" >> $L; timeout 300 python tools/time_chol.py $n >> $L 2>&1; done
timeout 600 python -m pytest tests/test_gpu_golden.py tests/test_gpu_oracle_parity.py tests/test_gpu_hp_tuning.py -q -x >> $L 2>&1
grep -v "^W2026\|^E2026" $L | tail -30
