#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r3_run9.log
: > $L
timeout 900 python -m pytest tests/test_gpu_mgpu.py -q -x >> $L 2>&1
timeout 1500 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_r3a.json 2> gpurun_out/bench_r3a.err
echo "bench rc=$?" >> $L
tail -c 600 gpurun_out/bench_r3a.err >> $L
grep -v "^W2026\|^E2026" $L | tail -40
