#!/bin/bash
# round 6: lml_wgf_kernel (64 <= n <= 255, <= 16 candidates, one launch): parity, then latency against the schedules it replaces
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1200 python -m pytest tests/test_gpu_lml_fused.py -q -x 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_gpu_lml_wg.py tests/test_gpu_hp_tuning.py tests/test_gpu_engine_traces.py tests/test_gpu_post_sampling.py -q -x 2>&1 | tail -5
for fused in 0 16; do
  for cfg in "64 1 2000" "100 1 2000" "128 1 2000" "128 3 2000" "200 1 1000" "200 8 1000" "255 1 1000"; do
    set -- $cfg
    DFH_LML_FUSED=$fused timeout 120 python tools/prof_small_calls.py $1 $2 $3 | sed "s/^/fused=$fused /"
  done
done
