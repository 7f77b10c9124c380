#!/bin/bash
# round 6: k_lml_tiny64 after the single logarithm / rectangular triangle loop: stamps, parity, latency
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 20 50 63; do
  echo "== stamps n=$n"; DFH_LIB=$GRAFT_REPO_ROOT/dragonfly_amd/libdfhip_dbg.so DFH_TINY_STAMPS=1 timeout 120 python tools/prof_small_calls.py $n 1 2000 2>&1 | grep -v "DFH_LIB" | tail -2
done
timeout 900 python -m pytest tests/test_gpu_hp_tuning.py tests/test_gpu_golden.py tests/test_gpu_post_sampling.py tests/test_gpu_trajectory.py tests/test_gpu_engine_traces.py -q -x 2>&1 | tail -4
for cfg in "10 1 3000" "30 1 3000" "50 1 5000" "50 3 5000" "63 1 3000" "63 8 2000"; do
  set -- $cfg
  timeout 120 python tools/prof_small_calls.py $1 $2 $3
done
