#!/bin/bash
# round 6: latency of small tuning calls (dfh_gp_lml_batch with 1 - 8 candidates): host-mapped direct path on / off,
# plain and under rocprofv3 --hip-trace --kernel-trace --stats; tuning parity tests on the new build
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6b; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_hp_tuning.py tests/test_gpu_lml_wg.py -q -x 2>&1 | tail -3
for mode in 0 16; do
  for cfg in "50 1 5000" "50 3 5000" "128 1 2000" "200 8 1000" "200 1 1000" "1000 8 300"; do
    set -- $cfg
    DFH_LML_DIRECT=$mode timeout 120 python tools/prof_small_calls.py $1 $2 $3 | sed "s/^/direct=$mode /"
  done
done
cd /tmp && export TMPDIR=/tmp
for mode in 0 16; do
  DFH_LML_DIRECT=$mode timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $O/trace_n50_direct$mode -o t -- python $R/tools/prof_small_calls.py 50 1 1000 > $O/trace_n50_direct$mode.log 2>&1
done
timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $O/trace_n200x8 -o t -- python $R/tools/prof_small_calls.py 200 8 500 > $O/trace_n200x8.log 2>&1
find $O -name '*.db' -size +30M -delete
ls -R $O | head -40
