#!/bin/bash
# Round 6: latency of small tuning calls (dfh_gp_lml_batch with 1 - 8 candidates, docs/NOTES_r06.md section 2).
#   bash tools/r6_small_calls.sh            parity files of the tuning objective, wall per call by size, bulk batches
#   bash tools/r6_small_calls.sh traces     + rocprofv3 --hip-trace --kernel-trace of 1000 one-candidate calls at n = 50:
#                                             round-5 path (DFH_LML_DIRECT=0 DFH_LML_TINY64=0), mapped buffer only, shipped
#   bash tools/r6_small_calls.sh stamps     + k_lml_tiny64's phases (diagnostics build: python -m dragonfly_amd.build --debug-hooks)
# Run through gpurun from the repo root; summaries: profiles/r06_small_calls.txt.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_small; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_lml_fused.py tests/test_gpu_lml_wg.py tests/test_gpu_hp_tuning.py tests/test_gpu_engine_traces.py tests/test_gpu_post_sampling.py tests/test_gpu_golden.py tests/test_gpu_trajectory.py -q -x 2>&1 | tail -6
for variant in "DFH_LML_DIRECT=0 DFH_LML_TINY64=0 DFH_LML_FUSED=0" "DFH_LML_TINY64=0 DFH_LML_FUSED=0" ""; do
  echo "== ${variant:-shipped}"
  for cfg in "10 1 3000" "30 1 3000" "50 1 5000" "50 3 5000" "63 1 3000" "64 1 2000" "100 1 2000" "128 1 2000" "128 3 2000" "160 1 1000" "191 1 1000" "191 8 1000" "200 1 1000" "200 8 1000" "1000 8 300"; do
    set -- $cfg
    env $variant timeout 120 python tools/prof_small_calls.py $1 $2 $3
  done
done
if [ "$1" = "stamps" ]; then
  for n in 20 50 63; do
    echo "== stamps n=$n"; DFH_LIB=$R/dragonfly_amd/libdfhip_dbg.so DFH_TINY_STAMPS=1 timeout 120 python tools/prof_small_calls.py $n 1 2000 2>&1 | grep -v "DFH_LIB" | tail -2
  done
fi
if [ "$1" = "traces" ]; then
  cd /tmp && export TMPDIR=/tmp
  for tag in "before:DFH_LML_DIRECT=0 DFH_LML_TINY64=0" "mapped:DFH_LML_TINY64=0" "shipped:"; do
    name=${tag%%:*}; variant=${tag#*:}
    env $variant timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $O/trace_n50_$name -o t -- python $R/tools/prof_small_calls.py 50 1 1000 > $O/trace_n50_$name.log 2>&1
  done
  find $O -name '*.db' -size +30M -delete
fi
