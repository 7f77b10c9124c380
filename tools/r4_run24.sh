#!/bin/bash
# the split launch of the one-launch panel (DFH_CHOL_FUSED_SPLIT): timings with and without, factorisation tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4w; mkdir -p $O
for sp in 0 1; do for n in 4096 8192 16384; do echo -n "SPLIT=$sp "; DFH_CHOL_FUSED_SPLIT=$sp timeout 120 python tools/time_chol.py $n; done; done > $O/time_chol.txt 2>&1
( time timeout 900 python -m pytest tests/test_gpu_chol_paths.py tests/test_gpu_properties.py tests/test_gpu_mgpu.py tests/test_gpu_conditioning.py tests/test_gpu_oracle_parity.py tests/test_gpu_incremental.py -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
cat $O/time_chol.txt; tail -c 500 $O/gpu_tests.log
