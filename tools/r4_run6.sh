#!/bin/bash
# Round 4, GPU call 6: pipelined update loops in the transposed panel; last-step stamps; factorisation tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4f; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
DBG=$PWD/dragonfly_amd/libdfhip_dbg.so
DFH_LIB=$DBG timeout 120 python tools/dbg_panel.py 0 | grep -v "^below" > $O/dbg_panel.txt 2>&1
DFH_LIB=$DBG timeout 120 python tools/dbg_panel.py 3584 | grep -v "^below" > $O/dbg_panel_3584.txt 2>&1
for n in 4096 8192 16384; do timeout 120 python tools/time_chol.py $n; done > $O/time_chol.txt 2>&1
( time timeout 900 python -m pytest tests/test_gpu_chol_paths.py tests/test_gpu_properties.py tests/test_gpu_mgpu.py tests/test_gpu_conditioning.py tests/test_gpu_oracle_parity.py tests/test_gpu_incremental.py -m gpu -q -x ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
cat $O/time_chol.txt; sed -n '/hop =/,$p' $O/dbg_panel.txt; grep "launch us" $O/dbg_panel_3584.txt; tail -c 600 $O/gpu_tests.log
