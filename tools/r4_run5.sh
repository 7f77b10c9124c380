#!/bin/bash
# Round 4, GPU call 5: transposed panel with the schedules' common rounding; announcement from the tail; pre-fit experiment; GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4e; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
DBG=$PWD/dragonfly_amd/libdfhip_dbg.so
DFH_LIB=$DBG timeout 120 python tools/dbg_panel.py 0 | grep -v "^below" > $O/dbg_panel.txt 2>&1
for n in 4096 8192 16384; do timeout 120 python tools/time_chol.py $n; done > $O/time_chol.txt 2>&1
for pre in 0 8192 9216 0 8192; do echo "PRE_N=$pre"; PRE_N=$pre timeout 120 python tools/time_fit_wall.py 16384 4; done > $O/fit_wall.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
cp gpurun_out/truth_bounds_applied.json $O/ 2>/dev/null
cat $O/time_chol.txt $O/fit_wall.txt; grep -A9 "hop =" $O/dbg_panel.txt; tail -c 1500 $O/gpu_tests.log
