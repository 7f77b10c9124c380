#!/bin/bash
# Round 4, GPU call 11: final state of the factorisation work -- stamps, timings, the whole GPU suite (product build),
# the factorisation tests against the diagnostics build, dry runs, the default bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4j; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
DBG=$PWD/dragonfly_amd/libdfhip_dbg.so
DFH_LIB=$DBG timeout 120 python tools/dbg_panel.py 0 | grep -v "^below" > $O/dbg_panel.txt 2>&1
DFH_LIB=$DBG timeout 120 python tools/dbg_panel.py 3584 | grep -v "^below" > $O/dbg_panel_3584.txt 2>&1
for n in 4096 8192 16384; do timeout 120 python tools/time_chol.py $n; done > $O/time_chol.txt 2>&1
timeout 200 python tools/time_kernmat.py > $O/time_kernmat.txt 2>&1
( time DFH_LIB=$DBG timeout 900 python -m pytest tests/test_gpu_chol_paths.py tests/test_gpu_properties.py tests/test_gpu_mgpu.py tests/test_gpu_conditioning.py tests/test_gpu_oracle_parity.py tests/test_gpu_incremental.py tests/test_gpu_hp_tuning.py -m gpu -q ) > $O/gpu_tests_dbg_build.log 2>&1; echo "rc=$?" >> $O/gpu_tests_dbg_build.log
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
cp gpurun_out/truth_bounds_applied.json $O/ 2>/dev/null
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for n in 2 4 8; do
  DFH_CHUNK_GIB=4 DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 900 python bench.py --gpus $n --steps 1 --warmup 0 --no-cpu-baseline --no-extras \
    > $O/dryrun_inprocess_$n.json 2> $O/dryrun_inprocess_$n.err; echo "inprocess $n rc=$?"
done
cat $O/time_chol.txt; grep "hop\|strip 7" $O/dbg_panel.txt | head -4; grep "launch us" $O/dbg_panel_3584.txt | cut -c1-100; tail -c 400 $O/gpu_tests_dbg_build.log; tail -c 500 $O/gpu_tests.log
