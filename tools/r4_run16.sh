#!/bin/bash
# Round 4, GPU call 16: final validation -- the whole GPU suite, the factorisation tests against the diagnostics build,
# smoke(), the default bench line (profiles/r04_bench.json), kernel-matrix timings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4p; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
DBG=$PWD/dragonfly_amd/libdfhip_dbg.so
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
cp gpurun_out/truth_bounds_applied.json $O/ 2>/dev/null
( time DFH_LIB=$DBG timeout 900 python -m pytest tests/test_gpu_chol_paths.py tests/test_gpu_properties.py tests/test_gpu_mgpu.py tests/test_gpu_conditioning.py tests/test_gpu_oracle_parity.py tests/test_gpu_incremental.py tests/test_gpu_hp_tuning.py -m gpu -q ) > $O/gpu_tests_dbg_build.log 2>&1; echo "rc=$?" >> $O/gpu_tests_dbg_build.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 200 python tools/time_kernmat.py > $O/time_kernmat.txt 2>&1
for n in 4096 8192 16384; do timeout 120 python tools/time_chol.py $n; done > $O/time_chol.txt 2>&1
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 500 $O/gpu_tests.log; tail -c 300 $O/gpu_tests_dbg_build.log; tail -2 $O/smoke.log; grep sym $O/time_kernmat.txt; cat $O/time_chol.txt
