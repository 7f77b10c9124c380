#!/bin/bash
# Round 4, GPU call 3: split publish + deferred announcement in the one-launch panel; warm-up experiment; the GPU suite; dry runs; bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
DBG=$PWD/dragonfly_amd/libdfhip_dbg.so
DFH_LIB=$DBG timeout 120 python tools/dbg_panel.py 0 | grep -v "^below" > $O/dbg_panel.txt 2>&1
for n in 4096 8192 16384; do timeout 120 python tools/time_chol.py $n; done > $O/time_chol.txt 2>&1
for wu in 0 3000 0 3000; do echo -n "DFH_CTX_WARMUP=$wu "; DFH_CTX_WARMUP=$wu timeout 120 python tools/time_fit_wall.py 16384 6; done > $O/fit_wall.txt 2>&1
timeout 200 python tools/time_kernmat.py > $O/time_kernmat.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
cp gpurun_out/truth_bounds_applied.json $O/ 2>/dev/null
for n in 4 8; do
  DFH_CHUNK_GIB=4 DFH_TS_BATCH=8 DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 600 python bench.py --gpus $n --steps 1 --warmup 0 --no-cpu-baseline --no-extras \
    > $O/dryrun_inprocess_$n.json 2> $O/dryrun_inprocess_$n.err; echo "inprocess $n rc=$?"
done
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cat $O/time_chol.txt $O/fit_wall.txt; grep -A9 "hop =" $O/dbg_panel.txt; tail -c 800 $O/gpu_tests.log
