#!/bin/bash
# Round 4, GPU call 21: the final tree -- whole GPU suite on the product build AND on the diagnostics build (same source,
# different code generation), four launcher processes on one device, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4u; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
cp gpurun_out/truth_bounds_applied.json $O/ 2>/dev/null
( time DFH_LIB=$PWD/dragonfly_amd/libdfhip_dbg.so timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests_dbg_build.log 2>&1; echo "rc=$?" >> $O/gpu_tests_dbg_build.log
DFH_CHUNK_GIB=4 DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 \
  --master-port 29513 bench.py --gpus 4 --steps 1 --warmup 0 > $O/dryrun_perprocess_4.json 2> $O/dryrun_perprocess_4.err; echo "perprocess 4 rc=$?"
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 400 $O/gpu_tests.log; tail -c 400 $O/gpu_tests_dbg_build.log
