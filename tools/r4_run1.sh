#!/bin/bash
# Round 4, GPU call 1: the whole GPU suite; the reference optimiser on the real engine (reference shipped as
# untracked scratch for this one call); dry runs of bench.py's N > 1 routes on one GPU; the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
if [ -d _refscratch/dragonfly ]; then
  ( time DRAGONFLY_REFERENCE=$PWD/_refscratch timeout 1200 python -m pytest tests/test_gpu_install_end_to_end.py -q -rA -s ) > $O/install_on_gpu.log 2>&1
  echo "rc=$?" >> $O/install_on_gpu.log
fi
for n in 2 4 8; do
  DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 600 python bench.py --gpus $n --scaling strong --steps 1 --warmup 0 --no-cpu-baseline \
    > $O/dryrun_inprocess_$n.json 2> $O/dryrun_inprocess_$n.err; echo "inprocess $n rc=$?"
done
DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 0 > $O/dryrun_perprocess_2.json 2> $O/dryrun_perprocess_2.err; echo "perprocess 2 rc=$?"
DFH_BENCH_PER_PROCESS=1 timeout 600 python bench.py --gpus 1 --steps 1 --warmup 1 --no-extras --no-cpu-baseline \
  > $O/dryrun_perprocess_1_rccl.json 2> $O/dryrun_perprocess_1_rccl.err; echo "perprocess 1 (RCCL) rc=$?"
for n in 4096 8192 16384; do timeout 120 python tools/time_chol.py $n; done > $O/time_chol.txt 2>&1
for rb in 0 3584; do DFH_LIB=$PWD/dragonfly_amd/libdfhip_dbg.so timeout 120 python tools/dbg_panel.py $rb; done > $O/dbg_panel.txt 2>&1
timeout 200 python tools/time_kernmat.py > $O/time_kernmat.txt 2>&1
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 600 $O/gpu_tests.log; tail -c 1500 $O/install_on_gpu.log
