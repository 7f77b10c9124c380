#!/bin/bash
# Round 4, GPU call 12: the round's rocprofv3 evidence (tools/profile_round.sh) + the process-per-GPU dry runs on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4k; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 0 > $O/dryrun_perprocess_2.json 2> $O/dryrun_perprocess_2.err; echo "perprocess 2 rc=$?"
DFH_BENCH_PER_PROCESS=1 timeout 600 python bench.py --gpus 1 --steps 1 --warmup 1 --no-extras --no-cpu-baseline \
  > $O/dryrun_perprocess_1_rccl.json 2> $O/dryrun_perprocess_1_rccl.err; echo "perprocess 1 (RCCL) rc=$?"
bash tools/profile_round.sh > $O/profile_round.log 2>&1
tail -5 $O/profile_round.log
