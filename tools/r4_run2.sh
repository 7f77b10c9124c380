#!/bin/bash
# Round 4, GPU call 2: the transposed one-launch panel + sc1 hand-offs (stamps, timings, the whole GPU suite), bench with the new report fields
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4b; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
DBG=$PWD/dragonfly_amd/libdfhip_dbg.so
for v in "1 1" "1 0" "0 0" "0 1"; do set -- $v
  echo "=== DFH_CHOL_FUSED_TR=$1 DFH_CHOL_FUSED_SC1=$2"; DFH_LIB=$DBG DFH_CHOL_FUSED_TR=$1 DFH_CHOL_FUSED_SC1=$2 timeout 120 python tools/dbg_panel.py 0 | grep -v "^below"
done > $O/dbg_panel_variants.txt 2>&1
DFH_LIB=$DBG timeout 120 python tools/dbg_panel.py 3584 | grep -v "^below" > $O/dbg_panel_3584.txt 2>&1
for v in "1 1" "1 0" "0 0"; do set -- $v
  for n in 4096 8192 16384; do echo -n "TR=$1 SC1=$2 "; DFH_CHOL_FUSED_TR=$1 DFH_CHOL_FUSED_SC1=$2 timeout 120 python tools/time_chol.py $n; done
done > $O/time_chol.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
cp gpurun_out/truth_bounds_applied.json $O/ 2>/dev/null
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
for n in 2 8; do
  DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 600 python bench.py --gpus $n --steps 1 --warmup 0 --no-cpu-baseline --no-extras \
    > $O/dryrun_inprocess_$n.json 2> $O/dryrun_inprocess_$n.err; echo "inprocess $n rc=$?"
done
cat $O/time_chol.txt; tail -c 1500 $O/gpu_tests.log
