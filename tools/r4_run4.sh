#!/bin/bash
# Round 4, GPU call 4: early partial own-block product; fallback closures + counters; first-fit trace; the whole GPU suite; dry runs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4d; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
DBG=$PWD/dragonfly_amd/libdfhip_dbg.so
DFH_LIB=$DBG timeout 120 python tools/dbg_panel.py 0 | grep -v "^below" > $O/dbg_panel.txt 2>&1
for n in 4096 8192 16384; do timeout 120 python tools/time_chol.py $n; done > $O/time_chol.txt 2>&1
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
cp gpurun_out/truth_bounds_applied.json $O/ 2>/dev/null
for n in 4 8; do
  DFH_CHUNK_GIB=4 DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1 DFH_CHOL_VERBOSE=1 timeout 900 python bench.py --gpus $n --steps 1 --warmup 0 --no-cpu-baseline --no-extras \
    > $O/dryrun_inprocess_$n.json 2> $O/dryrun_inprocess_$n.err; echo "inprocess $n rc=$?"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/firstfit -o ff -- python $GRAFT_REPO_ROOT/tools/time_fit_wall.py 16384 3 > $GRAFT_REPO_ROOT/$O/firstfit.log 2>&1
cd $GRAFT_REPO_ROOT
ls -la $O/firstfit/* | head; find $O/firstfit -name "*.db" -size +20M -delete
cat $O/time_chol.txt; grep -A9 "hop =" $O/dbg_panel.txt; tail -c 600 $O/gpu_tests.log
