#!/bin/bash
# Round profile on the MI355X box (run through gpurun from the repo root):
#   kernel trace + stats of bench.py, PMC passes over tools/pmc_workload.py and over bench.py.
# Outputs rocpd databases under gpurun_out/prof/; summarise locally with tools/rocpd_*.py.
#   bash tools/profile_round.sh            everything
#   bash tools/profile_round.sh bench-only  the passes over bench.py only (keeps the workload passes already there)
#   bash tools/profile_round.sh wl-only     the counter passes over tools/pmc_workload.py only
set -u
MODE=${1:-all}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
if [ "$MODE" = "all" ]; then rm -rf $OUT; elif [ "$MODE" = "bench-only" ]; then rm -rf $OUT/trace $OUT/bench_FETCH_SIZE $OUT/bench_WRITE_SIZE; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-extras"
if [ "$MODE" != "wl-only" ]; then
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $B --steps 2 --warmup 1 > $OUT/trace.log 2>&1
grep -h "^{\"metric" $OUT/trace.log | cut -c1-4000 > $OUT/bench_under_trace.json
fi
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  [ "$MODE" != "bench-only" ] || break
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d $OUT/wl_$tag -o wl -- python $REPO/tools/pmc_workload.py > $OUT/wl_$tag.log 2>&1
done
# (the resident look-ahead schedule hands tiles over between CONCURRENT kernels; --pmc serialises
#  kernels, every hand-off would time out and fall back, so the counter passes run the schedule
#  without it -- the dominant kernel's launches are the same products either way)
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  [ "$MODE" != "wl-only" ] || break
  DFH_CHOL_LR=0 rocprofv3 --kernel-trace --pmc $set -d $OUT/bench_$set -o bench -- $B --steps 1 --warmup 0 > $OUT/bench_$set.log 2>&1
done
find $OUT -name '*.db' -size +40M -delete
ls -la $OUT $OUT/*/ 2>/dev/null | head -60
