#!/bin/bash
# PMC bench passes again (schedule without inter-kernel hand-offs) + the tests touched since the last full run
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_polyexp.py tests/test_gpu_incremental.py tests/test_gpu_oracle_parity.py tests/test_gpu_post_sampling.py tests/test_gpu_nonpsd.py tests/test_gpu_mgpu.py -x -q -m gpu > $REPO/gpurun_out/r3_tests11.log 2>&1
tail -5 $REPO/gpurun_out/r3_tests11.log
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-c4-full"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf $OUT/bench_$set
  DFH_CHOL_LR=0 timeout 900 rocprofv3 --kernel-trace --pmc $set -d $OUT/bench_$set -o bench -- $B --steps 1 --warmup 0 > $OUT/bench_$set.log 2>&1
done
find $OUT -name '*.db' -size +40M -delete
ls -la $OUT/bench_FETCH_SIZE $OUT/bench_WRITE_SIZE
