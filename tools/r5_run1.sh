#!/bin/bash
# round 5, call 1: where a lock-step tuning batch spends its time (kernel trace of dfh_gp_lml_batch at
# n = 1000 x 64 and n = 200 x 64), and the timings the round starts from.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5a; mkdir -p $O
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp
  for cfg in "1000 64 5" "200 64 20" "500 64 10" "2000 64 3"; do
    set -- $cfg
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/lml_n$1 -o t -- python $R/tools/prof_lml.py $1 $2 $3 > $O/lml_n$1.log 2>&1
  done )
timeout 400 python tools/time_lml_batch.py > $O/lml_batch.txt 2>&1
for n in 4096 8192 16384; do timeout 200 python tools/time_chol.py $n 5; done > $O/chol.txt 2>&1
find $O -name '*.db' -size +30M -delete
cat $O/lml_n*.log | grep "ms per call"; cat $O/lml_batch.txt; cat $O/chol.txt
