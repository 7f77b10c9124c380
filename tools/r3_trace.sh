#!/bin/bash
# kernel-trace timeline of one n x n fit under the given environment: tools/r3_trace.sh TAG N [ENV=VAL ...]
TAG=$1; N=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace -d $OUT -o t -- python $REPO/tools/time_chol.py $N 2 > $OUT/run.log 2>&1
DB=$(find $OUT -name '*.db' | head -1)
python $REPO/tools/rocpd_timeline.py $DB > $REPO/gpurun_out/timeline_$TAG.txt 2>&1
find $OUT -name '*.db' -delete
tail -3 $OUT/run.log; tail -2 $REPO/gpurun_out/timeline_$TAG.txt
