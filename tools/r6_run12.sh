#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine_traces.py -q -x -s > $O/traces.txt 2>&1
tail -4 $O/traces.txt; grep "ties resolved the other way [1-9]" $O/traces.txt
