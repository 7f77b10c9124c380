#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine_traces.py -q -m gpu -s > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
