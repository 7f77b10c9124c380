#!/bin/bash
# round 5: the tuning-objective variants incl. small n through the one-workgroup / team forms
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5q; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_lml_wg.py -q -m gpu > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
