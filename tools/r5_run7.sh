#!/bin/bash
# round 5: the whole GPU suite on the tree with the one-workgroup / team tuning objective
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5g; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
