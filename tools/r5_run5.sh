#!/bin/bash
# round 5, call 5: deeper operand prefetch in the one-workgroup / team objective
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5e; mkdir -p $O
timeout 900 python tools/r5_lml_wg_check.py > $O/team_check.txt 2>&1; echo "rc=$?" >> $O/team_check.txt
cat $O/team_check.txt
