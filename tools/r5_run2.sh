#!/bin/bash
# round 5, call 2: first run of the one-workgroup-per-candidate tuning objective
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5b; mkdir -p $O
timeout 600 python tools/r5_lml_wg_check.py > $O/wg_check.txt 2>&1
echo "rc=$?" >> $O/wg_check.txt
DFH_LML_WG=0 timeout 300 python tools/r5_lml_wg_check.py quick > $O/wg_check_legacy.txt 2>&1
cat $O/wg_check.txt; echo ==== legacy; cat $O/wg_check_legacy.txt | tail -8
