#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
DFH_LIB=$GRAFT_REPO_ROOT/tools/_exp/libdfhip_dbg_bad.so timeout 120 python tools/dbg_panel_dump.py > $O/dump_bad.txt 2>&1
cat $O/dump_bad.txt
