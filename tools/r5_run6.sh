#!/bin/bash
# round 5: stamps of the team kernel (diagnostics build) + check/timings of the product build
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5f; mkdir -p $O
DFH_LIB=$GRAFT_REPO_ROOT/dragonfly_amd/libdfhip_dbg.so timeout 300 python tools/dbg_lmlt.py 1000 8 > $O/lmlt_1000_8.txt 2>&1
DFH_LIB=$GRAFT_REPO_ROOT/dragonfly_amd/libdfhip_dbg.so timeout 300 python tools/dbg_lmlt.py 1000 64 > $O/lmlt_1000_64.txt 2>&1
timeout 900 python tools/r5_lml_wg_check.py > $O/team_check.txt 2>&1; echo "rc=$?" >> $O/team_check.txt
grep -c "OK$" $O/team_check.txt; grep -v "OK$" $O/team_check.txt
grep -A40 "column 7 " $O/lmlt_1000_8.txt | head -12; grep -A20 "factor step" $O/lmlt_1000_8.txt; tail -3 $O/lmlt_1000_64.txt
