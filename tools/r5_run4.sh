#!/bin/bash
# round 5, call 4: the team kernel (T workgroups per candidate) -- parity at every size, timings against one workgroup per candidate
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5d; mkdir -p $O
timeout 900 python tools/r5_lml_wg_check.py > $O/team_check.txt 2>&1; echo "rc=$?" >> $O/team_check.txt
DFH_LML_TEAM=0 timeout 900 python tools/r5_lml_wg_check.py > $O/noteam_check.txt 2>&1; echo "rc=$?" >> $O/noteam_check.txt
DFH_LML_TEAM=4 timeout 900 python tools/r5_lml_wg_check.py quick > $O/team4_check.txt 2>&1; echo "rc=$?" >> $O/team4_check.txt
DFH_TEST_SPIN_LIMIT=0 timeout 900 python tools/r5_lml_wg_check.py quick > $O/team_timeout_check.txt 2>&1; echo "rc=$?" >> $O/team_timeout_check.txt
cat $O/team_check.txt; echo ==== no team; grep "nb=" $O/noteam_check.txt; echo === team4; tail -5 $O/team4_check.txt; echo === timeouts forced; tail -5 $O/team_timeout_check.txt
