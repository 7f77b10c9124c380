#!/bin/bash
# round 5: a 60-evaluation dragonfly.maximise_function run with install() after the candidate decode of install.py
# (docs/NOTES_r05.md section 5): wall-clock, and the trajectory against the one recorded before it (tools/r5_run11.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5v; mkdir -p $O
export DRAGONFLY_REFERENCE=$GRAFT_REPO_ROOT/_refscratch
BO_POINTS=$O/pts_install.npy timeout 200 python tools/bo_wallclock.py 60 install 2> $O/err.txt | grep '^{' > $O/bo_install_60.json
cat $O/bo_install_60.json | cut -c1-600; tail -3 $O/err.txt
