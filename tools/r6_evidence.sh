#!/bin/bash
# Round 6, evidence runs that need the Dragonfly checkout beside the GPU (shipped as untracked scratch: cp -r
# /root/reference/dragonfly _refscratch/): the reference optimiser on the real engine, 25 configurations
# (tests/test_gpu_install_end_to_end.py); a 1000-evaluation run with install(); and the counter passes over
# tools/pmc_workload.py (MFMA utilisation, HBM bytes, LDS conflicts per dispatch).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_evidence; mkdir -p $O
export DRAGONFLY_REFERENCE=$GRAFT_REPO_ROOT/_refscratch
( time timeout 1500 python -m pytest tests/test_gpu_install_end_to_end.py -q -rA -s ) > $O/install_on_gpu.log 2>&1; tail -5 $O/install_on_gpu.log
timeout 1500 python tools/bo_wallclock.py 1000 install 2> $O/bo_install_1000.err | grep '^{' > $O/bo_install_1000.json; cut -c1-300 $O/bo_install_1000.json
unset DRAGONFLY_REFERENCE
bash tools/profile_round.sh wl-only > $O/profile_wl.log 2>&1; echo "wl rc=$?"
