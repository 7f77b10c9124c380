#!/bin/bash
# round 5: the real run again after the batched fitter's Python was trimmed (stand-in for the per-candidate GP object,
# priors of the unchanged coordinates evaluated once); the reference optimiser on the real engine, 25 runs
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5p; mkdir -p $O
export DRAGONFLY_REFERENCE=$GRAFT_REPO_ROOT/_refscratch
BO_POINTS=$O/pts_install_60.npy timeout 600 python tools/bo_wallclock.py 60 install 2> /dev/null | grep '^{' > $O/bo_install_60.json; cat $O/bo_install_60.json
timeout 900 python tools/bo_wallclock.py 200 install 2> /dev/null | grep '^{' > $O/bo_install_200.json; cat $O/bo_install_200.json
( time timeout 1200 python -m pytest tests/test_gpu_install_end_to_end.py -q ) > $O/install_on_gpu.log 2>&1; tail -4 $O/install_on_gpu.log
