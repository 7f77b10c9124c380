#!/bin/bash
# round 6: a real run (dragonfly.maximise_function on Hartmann6, default options) with install(): 60 and 200 evaluations
# (the Dragonfly checkout is shipped as untracked scratch for this one call)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6f; mkdir -p $O
export DRAGONFLY_REFERENCE=$GRAFT_REPO_ROOT/_refscratch
for ev in 60 200; do
  BO_POINTS=$O/pts_install_$ev.npy timeout 900 python tools/bo_wallclock.py $ev install 2> $O/bo_install_$ev.err | grep '^{' > $O/bo_install_$ev.json
  cut -c1-700 $O/bo_install_$ev.json
done
