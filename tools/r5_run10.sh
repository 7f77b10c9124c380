#!/bin/bash
# round 5: wall-clock of a real Dragonfly run (maximise_function on Hartmann6, default options), the reference as it is
# and with dragonfly_amd.install(), on the GPU box's own CPU; the reference optimiser on the real engine (25 runs).
# The Dragonfly checkout is shipped as untracked scratch for this one call.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5j; mkdir -p $O
export DRAGONFLY_REFERENCE=$GRAFT_REPO_ROOT/_refscratch
for ev in 60 200; do
  for mode in install ref; do
    timeout 1500 python tools/bo_wallclock.py $ev $mode 2> $O/bo_${mode}_$ev.err | grep '^{' > $O/bo_${mode}_$ev.json
    cat $O/bo_${mode}_$ev.json
  done
done
timeout 1500 python tools/bo_wallclock.py 1000 install 2> $O/bo_install_1000.err | grep '^{' > $O/bo_install_1000.json; cat $O/bo_install_1000.json
( time timeout 1200 python -m pytest tests/test_gpu_install_end_to_end.py -q -rA -s ) > $O/install_on_gpu.log 2>&1; tail -5 $O/install_on_gpu.log
