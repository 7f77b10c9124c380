#!/bin/bash
# Round 6: a real run (dragonfly.maximise_function on Hartmann6, default options) with install(): 60 and 200 evaluations,
# and the 200-evaluation run once more with the decision margins recorded on the device (dragonfly_amd/gaplog.py).
# The Dragonfly checkout is shipped as untracked scratch for the call (cp -r /root/reference/dragonfly _refscratch/).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_real; mkdir -p $O
export DRAGONFLY_REFERENCE=$GRAFT_REPO_ROOT/_refscratch
for ev in 60 200; do
  BO_POINTS=$O/pts_install_$ev.npy timeout 900 python tools/bo_wallclock.py $ev install 2> $O/bo_install_$ev.err | grep '^{' > $O/bo_install_$ev.json
  cut -c1-400 $O/bo_install_$ev.json
done
DFH_GAP_LOG=$O/gaps_200.json timeout 900 python tools/bo_wallclock.py 200 install 2> $O/bo_gaps_200.err | grep '^{' > $O/bo_gaps_200.json
python -c "
import json; d=json.load(open('$O/bo_gaps_200.json')); print(d['wall_s'], d['max_val']); print({k:{kk:vv for kk,vv in v.items() if kk!='deciles'} for k,v in d['argmax_gaps'].items()})"
