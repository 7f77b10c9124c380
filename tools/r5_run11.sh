#!/bin/bash
# round 5: where a short real run with install() spends its time (cProfile), and where its trajectory leaves the reference's
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5k; mkdir -p $O
export DRAGONFLY_REFERENCE=$GRAFT_REPO_ROOT/_refscratch
BO_POINTS=$O/pts_install.npy BO_PROFILE=$O/profile_install_60.txt timeout 600 python tools/bo_wallclock.py 60 install 2> /dev/null | grep '^{' > $O/bo_install_60.json
BO_POINTS=$O/pts_ref.npy timeout 600 python tools/bo_wallclock.py 60 ref 2> /dev/null | grep '^{' > $O/bo_ref_60.json
python - <<'PY'
import numpy as np, os
O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r5k')
a=np.load(O+'/pts_install.npy'); b=np.load(O+'/pts_ref.npy')
k=[i for i in range(min(len(a),len(b))) if not np.array_equal(a[i],b[i])]
print('points', len(a), len(b), 'first differing evaluation:', k[0] if k else None)
if k:
  i=k[0]; print('install', a[i]); print('ref    ', b[i]); print('max abs diff', np.abs(a[i]-b[i]).max())
PY
head -60 $O/profile_install_60.txt
