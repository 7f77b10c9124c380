#!/bin/bash
# round 5: whole GPU suite + smoke on the tree with the lower-triangle-only Gram build; fit sections at n = 16384
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5m; mkdir -p $O
( time timeout 3000 python -m pytest tests -q -m gpu ) > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 300 python tools/time_kernmat.py 2>&1 | tail -8
for n in 4096 16384; do timeout 200 python tools/time_chol.py $n 4; done
