#!/bin/bash
# Round 4, last GPU call: the whole GPU suite and smoke() on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4x; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
tail -c 600 $O/gpu_tests.log; tail -2 $O/smoke.log
