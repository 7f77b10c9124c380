#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4h; mkdir -p $O
DBG=$PWD/dragonfly_amd/libdfhip_dbg.so
for tr in 1 0; do echo "== TR=$tr"; DFH_CHOL_FUSED_TR=$tr DFH_LIB=$DBG timeout 120 python tools/dbg_panel.py 0 | head -3 | cut -c1-120; done > $O/dbg_rc.txt 2>&1
# the factorisation tests against the diagnostics build of the same source (different register allocation)
DFH_LIB=$DBG timeout 600 python -m pytest tests/test_gpu_chol_paths.py tests/test_gpu_properties.py tests/test_gpu_oracle_parity.py -m gpu -q -x > $O/tests_dbg.log 2>&1
cat $O/dbg_rc.txt; tail -5 $O/tests_dbg.log
