#!/bin/bash
# Round 6: the two substitutions of a fit on k_trsv_* (docs/NOTES_r06.md section 3): sections of a fit at four sizes with the
# new kernels off / on, the parity files that reach them, a kernel trace of the fit at n = 16384.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_solves; mkdir -p $O
R=$GRAFT_REPO_ROOT
for f in 0 1; do echo "== DFH_TRSV_FAST=$f"; DFH_TRSV_FAST=$f timeout 300 python tools/time_fit_sections.py 1000 4096 8192 16384; done
timeout 1500 python -m pytest tests/test_gpu_oracle_parity.py tests/test_gpu_headline.py tests/test_gpu_conditioning.py tests/test_gpu_properties.py tests/test_gpu_configs.py tests/test_gpu_incremental.py tests/test_gpu_golden.py tests/test_gpu_upper_triangle_unread.py -q -x 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fit -o t -- python $R/tools/time_fit_sections.py 16384 > $O/trace_fit.log 2>&1
find $O -name '*.db' -size +30M -delete
