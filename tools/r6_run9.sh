#!/bin/bash
# round 6: kernel trace of fits at n = 4096 and 16384 (the substitutions' kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6i; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 300 python tools/time_fit_sections.py 4096 16384
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_fit -o t -- python $R/tools/time_fit_sections.py 16384 > $O/trace_fit.log 2>&1
find $O -name '*.db' -size +30M -delete; ls -la $O/trace_fit
