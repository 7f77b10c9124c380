#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4m; mkdir -p $O
run() { echo "== $*"; env "$@" timeout 120 python tools/time_fit_wall.py 16384 4; }
{ run DFH_CTX_WARMUP=0; run DFH_CTX_WARMUP=2000; run DFH_CTX_WARMUP=2000; run DFH_CTX_WARMUP=8000; run DFH_CTX_WARMUP=8000; run DFH_CTX_WARMUP=0; } > $O/fit_wall_warm.txt 2>&1
cat $O/fit_wall_warm.txt
