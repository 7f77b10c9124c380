#!/bin/bash
# from how many 128 x 128 tiles on does a product take them instead of 64 x 64 tiles? (the look-ahead products of a factorisation)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4y; mkdir -p $O
for t in 192 128 96 64 32; do for n in 4096 8192; do echo -n "SMALL_T128=$t "; DFH_GEMM_SMALL_T128=$t timeout 120 python tools/time_chol.py $n 5; done; done > $O/time_chol.txt 2>&1
cat $O/time_chol.txt
