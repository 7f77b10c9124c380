#!/bin/bash
# lock-step tuning batches: up to which batch size do the (round-4) one-launch panels win?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4r; mkdir -p $O
for mb in 16 32 64 1; do echo "== DFH_CHOL_FUSED_MAX_BATCH=$mb"; DFH_CHOL_FUSED_MAX_BATCH=$mb timeout 300 python tools/time_lml_batch.py; done > $O/lml_batch.txt 2>&1
cat $O/lml_batch.txt
