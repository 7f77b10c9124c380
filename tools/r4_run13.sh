#!/bin/bash
# first-fit stall: which runtime knob (if any) moves it?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4l; mkdir -p $O
run() { echo "== $*"; env "$@" timeout 120 python tools/time_fit_wall.py 16384 4; }
{
run A=1; run A=1
run ROC_SIGNAL_POOL_SIZE=4096; run ROC_SIGNAL_POOL_SIZE=4096
run GPU_MAX_HW_QUEUES=8; run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2; run GPU_MAX_HW_QUEUES=2
run HIP_FORCE_DEV_KERNARG=0; run HIP_FORCE_DEV_KERNARG=1
run ROC_ACTIVE_WAIT_TIMEOUT=0; run HSA_ENABLE_INTERRUPT=0
run DFH_CHOL_LR=0; run DFH_CHOL_LR=0
} > $O/fit_wall_knobs.txt 2>&1
cat $O/fit_wall_knobs.txt
