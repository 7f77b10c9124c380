#!/bin/bash
# Round 4, GPU call 7: what slowed the pipelined-update build -- polling pressure or the scheduling barriers?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4g; mkdir -p $O
DBG=$PWD/dragonfly_amd/libdfhip_dbg.so; NOSGB=$PWD/dragonfly_amd/libdfhip_nosgb.so
for v in "$DBG 1" "$DBG 8" "$DBG 32" "$NOSGB 8" "$NOSGB 1"; do set -- $v
  echo "=== $(basename $1) DFH_CHOL_PROG_SLEEP=$2"
  DFH_LIB=$1 DFH_CHOL_PROG_SLEEP=$2 timeout 120 python tools/dbg_panel.py 0 | sed -n '/hop =/,$p'
  DFH_LIB=$1 DFH_CHOL_PROG_SLEEP=$2 timeout 120 python tools/dbg_panel.py 3584 | grep "launch us"
done > $O/variants.txt 2>&1
for ps in 1 8 32; do echo -n "PROG_SLEEP=$ps "; DFH_CHOL_PROG_SLEEP=$ps timeout 120 python tools/time_chol.py 4096; done > $O/time_chol.txt 2>&1
cat $O/variants.txt $O/time_chol.txt
