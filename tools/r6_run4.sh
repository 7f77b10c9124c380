#!/bin/bash
# round 6: where k_lml_tiny64's time goes (diagnostics build, s_memrealtime stamps per phase)
cd "$GRAFT_REPO_ROOT" || exit 1
export DFH_LIB=$GRAFT_REPO_ROOT/dragonfly_amd/libdfhip_dbg.so DFH_TINY_STAMPS=1
for direct in 16 0; do for n in 20 50 63; do
  echo "== direct=$direct n=$n"; DFH_LML_DIRECT=$direct timeout 120 python tools/prof_small_calls.py $n 1 3000 2>&1 | grep -v "DFH_LIB" | tail -3
done; done
rocm-smi --showclocks 2>/dev/null | head -20
