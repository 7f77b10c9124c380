#!/bin/bash
# round 5: the scalar-wave-index anomaly proven on the hardware by a one-instruction patch (docs/NOTES_r05.md section 2).
# Both libraries are the SAME compiler output of the scalar-w source (hipcc -save-temps), re-assembled from the listing:
#   ctrl = the listing as the compiler wrote it (object file byte-identical to the compiler's own)
#   fix  = the listing + `v_accvgpr_mov_b32 a195, a191` in the strip-0 leaf of the block select (the piece the reload left out)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5t; mkdir -p $O
for v in ctrl fix; do
  DFH_LIB=$GRAFT_REPO_ROOT/tools/_exp/libdfhip_dbg_$v.so timeout 100 python tools/dbg_panel.py 0 2>&1 | grep "factor block\|rc " | head -12 > $O/panel_$v.txt
  echo "== $v"; cat $O/panel_$v.txt
done
DFH_LIB=$GRAFT_REPO_ROOT/tools/_exp/libdfhip_dbg_fix.so timeout 100 python tools/dbg_panel_dump.py 2>&1 | head -12 > $O/dump_fix.txt
cat $O/dump_fix.txt
