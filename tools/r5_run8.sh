#!/bin/bash
# round 5: the scalar-wave-index anomaly of the one-launch panel (docs/NOTES_r05.md section 2).  Libraries from
# tools/r5_build_scalar_w_variants.sh: the tree's diagnostics build (good), kernel-scope scalar w (bad), the same with
# s_nop 4 before every pivot readlane (badnop); each twice (deterministic?), then block 0 of the bad factor entry by entry.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5h; mkdir -p $O
for v in good bad badnop; do
  lib=$GRAFT_REPO_ROOT/tools/_exp/libdfhip_dbg_$v.so; [ $v = good ] && lib=$GRAFT_REPO_ROOT/dragonfly_amd/libdfhip_dbg.so
  for rep in 1 2; do DFH_LIB=$lib timeout 120 python tools/dbg_panel.py 0 2>&1 | grep "factor block\|rc " | head -12 > $O/panel_${v}_$rep.txt; done
  echo "== $v"; cat $O/panel_${v}_1.txt; echo "-- second run: wrong blocks"; grep -c WRONG $O/panel_${v}_2.txt
done
DFH_LIB=$GRAFT_REPO_ROOT/tools/_exp/libdfhip_dbg_bad.so timeout 120 python tools/dbg_panel_dump.py > $O/dump_bad.txt 2>&1
cat $O/dump_bad.txt
