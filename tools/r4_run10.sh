#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4i; mkdir -p $O
for lib in libdfhip_dbg.so libdbg_v1.so libdbg_v2.so; do echo "== $lib"; DFH_LIB=$PWD/dragonfly_amd/$lib timeout 120 python tools/dbg_panel.py 0 | head -2 | cut -c1-100; done > $O/dbg_rc.txt 2>&1
cat $O/dbg_rc.txt
