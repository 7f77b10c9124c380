#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4v; mkdir -p $O
for lib in libdfhip_dbg.so libdbg_scalar.so; do echo "== $lib"; DFH_LIB=$PWD/dragonfly_amd/$lib timeout 120 python tools/dbg_panel.py 0 | head -12 | cut -c1-140; done > $O/dbg_data.txt 2>&1
cat $O/dbg_data.txt
