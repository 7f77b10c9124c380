#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4n; mkdir -p $O
for i in 1 2; do
echo "== default"; timeout 200 python tools/time_kernmat.py | grep "sym"
echo "== nt stores"; DFH_LIB=$PWD/dragonfly_amd/libdfhip_nt.so timeout 200 python tools/time_kernmat.py | grep "sym"
done > $O/km_nt.txt 2>&1
cat $O/km_nt.txt
