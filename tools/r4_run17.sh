#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4q; mkdir -p $O
for i in 1 2; do
echo "== default"; timeout 200 python tools/time_kernmat.py | grep "sym"
echo "== direct tile stores"; DFH_LIB=$PWD/dragonfly_amd/libdfhip_direct.so timeout 200 python tools/time_kernmat.py | grep "sym"
echo "== default, no NT"; DFH_KM_NT=0 timeout 200 python tools/time_kernmat.py | grep "sym"
done > $O/km_direct.txt 2>&1
DFH_LIB=$PWD/dragonfly_amd/libdfhip_direct.so timeout 300 python -m pytest tests/test_gpu_oracle_parity.py -m gpu -q -k "kernel" 2>&1 | tail -2
cat $O/km_direct.txt
