#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4t; mkdir -p $O
for v in "A=1" "DFH_CHOL_FUSED_MAX_BATCH=64" "DFH_TS_BATCH=32" "DFH_TS_BATCH=32 DFH_CHOL_FUSED_MAX_BATCH=16"; do
  echo "== $v"; env $v timeout 300 python bench.py --scaling weak --steps 3 --warmup 1 --no-extras --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['sections_ms_extra_untimed_step_rank0'])"
done > $O/ts_variants.txt 2>&1
cat $O/ts_variants.txt
