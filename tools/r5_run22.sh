#!/bin/bash
# round 5: the one-product-per-block row solve (docs/NOTES_r05.md section 8): BASELINE configs 2 and 5 with and without
# it (bench.other_configs, one process each), then the parity files whose sizes reach it.  The experiment is not in the
# tree: `git apply docs/experiments/r05_trsm_one_product.patch && python -m dragonfly_amd.build` first (DFH_TRSM_FUSED
# does nothing without it).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r5u; mkdir -p $O
for f in 1 0; do
  DFH_TRSM_FUSED=$f timeout 120 python - > $O/configs_fused$f.json 2> $O/configs_fused$f.err <<'PY'
import json, bench
from dragonfly_amd.engine import get_engine
eng = get_engine()
r = bench.other_configs(eng)
print(json.dumps({k: r[k] for k in ('C2', 'C5')}))
PY
  echo "== DFH_TRSM_FUSED=$f"; cat $O/configs_fused$f.json; tail -2 $O/configs_fused$f.err
done
timeout 220 python -m pytest tests/test_gpu_configs.py tests/test_gpu_oracle_parity.py tests/test_gpu_properties.py tests/test_gpu_polyexp.py \
  tests/test_gpu_halluc_at_size.py tests/test_gpu_conditioning.py tests/test_gpu_mgpu.py tests/test_gpu_incremental.py \
  -x -q -m gpu --durations=8 2>&1 | tail -22
