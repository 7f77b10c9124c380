#!/bin/bash
# round 6: four-entry Gram loop, trimmed last diagonal tile: parity files of the tuning objective, then latency by size
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_lml_fused.py tests/test_gpu_lml_wg.py tests/test_gpu_hp_tuning.py tests/test_gpu_engine_traces.py tests/test_gpu_post_sampling.py tests/test_gpu_golden.py tests/test_gpu_trajectory.py -q -x 2>&1 | tail -6
for cfg in "10 1 3000" "30 1 3000" "50 1 5000" "50 3 5000" "63 1 3000" "64 1 2000" "100 1 2000" "128 1 2000" "128 3 2000" "160 1 1000" "191 1 1000" "191 8 1000" "200 1 1000" "200 8 1000" "1000 8 300"; do
  set -- $cfg
  timeout 120 python tools/prof_small_calls.py $1 $2 $3
done
python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, '.')
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
for n, d, nbs in ((50, 3, (500, 10000)), (200, 6, (500, 10000)), (1000, 6, (256, 2048)), (2000, 6, (512,))):
  rs = np.random.RandomState(n); X = rs.rand(n, d); Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  nb = max(nbs)
  specs = [KernelSpec('se', d, float(Y.var()) * np.exp(rs.randn()), np.exp(rs.uniform(np.log(0.3), np.log(3.0), size=d))) for _ in range(nb)]
  means = list(rs.randn(nb) * 0.1); noises = list(float(Y.var()) * np.exp(rs.uniform(np.log(0.005), np.log(0.2), size=nb)))
  Xd = eng.to_device(X)
  eng.gp_lml_batch(specs[:64], Xd, Y, means[:64], noises[:64])
  for k in nbs + nbs[-1:]:
    t0 = time.perf_counter(); eng.gp_lml_batch(specs[:k], Xd, Y, means[:k], noises[:k]); eng.sync()
    print('bulk n=%d nb=%d: %.2f ms' % (n, k, (time.perf_counter() - t0) * 1e3))
PY
