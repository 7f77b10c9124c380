#!/bin/bash
# round 6: k_lml_tiny64 (n <= 63 on the 64 x 64 barrier-free factorisation): parity files that reach it, then latency
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r6c; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hp_tuning.py tests/test_gpu_golden.py tests/test_gpu_post_sampling.py tests/test_gpu_trajectory.py tests/test_gpu_mf_fitter.py tests/test_gpu_polyexp.py -q -x 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_engine_traces.py -q -x 2>&1 | tail -5
for t64 in 0 1; do
  for cfg in "10 1 3000" "30 1 3000" "50 1 5000" "50 3 5000" "63 1 3000" "64 1 2000"; do
    set -- $cfg
    DFH_LML_TINY64=$t64 timeout 120 python tools/prof_small_calls.py $1 $2 $3 | sed "s/^/tiny64=$t64 /"
  done
done
python - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, '.')
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
for n, d in ((50, 3), (20, 3)):
  rs = np.random.RandomState(n); X = rs.rand(n, d); Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  nb = 10000
  specs = [KernelSpec('se', d, float(Y.var()) * np.exp(rs.randn()), np.exp(rs.uniform(np.log(0.3), np.log(3.0), size=d))) for _ in range(nb)]
  means = list(rs.randn(nb) * 0.1); noises = list(float(Y.var()) * np.exp(rs.uniform(np.log(0.005), np.log(0.2), size=nb)))
  Xd = eng.to_device(X)
  for k in (500, 2000, 10000, 10000):
    t0 = time.perf_counter(); eng.gp_lml_batch(specs[:k], Xd, Y, means[:k], noises[:k]); eng.sync()
    print('bulk n=%d nb=%d: %.2f ms' % (n, k, (time.perf_counter() - t0) * 1e3))
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --hip-trace --kernel-trace --stats -d $O/trace_n50_tiny64 -o t -- python $R/tools/prof_small_calls.py 50 1 1000 > $O/trace_n50_tiny64.log 2>&1
find $O -name '*.db' -size +30M -delete
