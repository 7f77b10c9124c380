"""The tuning objective as the slice sampler calls it (dragonfly/gp/gp_core.py:551-574 under sampling/slice.py):
many calls of 1 - 8 candidates.  Wall per call, binding included; run plain or under
rocprofv3 --hip-trace --kernel-trace --stats.
   python tools/prof_small_calls.py n nb calls [se|matern]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
n, nb, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else 'se'
rs = np.random.RandomState(n)
d = 3 if n <= 50 else 6
X = rs.rand(n, d); Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
Xd = eng.to_device(X)
pool = 64
specs = [KernelSpec(kind, d, float(Y.var()) * (0.5 + rs.rand()), 0.2 + 0.6 * rs.rand(d), nu=2.5 if kind == 'matern' else 0.0) for _ in range(pool * nb)]
means = list(0.1 * rs.randn(pool * nb)); noises = list(float(Y.var()) * (0.02 + 0.1 * rs.rand(pool * nb)))
for k in range(20):
  eng.gp_lml_batch(specs[:nb], Xd, Y, means[:nb], noises[:nb])
eng.sync()
t0 = time.perf_counter()
for k in range(calls):
  o = (k % pool) * nb
  eng.gp_lml_batch(specs[o:o + nb], Xd, Y, means[o:o + nb], noises[o:o + nb])
dt = time.perf_counter() - t0
print('%s n=%d nb=%d: %.2f us per call (%d calls, binding included)' % (kind, n, nb, dt * 1e6 / calls, calls))
