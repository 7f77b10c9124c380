"""One size of the batched tuning objective under rocprofv3 --kernel-trace --stats:
   python tools/prof_lml.py n nb reps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
n, nb, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rs = np.random.RandomState(n)
d = 6
X = rs.rand(n, d); Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
Xd = eng.to_device(X)
specs = [KernelSpec('se', d, float(Y.var()) * (0.5 + rs.rand()), 0.2 + 0.6 * rs.rand(d)) for _ in range(nb)]
means = [0.0] * nb; noises = [float(Y.var() * 0.05)] * nb
import time
eng.gp_lml_batch(specs, Xd, Y, means, noises)
t0 = time.perf_counter()
for _ in range(reps):
  eng.gp_lml_batch(specs, Xd, Y, means, noises)
print('n=%d nb=%d: %.3f ms per call' % (n, nb, (time.perf_counter() - t0) * 1e3 / reps))
