"""Times dfh_gp_lml_batch (the tuning objective of a batch of candidate kernels) for a few sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
for n in (300, 600, 1500, 3000):
  rs = np.random.RandomState(n)
  d = 4
  X = rs.rand(n, d); Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  Xd = eng.to_device(X)
  for nb in (4, 8, 16, 32, 64):
    specs = [KernelSpec('se', d, float(Y.var()) * (0.5 + rs.rand()), 0.2 + 0.6 * rs.rand(d)) for _ in range(nb)]
    means = [0.0] * nb; noises = [float(Y.var() * 0.05)] * nb
    eng.gp_lml_batch(specs, Xd, Y, means, noises)
    ts = []
    for _ in range(5):
      t0 = time.perf_counter(); eng.gp_lml_batch(specs, Xd, Y, means, noises); ts.append((time.perf_counter() - t0) * 1e3)
    print('n=%5d nb=%3d: %7.3f ms per call  (%6.1f us per candidate)' % (n, nb, sorted(ts)[2], sorted(ts)[2] * 1e3 / nb), flush=True)
