"""Times the n x n Cholesky (dfh_cholesky on a device buffer) and the batched TS-style SYRK."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 32
rs = np.random.RandomState(103)
X = rs.random_sample((n, d)); Y = (X ** 2).dot((np.arange(d) + 1.0) / d) + 0.01 * rs.randn(n)
spec = KernelSpec('se', d, float(Y.var()), 0.2 * np.sqrt(d) * (0.5 + np.arange(d) / 32.0))
Xd, yd = eng.to_device(X), eng.to_device(Y - np.median(Y))
noise = float(Y.var() / 20)
eng.gp_fit(spec, Xd, yd, noise).free()
ts = []
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
  eng.timings(True)
  gp = eng.gp_fit(spec, Xd, yd, noise)
  t = eng.timings(False)
  ts.append(t['chol'])
  gp.free()
print('n=%d cholesky ms: %s  median %.2f  (%.1f TF/s)' % (n, ' '.join('%.2f' % v for v in ts), sorted(ts)[len(ts) // 2], n ** 3 / 3 / (sorted(ts)[len(ts) // 2] * 1e-3) / 1e12))
