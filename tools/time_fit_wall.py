"""Wall-clock time of consecutive n x n GP fits (no section timers): python tools/time_fit_wall.py [n] [reps]"""
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
pre = int(os.environ.get('PRE_N', '0'))
if pre:      # a smaller fit first: does it absorb the one-off stall of the first large fit?
  eng.sync(); t0 = time.perf_counter()
  eng.gp_fit(spec, Xd.view(0, (pre, d)), yd.view(0, (pre,)), noise).free()
  eng.sync(); print('pre-fit n=%d: %.1f ms' % (pre, (time.perf_counter() - t0) * 1e3))
ts = []
prev = None
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
  eng.sync(); t0 = time.perf_counter()
  gp = eng.gp_fit(spec, Xd, yd, noise)
  eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
  if os.environ.get('KEEP_PREV'):          # the bench's pattern: the old fit is freed after the new one exists
    if prev is not None: prev.free()
    prev = gp
  else:
    gp.free()
print('n=%d fit wall ms: %s' % (n, ' '.join('%.1f' % v for v in ts)))
