"""Sections of one GP fit (kernel matrix, factorisation, the two solves + lml) at a few sizes, best of 5:
   python tools/time_fit_sections.py [n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
for n in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]:
  rs = np.random.RandomState(n)
  d = 8
  X = rs.rand(n, d)
  Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
  spec = KernelSpec('se', d, float(Y.var()), 0.2 * np.sqrt(d) * np.ones(d))
  noise = float(Y.var() / 20)
  Xd, yd = eng.to_device(X), eng.to_device(Y - float(np.median(Y)))
  for _ in range(2):
    eng.gp_fit(spec, Xd, yd, noise).free()
  eng.sync()
  best = None
  for _ in range(5):
    eng.timings(True)
    gp = eng.gp_fit(spec, Xd, yd, noise)
    eng.sync()
    s = eng.timings(False)
    lml = gp.lml
    gp.free()
    if best is None or s['solve'] < best['solve']:
      best = s
  print('n=%d: kernmat %.3f chol %.3f solve %.3f ms   lml %.15g' % (n, best['kernmat'], best['chol'], best['solve'], lml))
