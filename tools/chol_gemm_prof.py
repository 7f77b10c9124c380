"""GEMM time inside the n x n fit's factorisation from per-launch HIP events (untraced), and the same
trailing updates run alone on the same kind of data."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine, KernelSpec
from dragonfly_amd._lib import check
eng = get_engine()
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 32
rs = np.random.RandomState(103)
X = rs.random_sample((n, d)); Y = (X ** 2).dot((np.arange(d) + 1.0) / d) + 0.01 * rs.randn(n)
spec = KernelSpec('se', d, float(Y.var()), 0.2 * np.sqrt(d) * (0.5 + np.arange(d) / 32.0))
Xd, yd = eng.to_device(X), eng.to_device(Y - np.median(Y))
noise = float(Y.var() / 20)
eng.gp_fit(spec, Xd, yd, noise).free()
for _ in range(3):
  eng.gemm_profile(enable=True, fetch=False)
  eng.timings(True)
  gp = eng.gp_fit(spec, Xd, yd, noise)
  t = eng.timings(False)
  g = eng.gemm_profile(enable=False, fetch=True)
  gp.free()
  print('chol %.2f ms | ' % t['chol'] + ' | '.join('v%d: %d launches, sum %.2f ms, busy %.2f ms, %.1f TF/s of busy' %
        (v, x['launches'], x['ms'], x['busy_ms'], x['flop'] / (x['busy_ms'] * 1e-3) / 1e12) for v, x in enumerate(g) if x['launches']), flush=True)
# the updates alone, on the Gram matrix itself (same value distribution as the factorisation's first updates)
K = eng.empty((n, n))
eng.kernel_matrix(spec, Xd, None, diag_add=noise, out=K)
def upd(k, Kw=512):
  k0 = 512 * k
  rem = n - k0 - Kw
  A21 = K.offset((k0 + Kw) * n + k0); Cc = K.offset((k0 + Kw) * n + k0 + Kw)
  check(eng.lib.dfh_gemm(eng.ctx, 0, rem, rem, Kw, -1.0, A21, n, A21, n, 1.0, Cc, n, 1))
  return rem * (rem + 1.0) * Kw
for reps in range(3):
  eng.timer_begin()
  fl = sum(upd(k) for k in range(16))
  ms = eng.timer_end()
  print('Gram-matrix data, 16 updates back to back: %.3f ms %.1f TF/s' % (ms, fl / (ms * 1e-3) / 1e12), flush=True)
