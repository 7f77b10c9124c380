"""Manual MI355X check beyond BASELINE's sizes (not collected by pytest): fit at n = 32768 and
n = 50000 (not a multiple of the 512 panel), alpha residual on sampled rows against the oracle's
kernel rows, Cholesky rate.  Usage: python tests/gpu_big_n.py"""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from dragonfly_amd.engine import Engine, KernelSpec
from oracle import ref_numpy as O
eng = Engine()
for n in (32768, 50000):
  d = 32
  rs = np.random.RandomState(1)
  X = rs.rand(n, d); Y = (X ** 2).dot((np.arange(d) + 1.0) / d) + 0.01 * rs.randn(n)
  bw = 0.2 * np.sqrt(d) * (0.5 + np.arange(d) / 32.0)
  spec = KernelSpec('se', d, float(Y.var()), bw)
  mean_c, noise = float(np.median(Y)), float(Y.var() / 20)
  t0 = time.time(); gp = eng.gp_fit(spec, X, Y - mean_c, noise); t1 = time.time()
  eng.timings(True); gp2 = eng.gp_fit(spec, X, Y - mean_c, noise); t = eng.timings(True); gp2.free()
  alpha = gp.get_alpha()
  rows = rs.choice(n, 48, replace=False)
  Kr = O.KernelSpec('se', d, float(Y.var()), bw)(X[rows], X)
  res = Kr.dot(alpha) + noise * alpha[rows] - (Y[rows] - mean_c)
  Xs = rs.rand(4096, d)
  bv, bi = gp.acq_argmax('ei', Xs, params=(float(Y.max()), 0.0), mean_const=mean_c)
  print('n=%d fit %.1f ms (chol %.1f ms = %.1f TF/s) lml %.6f jitter %s | alpha residual %.2e | ei argmax %d'
        % (n, (t1 - t0) * 1e3, t['chol'], n ** 3 / 3.0 / t['chol'] / 1e9, gp.lml, gp.jitter_power,
           np.abs(res).max() / np.abs(Y - mean_c).max(), bi), flush=True)
  gp.free()
