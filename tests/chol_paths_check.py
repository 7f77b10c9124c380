"""Run in a subprocess by tests/test_gpu_chol_paths.py with the factorisation's schedule switches set
in the environment (they are read once per process): factors SPD matrices of assorted sizes -- single
and as lock-step batches through the tuning objective -- and compares with LAPACK / one-at-a-time
fits.  Prints OK on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dragonfly_amd import general_utils as G                  # noqa: E402
from dragonfly_amd.engine import KernelSpec, get_engine       # noqa: E402
from oracle import ref_numpy as O                              # noqa: E402


def relerr(a, b):
  return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


def main():
  eng = get_engine()
  for n in (513, 1024, 1100, 1536, 1601, 2048, 2500, 3100):
    rs = np.random.RandomState(n)
    X = rs.rand(n, 4)
    M = O.se_kernel(X, X, 1.0, np.full(4, 0.4)) + 0.05 * np.eye(n)
    L = G.stable_cholesky(M)
    Lr = np.linalg.cholesky(M)
    assert relerr(L, Lr) < 1e-11 and np.array_equal(np.triu(L, 1), np.zeros_like(L)), n
  # failures inside full 512-panels: a negative pivot in the second strip of the third panel, a NaN
  # further down; the jitter ladder on a rank-deficient 1300 x 1300 matrix picks NumPy's power
  rs = np.random.RandomState(5)
  n = 1700
  X = rs.rand(n, 3)
  M = O.se_kernel(X, X, 1.0, np.full(3, 0.4)) + 0.05 * np.eye(n)
  for bad_at, bad_val in ((1100, -1.0), (40, -1.0), (1300, np.nan)):
    Mb = M.copy(); Mb[bad_at, bad_at] = bad_val
    try:
      G.stable_cholesky(Mb, add_to_diag_till_psd=False)
      raise AssertionError('no error for a bad pivot at %d' % bad_at)
    except np.linalg.LinAlgError as e:
      # the first failing pivot, 1-based; a NaN entry makes the first pivot whose column it reaches fail
      if bad_val < 0:
        assert 'pivot %d' % (bad_at + 1) in str(e), str(e)
  A = rs.randn(1300, 40)
  Mr = A.dot(A.T)
  L, p = eng.stable_cholesky(Mr, return_power=True)
  _, pr = O.stable_cholesky(Mr, return_power=True)
  assert p == pr and relerr(L.dot(L.T), Mr) < 1e-7, (p, pr)
  # lock-step batches: 9 candidate kernels on n = 1300 / 2200 points against single fits
  for n in (1300, 2200):
    rs = np.random.RandomState(n)
    d = 3
    X = rs.rand(n, d)
    Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
    specs = [KernelSpec('se', d, float(Y.var()) * (0.5 + rs.rand()), 0.2 + 0.6 * rs.rand(d)) for _ in range(9)]
    means = [float(0.1 * rs.randn()) for _ in specs]
    noises = [float(Y.var() * (0.01 + 0.1 * rs.rand())) for _ in specs]
    lml = eng.gp_lml_batch(specs, X, Y, means, noises)
    for c in (0, 4, 8):
      one = eng.gp_fit(specs[c], X, Y - means[c], noises[c])
      assert abs(lml[c] - one.lml) <= 1e-11 * abs(one.lml), (n, c, lml[c], one.lml)
      one.free()
  print('OK')


if __name__ == '__main__':
  main()
