"""What the K = 512 trailing update's per-tile overhead is made of: the lower-triangular product
C (-)= A A^T at n = 15872 with and without reading C (beta), with a padded leading dimension, and the
full-square product for comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine
from dragonfly_amd._lib import check
eng = get_engine()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15872
for LD in (16384,):
  gen = np.random.Generator(np.random.Philox(7))
  A = eng.empty((n, LD)); Cd = eng.empty((n, LD))
  eng.random_candidates(n, LD, bounds=[[-0.5, 0.5]] * LD, rng=gen, out=A)
  for K in (512, 1024):
    for beta, lower in ((1.0, 1), (0.0, 1), (1.0, 0)):
      def run():
        check(eng.lib.dfh_gemm(eng.ctx, 0, n, n, K, -1.0, A.ptr, LD, A.ptr, LD, beta, Cd.ptr, LD, lower))
      run()
      ts = []
      for _ in range(5):
        eng.timer_begin(); run(); ts.append(eng.timer_end())
      ms = sorted(ts)[2]
      fl = n * (n + 1.0) * K if lower else 2.0 * n * n * K
      tiles = (n // 128) * (n // 128 + 1) // 2 if lower else (n // 128) ** 2
      print('LD=%d K=%4d beta=%.0f lower=%d: %7.3f ms %5.1f TF/s  per tile-slot %.1f us' % (LD, K, beta, lower, ms, fl / (ms * 1e-3) / 1e12, ms * 1e3 * 512 / tiles), flush=True)
  A.free(); Cd.free()

# the factorisation's sequence of trailing updates, in place in ONE n x n matrix (operand panel and
# C in the same array, ld = n), back to back: is the in-situ update slower than the isolated one?
N = 16384
Mx = eng.empty((N, N))
gen = np.random.Generator(np.random.Philox(9))
eng.random_candidates(N, N, bounds=[[-0.5, 0.5]] * N, rng=gen, out=Mx)
def upd(k, K=512):
  k0 = 512 * k
  rem = N - k0 - K
  A21 = Mx.offset((k0 + K) * N + k0)
  C = Mx.offset((k0 + K) * N + k0 + K)
  check(eng.lib.dfh_gemm(eng.ctx, 0, rem, rem, K, -1.0, A21, N, A21, N, 1.0, C, N, 1))
  return rem * (rem + 1.0) * K
for k in (0, 4, 8, 12):
  upd(k)
  ts = []
  for _ in range(5):
    eng.timer_begin(); fl = upd(k); ts.append(eng.timer_end())
  ms = sorted(ts)[2]
  print('in situ, alone: panel %2d rem=%5d: %.3f ms %.1f TF/s' % (k, N - 512 * (k + 1), ms, fl / (ms * 1e-3) / 1e12), flush=True)
for reps in range(3):
  eng.timer_begin()
  fl = sum(upd(k) for k in range(16))
  ms = eng.timer_end()
  print('in situ, 16 updates back to back: %.3f ms %.1f TF/s' % (ms, fl / (ms * 1e-3) / 1e12), flush=True)
