"""Correctness + timing of a few GEMM shapes (A/B of kernel variants): DFH_ROOT selects the tree."""
import os, sys, time
root = os.environ.get('DFH_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
from dragonfly_amd.engine import get_engine
from dragonfly_amd import _lib
eng = get_engine()
rs = np.random.RandomState(0)
for (M, N, K) in ((300, 200, 100), (1000, 520, 777), (2048, 2048, 512)):
  A, B = rs.rand(M, K) - 0.5, rs.rand(N, K) - 0.5
  C = eng.gemm(A, B)
  err = np.max(np.abs(C - A.dot(B.T))) / K
  print('check %d x %d x %d: max err / K = %.2e' % (M, N, K, err))
  assert err < 1e-15
for (M, N, K, lower) in ((32768, 512, 8192, 0), (8192, 8192, 8192, 0), (15872, 15872, 512, 1), (65536, 512, 2048, 0)):
  A = eng.to_device(rs.rand(M, K) - 0.5)
  B = A if lower else eng.to_device(rs.rand(N, K) - 0.5)
  Cd = eng.empty((M, N))
  from dragonfly_amd._lib import check
  def run():
    check(eng.lib.dfh_gemm(eng.ctx, 0, M, N, K, -1.0, A.ptr, K, B.ptr, K, 1.0, Cd.ptr, N, lower))
  run()
  ts = []
  for _ in range(5):
    eng.timer_begin(); run(); ts.append(eng.timer_end())
  ms = sorted(ts)[2]
  fl = (M * (M + 1.0) * K) if lower else 2.0 * M * N * K
  print('%6d x %6d x %6d %s: %8.3f ms  %5.1f TF/s' % (M, N, K, 'lower' if lower else '     ', ms, fl / (ms * 1e-3) / 1e12))
  A.free(); Cd.free()
  if not lower: B.free()
