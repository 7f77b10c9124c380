"""Times C[n x n] -= A[n x k] A^T (lower tiles only): the trailing update of the factorisation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine
eng = get_engine()
for n, k, ld in ((15872, 512, 16384), (12288, 512, 16384), (8192, 512, 16384), (4096, 16384, 16384)):
  rs = np.random.RandomState(0)
  # operands live inside a 16384-wide matrix as in the factorisation (row stride ld)
  buf = eng.to_device(rs.rand(max(n, 1) * ld).reshape(n, ld) * 1e-3)
  C = eng.to_device(rs.rand(n, ld) * 1e-3) if k < 4096 else eng.to_device(rs.rand(n, n) * 1e-3)
  ldc = ld if k < 4096 else n
  import ctypes as Ct
  from dragonfly_amd._lib import check
  def run():
    check(eng.lib.dfh_gemm(eng.ctx, 0, n, n, k, -1.0, buf.ptr, ld, buf.ptr, ld, 1.0, C.ptr, ldc, 1))
  run()
  ms = []
  for _ in range(5):
    eng.timer_begin(); run(); ms.append(eng.timer_end())
  m = sorted(ms)[2]
  print('SYRK n=%5d k=%5d: %.3f ms  %.1f TF/s' % (n, k, m, n * (n + 1.0) * k / (m * 1e-3) / 1e12))
  buf.free(); C.free()
