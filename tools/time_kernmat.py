"""Times the kernel-matrix kernels through the library (section timers = HIP events around the kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
rs = np.random.RandomState(0)
def run(kind, nu, n1, n2, d, reps=5):
  bw = 0.2 * np.sqrt(d) * (0.5 + np.arange(d) / float(d)) if kind == 'se' else 0.5 * np.ones(d)
  spec = KernelSpec(kind, d, 0.37, bw, nu=nu)
  X1 = eng.to_device(rs.rand(n1, d)); X2 = None if n2 is None else eng.to_device(rs.rand(n2, d))
  out = eng.empty((n1, n1 if n2 is None else n2))
  eng.kernel_matrix(spec, X1, X2, out=out)
  eng.timings(True)
  for _ in range(reps):
    eng.kernel_matrix(spec, X1, X2, out=out)
  t = eng.timings(False)
  ms = (t['kernmat'] + t['cross']) / reps
  m2 = n1 if n2 is None else n2
  gb = 8.0 * (n1 * m2 + (n1 + m2) * d) / 1e9
  print('%-7s nu=%.1f %6d x %6s d=%2d: %7.3f ms  %5.2f TB/s = %4.1f%% of 8 TB/s' % (kind, nu, n1, 'sym' if n2 is None else n2, d, ms, gb / ms, gb / ms / 8 * 100))
  for a in (X1, X2, out):
    if a is not None: a.free()
run('se', 0, 32768, 16384, 32)
run('matern', 2.5, 32768, 16384, 32)
run('se', 0, 65536, 4096, 6)
run('matern', 2.5, 65536, 4096, 6)
run('matern', 0.5, 65536, 4096, 6)
run('se', 0, 16384, None, 32)
run('matern', 2.5, 16384, None, 32)
run('matern', 2.5, 16384, None, 6)
run('se', 0, 4096, None, 6)
run('matern', 2.5, 4096, None, 6)
