"""Lower-triangular SYRK-shaped updates C -= A A^T (the factorisation's trailing update) at
different panel widths K: time and TF/s (DFH_ROOT selects the tree)."""
import os, sys
root = os.environ.get('DFH_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
from dragonfly_amd.engine import get_engine
from dragonfly_amd._lib import check
eng = get_engine()
LD = 16384
for n in (15872, 12288, 8192, 4096):
  for K in (512, 1024, 2048):
    gen = np.random.Generator(np.random.Philox(n + K))
    A = eng.empty((n, LD)); Cd = eng.empty((n, LD))
    eng.random_candidates(n, LD, bounds=[[-0.5, 0.5]] * LD, rng=gen, out=A)
    def run():
      check(eng.lib.dfh_gemm(eng.ctx, 0, n, n, K, -1.0, A.ptr, LD, A.ptr, LD, 1.0, Cd.ptr, LD, 1))
    run()
    ts = []
    for _ in range(5):
      eng.timer_begin(); run(); ts.append(eng.timer_end())
    ms = sorted(ts)[2]
    print('n=%6d K=%5d: %7.3f ms  %5.1f TF/s   (per 512 columns: %.3f ms)' % (n, K, ms, n * (n + 1.0) * K / (ms * 1e-3) / 1e12, ms * 512 / K), flush=True)
    A.free(); Cd.free()
