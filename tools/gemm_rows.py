"""Posterior-TRSM-shaped GEMM launches (M x 512 x K, C -= A.B^T): time per launch for one tree
(DFH_ROOT) and one DFH_GEMM_SPLIT_ROWS setting.  Used with rocprofv3 --pmc FETCH_SIZE as well."""
import os, sys
root = os.environ.get('DFH_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
from dragonfly_amd.engine import get_engine
from dragonfly_amd._lib import check
eng = get_engine()
M = int(os.environ.get('ROWS', '262144'))
N = int(os.environ.get('COLS', '512'))
PAD = int(os.environ.get('PAD', '0'))      # leading dimension = K + PAD
reps = int(os.environ.get('REPS', '3'))
for K in [int(k) for k in os.environ.get('KS', '4096,8192,16384').split(',')]:
  L = K + PAD
  A = eng.empty((M, L)); B = eng.empty((N, L)); Cd = eng.empty((M, N))
  gen = np.random.Generator(np.random.Philox(K))
  eng.random_candidates(M, L, bounds=[[-0.5, 0.5]] * L, rng=gen, out=A)
  eng.random_candidates(N, L, bounds=[[-0.5, 0.5]] * L, rng=gen, out=B)
  eng.random_candidates(M, N, rng=gen, out=Cd)
  def run():
    check(eng.lib.dfh_gemm(eng.ctx, 0, M, N, K, -1.0, A.ptr, L, B.ptr, L, 1.0, Cd.ptr, N, 0))
  run()
  ts = []
  for _ in range(reps):
    eng.timer_begin(); run(); ts.append(eng.timer_end())
  ms = sorted(ts)[len(ts) // 2]
  print('%s split=%s pad=%d  %d x %d x %d: %8.3f ms  %5.1f TF/s' % (os.path.basename(root), os.environ.get('DFH_GEMM_SPLIT_ROWS', '0'),
        PAD, M, N, K, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12), flush=True)
  A.free(); B.free(); Cd.free()
