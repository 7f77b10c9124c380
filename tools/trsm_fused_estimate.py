"""What one product per block would buy the posterior's row solve (docs/NOTES_r05.md section 8): for every 512-block of
a factor of order n, the two launches of today -- T = B_b - V L_b^T (K = c0) and X_b = T Linv_b^T (K = 512, triangular)
-- against ONE launch with K = c0 + 512 (X_b = [V | B_b] [-Linv_b L_b | Linv_b]^T): the latter timed on the existing GEMM
through dfh_gemm, to be compared with the bench's trsm section for the same m and n."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine
from dragonfly_amd._lib import check
eng = get_engine()
M = int(os.environ.get('ROWS', '65536')); n = int(os.environ.get('ORDER', '4096')); NB = 512
gen = np.random.Generator(np.random.Philox(1))
ld = n + NB
A = eng.empty((M, ld)); G = eng.empty((NB, ld)); T = eng.empty((M, NB))
eng.random_candidates(M, ld, bounds=[[-0.5, 0.5]] * ld, rng=gen, out=A)
eng.random_candidates(NB, ld, bounds=[[-0.5, 0.5]] * ld, rng=gen, out=G)
def timed(f, reps=3):
  f(); ts = []
  for _ in range(reps):
    eng.timer_begin(); f(); ts.append(eng.timer_end())
  return sorted(ts)[len(ts) // 2]
tot = 0.0
for c0 in range(0, n, NB):
  def one():
    check(eng.lib.dfh_gemm(eng.ctx, 0, M, NB, c0 + NB, 1.0, A.ptr, ld, G.ptr, ld, 0.0, T.ptr, NB, 0))
  t1 = timed(one)
  tot += t1
  print('c0 %5d: one launch, K = %5d: %7.3f ms  %5.1f TF/s' % (c0, c0 + NB, t1, 2.0 * M * NB * (c0 + NB) / t1 / 1e9), flush=True)
fl = float(M) * n * n
print('m = %d, n = %d: sum of the one-launch blocks %.3f ms = %.3f of 78.6 TF/s by m n^2 (the K range not yet cut to the triangle)'
      % (M, n, tot, fl / tot / 1e9 / 78.6e3))
