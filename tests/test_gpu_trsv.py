"""MI355X: the two substitutions on their own kernels (csrc/chol.hip: k_trsv_blk_fwd / k_trsv_upd_fwd / k_trsv_blk_bwd /
k_trsv_upd_bwd, round 6; solve_lower_triangular / solve_upper_triangular of dragonfly/utils/general_utils.py:208-221 as
GP.build_posterior uses them, gp_core.py:161-163) against scipy.linalg.solve_triangular: sizes around the 512-block edges
and the kernels' own edges (a last block of 1 .. 511 rows, panels shorter than the 2048 rows from which the forward update
takes eight rows per wave, fewer than 4096 columns where the backward update takes 16 columns per workgroup), both
directions through the C-ABI's dfh_solve_triangular, and alpha of a fit against the oracle at the same sizes."""
import numpy as np
import pytest
from scipy.linalg import solve_triangular

from conftest import relerr
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [1, 2, 63, 511, 512, 513, 1000, 1024, 1025, 1536, 2047, 2049, 2561, 3000, 4097, 5121])
def test_single_right_hand_side_both_directions(engine, n):
  rs = np.random.RandomState(n)
  A = rs.rand(n, min(n, 40))
  K = A.dot(A.T) / A.shape[1] + (0.5 + rs.rand(n)) * np.eye(n)        # well conditioned, SPD
  L = np.linalg.cholesky(K)
  b = rs.randn(n)
  x = engine.solve_triangular(L, b, upper=False)
  assert relerr(x, solve_triangular(L, b, lower=True)) < 1e-11
  xt = engine.solve_triangular(L, b, upper=True)                       # L^T x = b
  assert relerr(xt, solve_triangular(L.T, b, lower=False)) < 1e-11


@pytest.mark.parametrize('n', [513, 1025, 2049, 2600, 4608])
def test_alpha_of_a_fit_against_the_oracle(engine, n):
  from dragonfly_amd.engine import KernelSpec
  rs = np.random.RandomState(7 * n)
  d = 4
  X = rs.rand(n, d)
  Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
  bw = 0.3 + 0.4 * rs.rand(d)
  noise = float(Y.var() / 25)
  gp = engine.gp_fit(KernelSpec('matern', d, float(Y.var()), bw, nu=2.5), X, Y - 0.2, noise)
  og = O.GPOracle(X, Y, O.KernelSpec('matern', d, float(Y.var()), bw, nu=2.5), 0.2, noise)
  assert relerr(gp.get_alpha(), og.alpha) < 1e-10 and abs(gp.lml - og.lml()) <= 1e-10 * abs(og.lml())
  gp.free()
