"""MI355X: the projection onto the PSD cone (project_symmetric_to_psd_cone, general_utils.py:150-163)
without an eigen-decomposition -- matrix sign function by Newton-Schulz steps on the fp64 MFMA GEMM
(csrc/psdproj.hip) -- against the oracle's eigh route, and the GP / Cartesian-product GP classes for
kernels that are not guaranteed PSD against the real reference's outputs (gp_core.py:827-857,
cartesian_product_gp.py:208-248)."""
import numpy as np
import pytest

from conftest import relerr
from nonpsd_replay import check
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n', [1, 2, 7, 64, 129, 500, 1500])
@pytest.mark.parametrize('eps', [0.0, 0.03])
def test_projection_matches_the_eigh_route(engine, n, eps):
  rs = np.random.RandomState(n)
  A = rs.randn(n, n)
  A = (A + A.T) / 2                                   # indefinite: half the spectrum is clipped
  want = O.project_symmetric_to_psd_cone(A, epsilon=eps)
  got = engine.project_psd(A, epsilon=eps)
  assert relerr(got, want) < 1e-11
  assert np.array_equal(got, got.T)
  # a matrix that is already inside the cone comes back unchanged (to rounding)
  B = A.dot(A.T) / n + (eps + 0.5) * np.eye(n)
  assert relerr(engine.project_psd(B, epsilon=eps), B) < 1e-12
  # eigenvalues spread over many orders of magnitude, some barely negative
  w = np.concatenate([np.logspace(-12, 0, n - n // 3), -np.logspace(-14, -1, n // 3)]) if n >= 3 else np.array([1.0, -1e-9][:n])
  Q, _ = np.linalg.qr(rs.randn(n, n))
  Cm = (Q * w).dot(Q.T)
  Cm = (Cm + Cm.T) / 2
  assert relerr(engine.project_psd(Cm, epsilon=eps), O.project_symmetric_to_psd_cone(Cm, epsilon=eps)) < 1e-10


def test_fit_gram_branches_against_the_oracle(engine):
  """ _get_cholesky_decomp (gp_core.py:827-847): the three branches through dfh_gp_fit_gram """
  rs = np.random.RandomState(8)
  n = 300
  A = rs.randn(n, n)
  K_indef = (A + A.T) / 2
  K_psd = A.dot(A.T) / n
  y = rs.randn(n)
  for K, mode in ((K_indef, 'project_first'), (K_indef, 'try_before_project'), (K_psd, 'try_before_project'),
                  (K_psd, 'project_first'), (K_psd, 'guaranteed_psd')):
    L = O.get_cholesky_decomp(K, 0.1, mode)
    alpha = O.solve_upper_triangular(L.T, O.solve_lower_triangular(L, y))
    gp = engine.gp_fit_gram(K, y, 0.1, handle_non_psd_kernels=mode)
    assert relerr(gp.get_L(), L) < 1e-10 and relerr(gp.get_alpha(), alpha) < 1e-10
    gp.free()
  with pytest.raises(np.linalg.LinAlgError):
    engine.gp_fit_gram(K_indef, y, 0.1, allow_jitter=False)
  with pytest.raises(ValueError):
    engine.gp_fit_gram(K_psd, y, 0.1, handle_non_psd_kernels='something_else')


def test_nonpsd_gp_and_cpgp_against_reference_outputs(engine):
  check(tol=1e-10)


def test_projection_stops_early_and_rejects_non_finite_input(engine):
  """ round 3 (advisor): a spectrum bounded away from zero converges in ~25 Newton-Schulz steps -- the
      iteration notices -- with the result of the eigen-decomposition route all the same; NaN / Inf
      entries are an error (np.linalg.eigh raises LinAlgError in the reference's route) """
  rs = np.random.RandomState(4)
  n = 400
  Q, _ = np.linalg.qr(rs.randn(n, n))
  lam = np.concatenate([rs.uniform(0.5, 2.0, n // 2), -rs.uniform(0.5, 2.0, n - n // 2)])
  M = (Q * lam).dot(Q.T)
  M = (M + M.T) / 2
  want = O.project_symmetric_to_psd_cone(M, 0.05)
  got = engine.project_psd(M, 0.05)
  assert relerr(got, want) < 1e-12 and np.array_equal(got, got.T)
  bad = M.copy()
  bad[3, 7] = bad[7, 3] = np.nan
  with pytest.raises(np.linalg.LinAlgError):
    engine.project_psd(bad, 0.0)
  bad[3, 7] = bad[7, 3] = np.inf
  with pytest.raises(np.linalg.LinAlgError):
    engine.project_psd(bad, 0.0)
