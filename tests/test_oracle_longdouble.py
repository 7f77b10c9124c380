"""CPU: the extended-precision truth (oracle/ld_truth.c, x87 long double) against the NumPy oracle
on well-conditioned problems, and against closed forms.  It adjudicates the conditioning sweep
of tests/test_gpu_conditioning.py."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ref_longdouble as T
from oracle import ref_numpy as O


@pytest.mark.parametrize('kind,nu,d', [('se', 0.0, 4), ('matern', 2.5, 3), ('matern', 1.5, 2), ('matern', 0.5, 2)])
def test_truth_agrees_with_oracle_when_well_conditioned(kind, nu, d):
  rs = np.random.RandomState(7)
  n, m = 300, 40
  X = rs.rand(n, d)
  Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
  mean_c, noise, bw = float(np.median(Y)), float(Y.var() / 20), np.linspace(0.3, 0.6, d)
  Xs, U = rs.rand(m, d), rs.randn(m)
  og = O.GPOracle(X, Y, O.KernelSpec(kind, d, float(Y.var()), bw, nu=nu), mean_c, noise)
  mu, cov = og.eval(Xs, 'covar')
  tr = T.gp_truth(kind, bw, float(Y.var()), X, Y - mean_c, noise, Xs, mean_c, float(Y.max()), nu=nu,
                  want_L=True, want_K=True, ts_normals=U, want_cov=True)
  if nu == 0.5:
    # exp(-r) is not smooth at r = 0: the reference's diagonal carries sqrt(cancellation noise of
    # dist_squared) ~ 1e-8 -- the REFERENCE is this far from the truth here
    assert 1e-9 < relerr(og.K_trtr_wo_noise, tr['K']) < 2e-7
    assert relerr(og.alpha, tr['alpha']) < 1e-4 and relerr(mu, tr['mu']) < 1e-6
    return
  assert relerr(og.K_trtr_wo_noise, tr['K']) < 1e-13
  assert relerr(og.L, tr['L']) < 1e-12
  assert relerr(og.alpha, tr['alpha']) < 1e-11
  assert abs(og.lml() - tr['lml']) < 1e-12 * abs(tr['lml'])
  assert relerr(mu, tr['mu']) < 1e-12 and relerr(cov, tr['cov']) < 1e-11
  assert relerr(np.sqrt(np.diag(cov)), tr['sd']) < 1e-11
  assert relerr(O.acq_values('ei', mu, np.sqrt(np.diag(cov)), float(Y.max())), tr['ei']) < 1e-10
  assert relerr(og.draw_samples_blocked(Xs, U, m), tr['draw']) < 1e-9


def test_truth_kernel_known_answers():
  """ the reference's own kernel vectors (gp/unittest_kernel.py:38-53, 82-124): data_1 / data_2,
      bandwidths [0.1, 1], SE scale 2, Matern scale 2.1, squared distances 406.25 / 404 / 0.25 """
  data_1 = np.array([[1, 2], [3, 4.5]])
  data_2 = np.array([[1, 2], [3, 4]])
  bw = np.array([0.1, 1.0])
  d11 = np.array([[0, 406.25], [406.25, 0]])
  d22 = np.array([[0, 404.0], [404.0, 0]])
  d12 = np.array([[0, 404.0], [406.25, 0.25]])
  for A, B, dist in ((data_1, data_1, d11), (data_2, data_2, d22), (data_1, data_2, d12)):
    assert np.linalg.norm(T.kernel_matrix('se', bw, 2.0, A, B) - 2.0 * np.exp(-dist / 2)) < 1e-14
    r = np.sqrt(dist)
    closed = {0.5: np.exp(-r), 1.5: np.exp(-np.sqrt(3) * r) * (1 + np.sqrt(3) * r),
              2.5: np.exp(-np.sqrt(5) * r) * (1 + np.sqrt(5) * r + (5 / 3.0) * r ** 2)}
    for nu, val in closed.items():
      assert np.linalg.norm(T.kernel_matrix('matern', bw, 2.1, A, B, nu=nu) - 2.1 * val) < 1e-14


def test_truth_not_pd_raises():
  X = np.zeros((4, 2))
  with pytest.raises(np.linalg.LinAlgError):
    T.gp_truth('se', np.ones(2), 1.0, X, np.zeros(4), 0.0)
