"""MI355X: GPs whose kernel the host evaluates: any positive semi-definite
Kernel object without a device description, or a subclass overriding the documented hook
GP._get_training_kernel_matrix (gp_core.py:149).  Gram / cross matrices come from the caller, the
factorisation, solves and posterior run on the device (dfh_gp_fit_gram, dfh_gp_predict_gram)."""
from argparse import Namespace

import numpy as np
import pytest

from conftest import relerr
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


class HostPolyKernel(object):
  """ A host-side polynomial kernel in the reference's Kernel protocol (callable, hyperparams,
      is_guaranteed_psd), as gp/kernel.py:331-371 defines it: scale * (1 + <x/s, y/s>)**order. """

  def __init__(self, dim, order, scale, dim_scalings):
    self.dim = dim
    self.hyperparams = dict(order=order, scale=scale, dim_scalings=np.asarray(dim_scalings, dtype=float))

  def is_guaranteed_psd(self):
    return True

  def __call__(self, X1, X2=None):
    X2 = X1 if X2 is None else X2
    if len(X1) == 0 or len(X2) == 0:
      return np.zeros((len(X1), len(X2)))
    A = np.asarray(X1, dtype=float) / self.hyperparams['dim_scalings']
    B = np.asarray(X2, dtype=float) / self.hyperparams['dim_scalings']
    return self.hyperparams['scale'] * (1.0 + A.dot(B.T)) ** self.hyperparams['order']


def _problem(n=260, d=3, seed=4):
  rs = np.random.RandomState(seed)
  X = rs.rand(n, d)
  Y = (X ** 2).sum(axis=1) - X[:, 0] * X[:, 1] + 0.05 * rs.randn(n)
  return rs, X, Y


def test_host_kernel_gp_matches_oracle(engine):
  from dragonfly_amd.gp_core import GP
  rs, X, Y = _problem()
  kern = HostPolyKernel(3, 3, 0.7, [0.8, 1.1, 1.4])
  mean_c, noise = float(np.mean(Y)), 0.01
  gp = GP(list(X), list(Y), kern, lambda x: np.array([mean_c] * len(x)), noise)
  assert gp._generic and gp.device_gp.spec is None
  og = O.GPOracle(X, Y, kern, mean_c, noise)
  assert relerr(gp.alpha, og.alpha) < TOL and relerr(np.tril(gp.L), og.L) < TOL
  assert abs(gp.compute_log_marginal_likelihood() - og.lml()) <= TOL * abs(og.lml())
  assert np.array_equal(gp.K_trtr_wo_noise, kern(X, X))
  Xs = rs.rand(777, 3)
  mu, sd = gp.eval(Xs, 'std')
  mur, sdr = og.eval(Xs, 'std')
  # sd = sqrt(k** - |L^-1 k*|^2) cancels; the bound is twice the oracle's distance from the same linear
  # algebra in extended precision on the same Gram matrices (tests/truth_bounds.py)
  from truth_bounds import gram_bounds
  Kxx = kern(X, X)
  b = gram_bounds(Kxx, noise, Y - mean_c, dict(mu=mur, sd=sdr), kern(Xs, X), np.diag(kern(Xs, Xs)), mean_const=mean_c)
  assert relerr(mu, mur) <= b['mu'] and relerr(sd, sdr) <= b['sd'], (relerr(sd, sdr), b)
  mu0, none = gp.eval(Xs)
  assert none is None and relerr(mu0, mur) < TOL
  _, cov = gp.eval(Xs[:90], 'covar')
  _, covr = og.eval(Xs[:90], 'covar')
  bc = gram_bounds(Kxx, noise, Y - mean_c, dict(cov=covr), kern(Xs[:90], X), np.diag(kern(Xs[:90], Xs[:90])),
                   kern(Xs[:90], Xs[:90]))
  assert relerr(cov, covr) <= bc['cov'], (relerr(cov, covr), bc)
  Xh = rs.rand(5, 3)
  _, sdh = gp.eval_with_hallucinated_observations(Xs[:200], Xh, 'std')
  _, sdhr = og.eval_with_hallucinated_observations(Xs[:200], Xh, 'std')
  Xa = np.concatenate([X, Xh], axis=0)
  bh = gram_bounds(kern(Xa, Xa), noise, np.zeros(len(Xa)), dict(sd=sdhr), kern(Xs[:200], Xa), np.diag(kern(Xs[:200], Xs[:200])))
  assert relerr(sdh, sdhr) <= bh['sd'], (relerr(sdh, sdhr), bh)
  # adding data rebuilds (no kernel on the device to append with)
  gp.add_data_multiple([rs.rand(3)], [0.3])
  assert gp.num_tr_data == len(Y) + 1 and gp.device_gp.n == len(Y) + 1
  np.random.seed(3)
  s = gp.draw_samples(2, Xs[:40])
  assert s.shape == (2, 40) and np.all(np.isfinite(s))


def test_overridden_training_kernel_hook_is_honoured(engine):
  """ gp_core.py:149-153: a subclass may supply the training Gram matrix itself """
  from dragonfly_amd.gp_core import GP
  from dragonfly_amd import kernel as K
  rs, X, Y = _problem(n=150)
  bump = 0.05

  class BumpedGP(GP):
    def _get_training_kernel_matrix(self):
      return self.kernel(self.X, self.X) + bump * np.eye(len(self.X))
    def _child_str(self):
      return 'bumped'

  kern = K.SEKernel(3, 1.3, [0.5, 0.6, 0.7])
  gp = BumpedGP(list(X), list(Y), kern, lambda x: np.zeros(len(x)), 0.02)
  ok = O.KernelSpec('se', 3, 1.3, [0.5, 0.6, 0.7])
  og = O.GPOracle(X, Y, ok, 0.0, 0.02 + bump)        # K + bump I + noise I
  assert gp._generic and relerr(gp.alpha, og.alpha) < TOL
  mu, _ = gp.eval(rs.rand(50, 3))
  assert mu.shape == (50,)


def test_acquisitions_take_the_closure_route_for_host_kernels(engine):
  from dragonfly_amd.gp_core import GP
  from dragonfly_amd import gpb_acquisitions as A
  from dragonfly_amd.oper_utils import EuclideanDomain
  rs, X, Y = _problem(n=120)
  kern = HostPolyKernel(3, 2, 1.0, [1.0, 1.0, 1.0])
  gp = GP(list(X), list(Y), kern, lambda x: np.zeros(len(x)), 0.05)
  anc = Namespace(max_evals=500, t=len(Y), domain=EuclideanDomain([[0, 1]] * 3), acq_opt_method='rand',
                  curr_max_val=float(max(Y)), handle_parallel='halluc', eval_points_in_progress=[], is_mf=False)
  for acq in ('ucb', 'ei', 'pi', 'ttei', 'ts'):
    np.random.seed(21)
    x = np.asarray(getattr(A.asy, acq)(gp, anc), dtype=float)
    assert x.shape == (3,) and np.all((x >= 0) & (x <= 1))
  # UCB: the same point as evaluating the formula on the oracle's posterior with the same draws
  np.random.seed(21)
  x = np.asarray(A.asy.ucb(gp, anc), dtype=float)
  np.random.seed(21)
  cands = np.random.random((500, 3))
  og = O.GPOracle(X, Y, kern, 0.0, 0.05)
  mu, sd = og.eval(cands, 'std')
  beta = O.ucb_beta_th(3, len(Y))
  assert np.array_equal(x, cands[int(np.argmax(mu + beta * sd))])


def test_product_kernel_with_a_host_factor_runs_in_host_kernel_mode(engine):
  """ multi-fidelity GP whose fidelity kernel the device does not evaluate (poly): the product is
      composed on the host from its factors, the posterior still runs on the device """
  from dragonfly_amd.mf_gp import EuclideanMFGP
  from dragonfly_amd import kernel as K
  rs = np.random.RandomState(12)
  n, fd, dd = 90, 1, 2
  ZZ, XX = rs.rand(n, fd), rs.rand(n, dd)
  YY = np.sin(3 * XX.sum(axis=1)) * (0.5 + ZZ[:, 0]) + 0.05 * rs.randn(n)
  fidel = HostPolyKernel(fd, 2, 1.0, [0.9])
  domain = K.SEKernel(dd, 1.0, [0.4, 0.6])
  gp = EuclideanMFGP(list(ZZ), list(XX), list(YY), None, 1.5, fidel, domain, lambda x: np.zeros(len(x)), 0.02)
  assert gp._generic and not gp.kernel.has_device_spec()
  ZX = np.concatenate([ZZ, XX], axis=1)
  ok = lambda A, B=None: 1.5 * fidel(A[:, :fd], (A if B is None else B)[:, :fd]) * \
      O.se_kernel(A[:, fd:], (A if B is None else B)[:, fd:], 1.0, np.array([0.4, 0.6]))

  class _Spec(object):
    def __call__(self, A, B=None):
      A = np.asarray(A, dtype=float)
      return ok(A, None if B is None else np.asarray(B, dtype=float))
  og = O.GPOracle(ZX, YY, _Spec(), 0.0, 0.02)
  assert relerr(gp.alpha, og.alpha) < TOL
  Zs, Xs = rs.rand(40, fd), rs.rand(40, dd)
  mu, sd = gp.eval_at_fidel(list(Zs), list(Xs), 'std')
  ZXs = np.concatenate([Zs, Xs], axis=1)
  mur, sdr = og.eval(ZXs, 'std')
  from truth_bounds import gram_bounds
  b = gram_bounds(ok(ZX), 0.02, YY, dict(mu=mur, sd=sdr), ok(ZXs, ZX), np.diag(ok(ZXs)))
  assert relerr(mu, mur) <= b['mu'] and relerr(sd, sdr) <= b['sd'], (relerr(sd, sdr), b)
