"""MI355X: the polynomial and exponential-decay kernels on the device (DFH_KERNEL_POLY /
DFH_KERNEL_EXPDECAY, alone and as factors of the product kernel), the per-candidate prior variance
their posteriors need, against the real reference's outputs and the oracle."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ref_numpy as O
from polyexp_replay import check

pytestmark = pytest.mark.gpu


def test_kernels_gp_and_mf_gp_against_reference_outputs(engine):
  check(tol=1e-10)


def test_reference_known_answers_on_device(engine):
  """ gp/unittest_kernel.py:126-151 (polynomial kernel; SE x polynomial over the same coordinates) """
  from dragonfly_amd import kernel as K
  from test_polyexp_cpu import known_answers
  known_answers(K, comb_rel_tol=1e-15)


@pytest.mark.parametrize('n1,n2,d', [(1, 1, 1), (130, 67, 5), (300, 515, 8)])
def test_kernel_matrices_against_the_oracle(engine, n1, n2, d):
  """ ragged sizes (tile edges), symmetric and cross, both kinds alone and inside a product """
  from dragonfly_amd import kernel as K
  rs = np.random.RandomState(n1 + d)
  X1, X2 = rs.random_sample((n1, d)), rs.random_sample((n2, d))
  sc, pw = rs.random_sample(d) + 0.3, 3 * rs.random_sample(d) + 0.1
  pairs = [(K.PolyKernel(d, 4, 0.8, sc), O.KernelSpec('poly', d, 0.8, sc, nu=4)),
           (K.ExpDecayKernel(d, 1.1, 0.05, pw), O.KernelSpec('expdecay', d, 1.1, pw, nu=0.05))]
  if d >= 5:
    g0, g1, g2 = [0, 3], [1, 2], list(range(4, d))
    bw = rs.random_sample(len(g2)) + 0.4
    pairs.append((K.CoordinateProductKernel(d, 1.9, [K.ExpDecayKernel(2, 1.0, 0.1, pw[:2]), K.PolyKernel(2, 2, 1.0, sc[:2]),
                                                      K.MaternKernel(len(g2), 1.5, 1.0, bw)], [g0, g1, g2]),
                  O.KernelSpec('product', d, 1.9, groups=[g0, g1, g2],
                               subs=[O.KernelSpec('expdecay', 2, 1.0, pw[:2], nu=0.1), O.KernelSpec('poly', 2, 1.0, sc[:2], nu=2),
                                     O.KernelSpec('matern', len(g2), 1.0, bw, nu=2.5 - 1.0)])))
  for kern, spec in pairs:
    assert relerr(kern(X1, X2), spec(X1, X2)) < 1e-13
    assert relerr(kern(X1), spec(X1)) < 1e-13


def test_posterior_and_acquisitions_with_a_non_stationary_kernel(engine):
  """ fused posterior + EI / UCB arg-max and a joint Thompson block with per-candidate prior variances """
  from dragonfly_amd import kernel as K
  from dragonfly_amd.gp_core import GP
  rs = np.random.RandomState(5)
  n, fd, dd, m = 700, 1, 4, 5000
  X = rs.random_sample((n, fd + dd))
  Y = np.sin(3 * X[:, 1:].sum(axis=1)) * (1 - 0.5 / (1 + 4 * X[:, 0])) + 0.05 * rs.randn(n)
  pw, bw = np.array([1.7]), np.full(dd, 0.5)
  mean_c, noise = float(np.median(Y)), float(Y.var() / 20)
  kern = K.CoordinateProductKernel(fd + dd, float(Y.var()), [K.ExpDecayKernel(fd, 1.0, 0.2, pw), K.SEKernel(dd, 1.0, bw)],
                                   [[0], [1, 2, 3, 4]])
  spec = O.KernelSpec('product', fd + dd, float(Y.var()), groups=[[0], [1, 2, 3, 4]],
                      subs=[O.KernelSpec('expdecay', fd, 1.0, pw, nu=0.2), O.KernelSpec('se', dd, 1.0, bw)])
  gp = GP(list(X), list(Y), kern, lambda x: np.array([mean_c] * len(x)), noise)
  og = O.GPOracle(X, Y, spec, mean_c, noise)
  Xs = rs.random_sample((m, fd + dd))
  mu, sd = gp.eval(Xs, 'std')
  mur, sdr = og.eval(Xs, 'std')
  assert relerr(gp.alpha, og.alpha) < 1e-10 and relerr(mu, mur) < 1e-10 and relerr(sd, sdr) < 1e-10
  prior = np.diag(spec(Xs[:50]))
  assert prior.max() / prior.min() > 1.2                 # the prior variance really varies
  for acq, params in (('ei', (float(Y.max()), 0.0)), ('ucb', (2.5, 0.0))):
    bv, bi, vals = gp.device_gp.acq_argmax(acq, Xs, params=params, mean_const=mean_c, return_vals=True)
    want = O.acq_values(acq, mur, sdr, params[0])
    assert relerr(vals, want) < 1e-10 and bi == int(np.argmax(want))
  U = rs.randn(1024)
  _, ti, samp, _ = gp.device_gp.thompson(Xs[:1024], U, block=512, mean_const=mean_c, return_samples=True)
  want = og.draw_samples_blocked(Xs[:1024], U, 512)
  from truth_bounds import draw_bound
  for b in range(2):                 # per block: twice the oracle draw's distance from the extended-precision draw
    sl = slice(512 * b, 512 * (b + 1))
    mu_b, cov_b = og.eval(Xs[sl], 'covar')
    tol_b = draw_bound(mu_b, cov_b, U[sl], want[sl])
    assert relerr(samp[sl], want[sl]) <= tol_b, (b, relerr(samp[sl], want[sl]), tol_b)
  assert ti == int(np.argmax(want))


def test_bad_descriptions_are_rejected(engine):
  from dragonfly_amd import kernel as K
  X = np.random.RandomState(0).random_sample((5, 9))
  with pytest.raises(ValueError):
    K.ExpDecayKernel(9, 1.0, 0.1, np.ones(9))(X)           # more than 8 dimensions
  with pytest.raises(ValueError):
    K.PolyKernel(9, 2.5, 1.0, np.ones(9))(X)               # the order has to be an integer
  add = K.AdditiveKernel(1.0, [K.ExpDecayKernel(4, 1.0, 0.1, np.ones(4)), K.SEKernel(5, 1.0, np.ones(5))],
                         [[0, 1, 2, 3], [4, 5, 6, 7, 8]])
  assert not add.has_device_spec()                         # an exponential-decay factor only inside a product


def test_polynomial_groups_in_an_additive_kernel(engine):
  """ the reference's factory builds additive kernels of polynomial groups (euclidean_gp.py:870-879,
      kernel.py:461-501): Gram / cross matrices, the posterior with its per-candidate prior variance and
      the add-UCB group acquisitions (gpb_acquisitions.py:161-176), one group at a time and stacked """
  from dragonfly_amd import kernel as K
  from dragonfly_amd.gp_core import GP
  rs = np.random.RandomState(11)
  n, d, m = 450, 9, 1300
  groups = [[0, 4, 7], [1, 2], [3, 5, 6, 8]]
  sc = [rs.random_sample(len(g)) + 0.4 for g in groups]
  bw = rs.random_sample(2) + 0.4
  subs_dev = [K.PolyKernel(3, 3, 0.7, sc[0]), K.SEKernel(2, 1.3, bw), K.PolyKernel(4, 2, 1.1, sc[2])]
  subs_ora = [O.KernelSpec('poly', 3, 0.7, sc[0], nu=3), O.KernelSpec('se', 2, 1.3, bw), O.KernelSpec('poly', 4, 1.1, sc[2], nu=2)]
  kern = K.AdditiveKernel(1.6, subs_dev, groups)
  spec = O.KernelSpec('additive', d, 1.6, groups=groups, subs=subs_ora)
  assert kern.has_device_spec()
  X, Xs = rs.random_sample((n, d)), rs.random_sample((m, d))
  assert relerr(kern(X), spec(X)) < 1e-13 and relerr(kern(Xs, X), spec(Xs, X)) < 1e-13
  Y = (X[:, groups[0]].sum(axis=1)) ** 2 + np.sin(4 * X[:, 1]) + 0.05 * rs.randn(n)
  mean_c, noise = float(np.median(Y)), float(Y.var() / 15)
  gp = GP(list(X), list(Y), kern, lambda x: np.array([mean_c] * len(x)), noise)
  og = O.GPOracle(X, Y, spec, mean_c, noise)
  mu, sd = gp.eval(Xs, 'std')
  mur, sdr = og.eval(Xs, 'std')
  assert relerr(gp.alpha, og.alpha) < 1e-10 and relerr(mu, mur) < 1e-10 and relerr(sd, sdr) < 1e-10
  t = 37
  cands = [rs.random_sample((700 + 50 * j, len(g))) for j, g in enumerate(groups)]
  want = [O.add_ucb_group_values(og, j, cands[j], t) for j in range(len(groups))]
  betas = [O.add_ucb_beta_th(len(g), t) for g in groups]
  bvs, bis, vals_all = gp.device_gp.add_ucb_all(betas, cands, return_vals=True)
  for j in range(len(groups)):
    bv, bi, vals = gp.device_gp.add_ucb_group(j, betas[j], cands[j], return_vals=True)
    assert relerr(vals, want[j]) < 1e-10 and relerr(vals_all[j], want[j]) < 1e-10
    assert bi == int(np.argmax(want[j])) == bis[j] and bv == vals[bi] and bvs[j] == vals_all[j][bi]


def _nested_pair(rs, fidel_kind='expdecay'):
  """ scale * k_fidel(z) * [s_add * (k_1(x_g1) + k_2(x_g2) + k_3(x_g3))] * k_extra(x_g4): a product kernel with an
      ADDITIVE factor in the middle (euclidean_gp.py:696-707), as our classes and as the oracle describe it """
  from dragonfly_amd import kernel as K
  d = 2 + 7 + 2
  coords = [[0, 1], [2, 3, 4, 5, 6, 7, 8], [9, 10]]
  groupings = [[4, 0, 6], [1, 5], [3, 2]]                      # relative to the factor's own 7 columns
  pw, bw_f = 2 * rs.random_sample(2) + 0.2, rs.random_sample(2) + 0.5
  sc_p = rs.random_sample(3) + 0.4
  bw2, bw3, bw4 = rs.random_sample(2) + 0.4, rs.random_sample(2) + 0.4, rs.random_sample(2) + 0.6
  if fidel_kind == 'expdecay':
    fid, fid_o = K.ExpDecayKernel(2, 1.0, 0.15, pw), O.KernelSpec('expdecay', 2, 1.0, pw, nu=0.15)
  else:
    fid, fid_o = K.SEKernel(2, 1.0, bw_f), O.KernelSpec('se', 2, 1.0, bw_f)
  subs = [K.PolyKernel(3, 2, 0.6, sc_p), K.SEKernel(2, 1.3, bw2), K.MaternKernel(2, 2.5, 0.8, bw3)]
  subs_o = [O.KernelSpec('poly', 3, 0.6, sc_p, nu=2), O.KernelSpec('se', 2, 1.3, bw2), O.KernelSpec('matern', 2, 0.8, bw3, nu=2.5)]
  if fidel_kind != 'expdecay':          # an all-stationary variant (constant prior variance, one-launch tuning objective)
    subs[0], subs_o[0] = K.SEKernel(3, 0.6, sc_p), O.KernelSpec('se', 3, 0.6, sc_p)
  add, add_o = K.AdditiveKernel(1.7, subs, groupings), O.KernelSpec('additive', 7, 1.7, groups=groupings, subs=subs_o)
  extra, extra_o = K.MaternKernel(2, 1.5, 1.0, bw4), O.KernelSpec('matern', 2, 1.0, bw4, nu=1.5)
  kern = K.CoordinateProductKernel(d, 2.1, [fid, add, extra], coords)
  spec = O.KernelSpec('product', d, 2.1, groups=coords, subs=[fid_o, add_o, extra_o])
  return kern, spec, d


@pytest.mark.parametrize('fidel_kind', ['expdecay', 'se'])
def test_product_kernel_with_an_additive_factor(engine, fidel_kind):
  """ struct dfh_kernel_desc's group_factor / factor_is_sum / factor_scale: Gram and cross matrices at
      ragged sizes, the GP (alpha, lml, mean, std with its prior variance, acquisitions, a joint
      Thompson block) and the batched tuning objective (one-launch path for n <= 128 and the
      lock-step path) against the oracle """
  from dragonfly_amd.gp_core import GP
  rs = np.random.RandomState(3 if fidel_kind == 'se' else 4)
  kern, spec, d = _nested_pair(rs, fidel_kind)
  assert kern.has_device_spec()
  for n1, n2 in ((1, 1), (70, 131), (300, 257)):
    X1, X2 = rs.random_sample((n1, d)), rs.random_sample((n2, d))
    assert relerr(kern(X1, X2), spec(X1, X2)) < 1e-13 and relerr(kern(X1), spec(X1)) < 1e-13
  n, m = 600, 3000
  X = rs.random_sample((n, d))
  Y = np.sin(2 * X[:, 2:9].sum(axis=1)) * (1 - 0.4 / (1 + 3 * X[:, :2].sum(axis=1))) + 0.05 * rs.randn(n)
  mean_c, noise = float(np.median(Y)), float(Y.var() / 15)
  gp = GP(list(X), list(Y), kern, lambda x: np.array([mean_c] * len(x)), noise)
  og = O.GPOracle(X, Y, spec, mean_c, noise)
  assert not gp._generic
  assert relerr(gp.alpha, og.alpha) < 1e-10 and abs(gp.compute_log_marginal_likelihood() - og.lml()) <= 1e-10 * abs(og.lml())
  Xs = rs.random_sample((m, d))
  mu, sd = gp.eval(Xs, 'std')
  mur, sdr = og.eval(Xs, 'std')
  assert relerr(mu, mur) < 1e-10 and relerr(sd, sdr) < 1e-10
  bv, bi, vals = gp.device_gp.acq_argmax('ei', Xs, params=(float(Y.max()), 0.0), mean_const=mean_c, return_vals=True)
  want = O.acq_values('ei', mur, sdr, float(Y.max()))
  assert relerr(vals, want) < 1e-10 and bi == int(np.argmax(want))
  # the tuning objective of a few such kernels in one call: n = 100 (one launch, stationary kernels only) and n = 600
  from dragonfly_amd.engine import get_engine
  for nn in (100, 600):
    specs, want = [], []
    for s in (0.7, 1.0, 1.9):
      k2, o2, _ = _nested_pair(np.random.RandomState(int(10 * s)), fidel_kind)
      k2.hyperparams['scale'] = s; o2.scale = s
      specs.append(k2.to_spec())
      want.append(O.GPOracle(X[:nn], Y[:nn], o2, mean_c, noise).lml())
    got = get_engine().gp_lml_batch(specs, X[:nn], Y[:nn], [mean_c] * 3, [noise] * 3)
    assert relerr(got, want) < 1e-10, (nn, got, want)


def test_bad_additive_factor_descriptions_are_rejected(engine):
  from dragonfly_amd.engine import KernelSpec
  X = np.random.RandomState(0).random_sample((6, 4))
  base = dict(groups=[[0], [1, 2], [3]], sub_kinds=['se', 'se', 'se'], sub_scales=[1.0] * 3, sub_nus=[0.0] * 3,
              sub_bandwidths=[np.ones(1), np.ones(2), np.ones(1)])
  ok = KernelSpec('product', 4, 1.0, group_factors=[0, 1, 1], factor_sums=[False, True], factor_scales=[1.0, 2.0], **base)
  assert engine.kernel_matrix(ok, X).shape == (6, 6)
  for gf, fs in (([0, 0, 1], [False, True]),        # a plain factor with two groups
                 ([1, 0, 1], [True, True]),          # factor indices must not decrease
                 ([0, 2, 2], [False, True, True])):  # ... nor skip
    bad = KernelSpec('product', 4, 1.0, group_factors=gf, factor_sums=fs, factor_scales=[1.0] * len(fs), **base)
    with pytest.raises(ValueError):
      engine.kernel_matrix(bad, X)
  with pytest.raises(ValueError):                    # additive factors exist in product kernels only
    engine.kernel_matrix(KernelSpec('additive', 4, 1.0, group_factors=[0, 1, 1], factor_sums=[False, True],
                                    factor_scales=[1.0, 1.0], **base), X)
