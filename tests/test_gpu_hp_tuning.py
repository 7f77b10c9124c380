"""MI355X: the hyper-parameter-tuning inner loop as one device call (SURVEY.md section 8f-1,
dfh_gp_lml_batch): batched log marginal likelihoods against the oracle and against one fit per
candidate, including candidates that need the stable_cholesky ladder, groups larger than one
lock-step batch, and the fitter's random-search tuners."""
from argparse import Namespace

import numpy as np
import pytest

from conftest import load_golden
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _candidates(rs, d, nb, y_var):
  from dragonfly_amd.engine import KernelSpec
  specs, ospecs, means, noises = [], [], [], []
  for c in range(nb):
    scale = float(np.exp(rs.uniform(np.log(0.1 * y_var), np.log(10 * y_var))))
    bw = np.exp(rs.uniform(np.log(0.05), np.log(5.0), size=d))
    if c % 3 == 2:
      nu = [0.5, 1.5, 2.5][(c // 3) % 3]
      specs.append(KernelSpec('matern', d, scale, bw, nu=nu))
      ospecs.append(O.KernelSpec('matern', d, scale, bw, nu=nu))
    else:
      specs.append(KernelSpec('se', d, scale, bw))
      ospecs.append(O.KernelSpec('se', d, scale, bw))
    means.append(float(rs.randn()))
    noises.append(float(np.exp(rs.uniform(np.log(0.005 * y_var), np.log(0.2 * y_var)))))
  return specs, ospecs, means, noises


# the last two are lock-step batches with more than 512 row strips below the first 512-block: the panel
# strip kernel (csrc/chol.hip), with a ragged last strip (988 = 15*64 + 28, 1588 = 24*64 + 52 rows)
@pytest.mark.parametrize('n,d,nb', [(45, 3, 7), (200, 2, 150), (700, 6, 9), (1500, 4, 5), (1500, 4, 64), (2100, 3, 40)])
def test_lml_batch_matches_oracle_and_single_fits(engine, n, d, nb):
  rs = np.random.RandomState(n + nb)
  X = rs.rand(n, d)
  Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  specs, ospecs, means, noises = _candidates(rs, d, nb, float(Y.var()))
  lml, powers = engine.gp_lml_batch(specs, X, Y, means, noises, return_powers=True)
  assert lml.shape == (nb,)
  check = range(nb) if nb <= 20 else sorted(set(c for c in list(range(0, nb, 11)) + [63, 64, 65, nb - 1] if c < nb))
  for c in check:
    ref = O.GPOracle(X, Y, ospecs[c], means[c], noises[c]).lml()
    assert abs(lml[c] - ref) <= TOL * abs(ref), (c, lml[c], ref)
    one = engine.gp_fit(specs[c], X, Y - means[c], noises[c])
    assert abs(lml[c] - one.lml) <= 1e-12 * abs(one.lml) and powers[c] == one.jitter_power
    one.free()


def test_lml_batch_jitter_ladder_per_candidate(engine):
  """ duplicated training points and a vanishing noise variance: the Gram matrix of some
      candidates is numerically singular -> those (and only those) go through the ladder """
  from dragonfly_amd.engine import KernelSpec
  rs = np.random.RandomState(77)
  n, d = 300, 2
  X = rs.rand(n, d)
  X[150:] = X[:150]
  Y = np.cos(3 * X[:, 0]) + X[:, 1]
  specs = [KernelSpec('se', d, 1.0, np.full(d, bw)) for bw in (0.3, 2.0, 0.5, 3.0, 0.2)]
  ospecs = [O.KernelSpec('se', d, 1.0, np.full(d, bw)) for bw in (0.3, 2.0, 0.5, 3.0, 0.2)]
  noises = [1e-3, 0.0, 1e-2, 1e-18, 1e-3]
  lml, powers = engine.gp_lml_batch(specs, X, Y, None, noises, return_powers=True)
  assert powers[0] is None and powers[2] is None and powers[4] is None
  assert powers[1] is not None and powers[3] is not None
  for c in range(5):
    og = O.GPOracle(X, Y, ospecs[c], 0.0, noises[c])
    assert og.jitter_power == powers[c]
    ref = og.lml()
    # after the ladder K + 1e-11 max(diag) I still has cond ~1e11-1e13: y^T alpha amplifies the 1e-16
    # differences between two correct factorisations.  Bound: twice the oracle's own distance from the
    # same solve in extended precision on the same (jittered) Gram matrix
    tol = TOL
    if powers[c] is not None:
      from truth_bounds import gram_bounds
      Kc = ospecs[c](X, X)
      jit = (10.0 ** powers[c]) * float(np.diag(Kc + noises[c] * np.eye(n)).max())
      tol = gram_bounds(Kc, noises[c] + jit, Y, dict(alpha=og.alpha, lml=ref))['lml']
    assert abs(lml[c] - ref) <= tol * abs(ref), (c, lml[c], ref, tol)
    one = engine.gp_fit(specs[c], X, Y, noises[c])
    # the batch evaluates y^T alpha as ||L^-1 y||^2 (forward solve only), the single fit through both
    # solves: identical to 1e-12 for well-conditioned candidates, to the conditioning's share of it after the ladder
    assert one.jitter_power == powers[c] and abs(one.lml - lml[c]) <= (1e-12 if powers[c] is None else tol) * abs(one.lml)
  with pytest.raises(np.linalg.LinAlgError):
    engine.gp_lml_batch(specs, X, Y, None, noises, allow_jitter=False)


@pytest.mark.parametrize('kt', ['se', 'matern'])
def test_fitter_batched_and_per_candidate_tuning_agree(engine, kt):
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  g = load_golden('fitter_d3_n45')
  out = []
  for batch in (True, False):
    opts = Namespace(kernel_type=kt, ml_hp_tune_opt='rand', hp_tune_max_evals=60, hp_tune_criterion='ml')
    np.random.seed(4242)
    fitter = EuclideanGPFitter(list(g['X']), list(g['Y']), options=opts)
    fitter.batch_lml = batch
    kind, gp, hps = fitter.fit_gp()
    out.append((np.array(hps[0], dtype=float), list(hps[1]), gp.compute_log_marginal_likelihood()))
  assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]
  assert np.array_equal(out[0][0], g[kt + '_cts_hps'])          # the reference's own choice
  assert abs(out[0][2] - out[1][2]) <= 1e-12 * abs(out[1][2])


def test_fitter_rand_exp_sampling_probabilities(engine):
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  g = load_golden('fitter_d3_n45')
  res = []
  for batch in (True, False):
    opts = Namespace(kernel_type='matern', matern_nu=-1.0, ml_hp_tune_opt='rand_exp_sampling',
                     hp_tune_max_evals=40, hp_tune_criterion='ml')
    np.random.seed(99)
    fitter = EuclideanGPFitter(list(g['X']), list(g['Y']), options=opts)
    fitter.batch_lml = batch
    ret = fitter.fit_gp()
    assert ret[0] == 'sample_hps_with_probs'
    res.append(ret)
  assert np.array_equal(np.asarray(res[0][1]), np.asarray(res[1][1])) and res[0][2] == res[1][2]
  assert np.allclose(res[0][4], res[1][4], rtol=1e-10, atol=1e-300)
  assert abs(res[0][4].sum() - 1.0) < 1e-12


def test_additive_model_tuning_batched_and_per_candidate_agree(engine):
  """ use_additive_gp: random groupings x random continuous hyper-parameters (euclidean_gp.py:718-746);
      the additive kernels go through the non-uniform batch path """
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  rs = np.random.RandomState(5)
  n, d = 60, 6
  X = rs.rand(n, d)
  Y = (X[:, :2] ** 2).sum(axis=1) + np.sin(3 * X[:, 2]) + 0.05 * rs.randn(n)
  out = []
  for batch in (True, False):
    opts = Namespace(kernel_type='se', ml_hp_tune_opt='rand', hp_tune_max_evals=600, hp_tune_criterion='ml',
                     use_additive_gp=True, add_max_group_size=2, num_groups_per_group_size=3)
    np.random.seed(77)
    fitter = EuclideanGPFitter(list(X), list(Y), options=opts)
    fitter.batch_lml = batch
    kind, gp, hps = fitter.fit_gp()
    assert kind == 'fitted_gp' and type(gp.kernel).__name__ == 'AdditiveKernel'
    out.append((np.array(hps[0], dtype=float), list(hps[1]), gp.compute_log_marginal_likelihood(),
                [list(g) for g in gp.kernel.groupings]))
  assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1] and out[0][3] == out[1][3]
  assert abs(out[0][2] - out[1][2]) <= 1e-12 * abs(out[1][2])


def _tiny_specs(rs, d, y_var):
  """ one candidate of every kernel structure the one-launch path handles """
  from dragonfly_amd.engine import KernelSpec
  bw = lambda k: np.exp(rs.uniform(np.log(0.1), np.log(3.0), size=k))
  pairs = []
  for kind, nu in (('se', None), ('matern', 0.5), ('matern', 1.5), ('matern', 2.5)):
    b = bw(d)
    pairs.append((KernelSpec(kind, d, y_var, b, nu=nu or 0.0), O.KernelSpec(kind, d, y_var, b, nu=nu)))
  if d >= 4:
    perm = list(rs.permutation(d))
    groups = [perm[:d // 2], perm[d // 2:]]
    for multi in ('additive', 'product'):
      bws = [bw(len(g)) for g in groups]
      kinds, nus = ['se', 'matern'], [0.0, 2.5]
      pairs.append((KernelSpec(multi, d, 0.7 * y_var, groups=groups, sub_kinds=kinds, sub_scales=[1.0, 1.0],
                               sub_nus=nus, sub_bandwidths=bws),
                    O.KernelSpec(multi, d, 0.7 * y_var, groups=groups,
                                 subs=[O.KernelSpec('se', len(groups[0]), 1.0, bws[0]),
                                       O.KernelSpec('matern', len(groups[1]), 1.0, bws[1], nu=2.5)])))
  return pairs


@pytest.mark.parametrize('n,d', [(1, 1), (2, 3), (17, 2), (64, 5), (100, 32), (128, 6), (128, 64)])
def test_small_problems_take_the_one_launch_path_and_match(engine, n, d):
  """ n <= 128: pack, Gram matrix, stable_cholesky and solve of every candidate are one kernel
      (k_lml_tiny); same numbers as the oracle and as one dfh_gp_fit per candidate """
  rs = np.random.RandomState(1000 * n + d)
  X = rs.rand(n, d)
  Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  y_var = float(Y.var()) if n > 1 else 1.0
  pairs = _tiny_specs(rs, d, y_var) * 3
  means = [float(rs.randn()) for _ in pairs]
  noises = [float(np.exp(rs.uniform(np.log(0.005 * y_var), np.log(0.2 * y_var)))) for _ in pairs]
  lml, powers = engine.gp_lml_batch([p[0] for p in pairs], X, Y, means, noises, return_powers=True)
  for c, (spec, ospec) in enumerate(pairs):
    ref = O.GPOracle(X, Y, ospec, means[c], noises[c]).lml()
    tol = TOL
    if ospec.kind == 'matern' and ospec.nu == 0.5 and n > 1:
      # sqrt of a squared distance that is rounding noise on the diagonal: twice the
      # oracle's own distance from the extended-precision value
      from oracle import ref_longdouble as T
      from truth_bounds import bound
      tr = T.gp_truth('matern', ospec.bandwidths, ospec.scale, X, Y - means[c], noises[c], nu=0.5)
      tol = max(bound([ref], [tr['lml']]), TOL)
    assert abs(lml[c] - ref) <= tol * max(abs(ref), 1.0), (c, ospec.kind, lml[c], ref, tol)
    one = engine.gp_fit(spec, X, Y - means[c], noises[c])
    assert abs(lml[c] - one.lml) <= 1e-11 * max(abs(one.lml), 1.0) and powers[c] == one.jitter_power
    one.free()


def test_small_problem_ladder_and_failures(engine):
  from dragonfly_amd.engine import KernelSpec
  rs = np.random.RandomState(5)
  n, d = 60, 2
  X = rs.rand(n, d)
  X[30:] = X[:30]                                   # exact duplicates: singular without noise
  Y = np.cos(3 * X[:, 0]) + X[:, 1]
  bws = (0.3, 2.0, 0.5)
  specs = [KernelSpec('se', d, 1.0, np.full(d, b)) for b in bws]
  noises = [1e-3, 0.0, 1e-2]
  lml, powers = engine.gp_lml_batch(specs, X, Y, None, noises, return_powers=True)
  for c, b in enumerate(bws):
    og = O.GPOracle(X, Y, O.KernelSpec('se', d, 1.0, np.full(d, b)), 0.0, noises[c])
    assert og.jitter_power == powers[c]
    tol = TOL if powers[c] is None else 1e-4
    assert abs(lml[c] - og.lml()) <= tol * abs(og.lml())
  assert powers[0] is None and powers[1] is not None
  with pytest.raises(np.linalg.LinAlgError):
    engine.gp_lml_batch(specs, X, Y, None, noises, allow_jitter=False)
  Ynan = Y.copy()
  Xnan = X.copy()
  Xnan[3, 1] = np.nan                               # NaN in the Gram matrix: the ladder cannot help
  with pytest.raises(ValueError):
    engine.gp_lml_batch(specs[:1], Xnan, Ynan, None, noises[:1])
