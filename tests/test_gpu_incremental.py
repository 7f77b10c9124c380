"""MI355X: incremental posterior update (dfh_gp_append) against the oracle
fitted on the extended data and against a full device refit -- block-boundary crossings, chains of
single-point updates, the ladder fall-backs, additive kernels, and GP.add_data_multiple."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _problem(n, d, seed):
  rs = np.random.RandomState(seed)
  X = rs.rand(n, d)
  Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
  return rs, X, Y


@pytest.mark.parametrize('kind,n0,q', [('se', 500, 30), ('se', 1000, 1), ('matern', 64, 700), ('se', 1500, 600),
                                       ('matern', 512, 512), ('se', 1, 5)])
def test_append_matches_oracle_and_full_refit(engine, kind, n0, q):
  from dragonfly_amd.engine import KernelSpec
  d = 4
  rs, X, Y = _problem(n0 + q, d, n0 * 7 + q)
  bw = 0.4 + rs.rand(d)
  if kind == 'se':
    spec, ospec = KernelSpec('se', d, 1.3, bw), O.KernelSpec('se', d, 1.3, bw)
  else:
    spec, ospec = KernelSpec('matern', d, 1.3, bw, nu=2.5), O.KernelSpec('matern', d, 1.3, bw, nu=2.5)
  mean_c, noise = float(np.median(Y)), float(Y.var() / 20)
  base = engine.gp_fit(spec, X[:n0], Y[:n0] - mean_c, noise)
  base_alpha = base.get_alpha()
  ext = base.append(X[n0:], Y - mean_c)
  full = engine.gp_fit(spec, X, Y - mean_c, noise)
  og = O.GPOracle(X, Y, ospec, mean_c, noise)
  assert ext.n == n0 + q and ext.jitter_power is None
  assert relerr(ext.get_alpha(), og.alpha) < TOL and relerr(ext.get_alpha(), full.get_alpha()) < TOL
  assert relerr(np.tril(ext.get_L()), og.L) < TOL
  assert np.array_equal(np.triu(ext.get_L(), 1), np.zeros((n0 + q, n0 + q)))
  assert abs(ext.lml - og.lml()) <= TOL * abs(og.lml())
  Xs = rs.rand(700, d)
  mu, sd = ext.predict(Xs)
  mur, sdr = og.eval_chunked(Xs, chunk=512)
  assert relerr(mu + mean_c, mur) < TOL and relerr(sd, sdr) < TOL
  best = float(Y.max())
  a = ext.acq_argmax('ei', Xs, params=(best, 0.0), mean_const=mean_c)
  b = full.acq_argmax('ei', Xs, params=(best, 0.0), mean_const=mean_c)
  assert a[1] == b[1] and abs(a[0] - b[0]) <= TOL * abs(b[0])
  # the handle that was extended is untouched
  assert base.n == n0 and np.array_equal(base.get_alpha(), base_alpha)


def test_chain_of_single_point_updates(engine):
  from dragonfly_amd.engine import KernelSpec
  d, n0, steps = 3, 505, 12
  rs, X, Y = _problem(n0 + steps, d, 31)
  bw = np.full(d, 0.6)
  spec, ospec = KernelSpec('se', d, 2.0, bw), O.KernelSpec('se', d, 2.0, bw)
  noise = 0.01
  gp = engine.gp_fit(spec, X[:n0], Y[:n0], noise)
  for i in range(steps):
    gp = gp.append(X[n0 + i], Y[:n0 + i + 1])
  og = O.GPOracle(X, Y, ospec, 0.0, noise)
  assert gp.n == n0 + steps
  assert relerr(gp.get_alpha(), og.alpha) < TOL and abs(gp.lml - og.lml()) <= TOL * abs(og.lml())


def test_append_falls_back_to_the_ladder_like_a_rebuild(engine):
  from dragonfly_amd.engine import KernelSpec
  d, n0, q = 2, 30, 40
  rs, X, Y = _problem(n0, d, 5)
  # well separated points, no noise: K is close to the identity -> plainly positive definite
  spec, ospec = KernelSpec('se', d, 1.0, np.full(d, 0.05)), O.KernelSpec('se', d, 1.0, np.full(d, 0.05))
  base = engine.gp_fit(spec, X, Y, 0.0)
  assert base.jitter_power is None
  # 40 exact duplicates: the Schur complement is a zero matrix up to rounding -> not PD, and the
  # rebuilt matrix needs the ladder exactly as the reference's build_posterior would
  dup = rs.randint(0, n0, size=q)
  Xd = np.vstack([X, X[dup]])
  Yd = np.concatenate([Y, Y[dup]])
  ext = base.append(X[dup], Yd)
  og = O.GPOracle(Xd, Yd, ospec, 0.0, 0.0)
  assert og.jitter_power is not None and ext.jitter_power == og.jitter_power
  full = engine.gp_fit(spec, Xd, Yd, 0.0)
  assert full.jitter_power == ext.jitter_power and abs(full.lml - ext.lml) <= 1e-12 * abs(full.lml)
  # cond ~1e11 after the ladder: bound = twice the oracle's own distance from the same solve in extended
  # precision on the same jittered Gram matrix
  from truth_bounds import gram_bounds
  Kd = ospec(Xd, Xd)
  jit = (10.0 ** og.jitter_power) * float(np.diag(Kd).max())
  tol_l = gram_bounds(Kd, jit, Yd, dict(alpha=og.alpha, lml=og.lml()))['lml']
  assert abs(ext.lml - og.lml()) <= tol_l * abs(og.lml()), (abs(ext.lml - og.lml()) / abs(og.lml()), tol_l)
  with pytest.raises(np.linalg.LinAlgError):
    base.append(X[dup], Yd, allow_jitter=False)
  # an existing fit that needed the ladder is rebuilt as well (plain attempt first, then ladder)
  xn = rs.rand(3, d)
  yn = np.concatenate([Yd, rs.randn(3)])
  ext2 = ext.append(xn, yn)
  og2 = O.GPOracle(np.vstack([Xd, xn]), yn, ospec, 0.0, 0.0)
  assert og2.jitter_power is not None and ext2.jitter_power == og2.jitter_power
  Kd2 = ospec(np.vstack([Xd, xn]), np.vstack([Xd, xn]))
  jit2 = (10.0 ** og2.jitter_power) * float(np.diag(Kd2).max())
  tol_l2 = gram_bounds(Kd2, jit2, yn, dict(alpha=og2.alpha, lml=og2.lml()))['lml']
  assert ext2.n == n0 + q + 3 and abs(ext2.lml - og2.lml()) <= tol_l2 * abs(og2.lml())


def test_additive_kernel_append(engine):
  from dragonfly_amd.engine import KernelSpec
  rs = np.random.RandomState(8)
  d, n0, q = 10, 600, 40
  perm = list(rs.permutation(d))
  groups = [perm[i:i + 4] for i in range(0, d, 4)]
  bws = [0.5 + rs.rand(len(g)) for g in groups]
  G = len(groups)
  spec = KernelSpec('additive', d, 1.7, groups=groups, sub_kinds=['se'] * G, sub_scales=[1.0] * G,
                    sub_nus=[0.0] * G, sub_bandwidths=bws)
  ospec = O.KernelSpec('additive', d, 1.7, groups=groups, subs=[O.KernelSpec('se', len(g), 1.0, b) for g, b in zip(groups, bws)])
  X = rs.rand(n0 + q, d)
  Y = (X ** 2).sum(axis=1) + 0.05 * rs.randn(n0 + q)
  noise = float(Y.var() / 20)
  ext = engine.gp_fit(spec, X[:n0], Y[:n0], noise).append(X[n0:], Y)
  og = O.GPOracle(X, Y, ospec, 0.0, noise)
  assert relerr(ext.get_alpha(), og.alpha) < TOL and abs(ext.lml - og.lml()) <= TOL * abs(og.lml())
  j = 1
  Xj = rs.rand(200, len(groups[j]))
  bv, bi, vals = ext.add_ucb_group(j, 1.5, Xj, return_vals=True)
  og_vals = O.add_ucb_group_values(og, j, Xj, n0 + q)
  assert bi == int(np.argmax(og_vals))


def test_gp_add_data_multiple_appends_and_matches_rebuild(engine):
  from dragonfly_amd.gp_core import GP
  from dragonfly_amd import kernel as K
  d, n0 = 3, 300
  rs, X, Y = _problem(n0 + 25, d, 17)
  mean_c = float(np.mean(Y))
  mean_func = lambda x: np.array([mean_c] * len(x))
  out = []
  for inc in (True, False):
    gp = GP(list(X[:n0]), list(Y[:n0]), K.SEKernel(d, 1.5, np.full(d, 0.7)), mean_func, 0.02, build_posterior=False)
    gp.incremental_updates = inc
    gp.build_posterior()
    first = gp.device_gp
    gp.add_data_multiple(list(X[n0:n0 + 20]), list(Y[n0:n0 + 20]))
    assert (gp.device_gp is not first) and first.n == n0
    gp.add_data_single(X[n0 + 20], Y[n0 + 20])
    assert gp.num_tr_data == n0 + 21 and gp.device_gp.n == n0 + 21
    # a changed kernel must not reuse the cached factor
    gp.kernel = K.SEKernel(d, 1.5, np.full(d, 0.9))
    gp.add_data_multiple(list(X[n0 + 21:]), list(Y[n0 + 21:]))
    mu, sd = gp.eval(X[:50] + 0.01, 'std')
    out.append((gp.alpha.copy(), gp.compute_log_marginal_likelihood(), mu, sd))
  assert relerr(out[0][0], out[1][0]) < TOL and abs(out[0][1] - out[1][1]) <= TOL * abs(out[1][1])
  assert relerr(out[0][2], out[1][2]) < TOL and relerr(out[0][3], out[1][3]) < TOL
  og = O.GPOracle(X, Y, O.KernelSpec('se', d, 1.5, np.full(d, 0.9)), mean_c, 0.02)
  assert relerr(out[0][0], og.alpha) < TOL
