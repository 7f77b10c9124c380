"""MI355X: HIP path vs the NumPy oracle on identical seeded inputs, at sizes the oracle finishes
in seconds, plus the edge cases the reference handles (empty / ragged / single-point inputs,
non-PD matrices, the jitter ladder, NaN variance and np.argmax's NaN rule)."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ref_numpy as O

from truth_bounds import draw_bound, kernel_draw_bound    # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _spec_pair(kind, d, rs, scale=1.7, nu=2.5):
  from dragonfly_amd.engine import KernelSpec
  bw = 0.3 + rs.rand(d)
  if kind == 'se':
    return KernelSpec('se', d, scale, bw), O.KernelSpec('se', d, scale, bw)
  return KernelSpec('matern', d, scale, bw, nu=nu), O.KernelSpec('matern', d, scale, bw, nu=nu)


@pytest.mark.parametrize('kind,nu', [('se', None), ('matern', 0.5), ('matern', 1.5), ('matern', 2.5)])
@pytest.mark.parametrize('shape', [(1, 1, 1), (3, 5, 2), (128, 128, 6), (257, 130, 32), (70, 515, 45)])
def test_kernel_matrix(engine, kind, nu, shape):
  n1, n2, d = shape
  rs = np.random.RandomState(n1 * 7 + n2)
  spec, ospec = _spec_pair(kind, d, rs, nu=nu or 2.5)
  X1, X2 = rs.rand(n1, d), rs.rand(n2, d)
  assert relerr(engine.kernel_matrix(spec, X1, X2), ospec(X1, X2)) < 1e-12
  Ks = engine.kernel_matrix(spec, X1, None, diag_add=0.3)
  assert np.array_equal(Ks, Ks.T)                                       # bitwise symmetric
  tol = 1e-12
  if nu == 0.5:
    # the reference's diagonal carries sqrt(rounding of dist_sq) (kernel.py:296 on general_utils.py:66-68):
    # bound = twice the oracle's own distance from the extended-precision kernel matrix
    from oracle import ref_longdouble as T
    from truth_bounds import bound
    tol = bound(ospec(X1, X1), T.kernel_matrix('matern', ospec.bandwidths, ospec.scale, X1, X1, nu=0.5))
  assert relerr(Ks, ospec(X1, X1) + 0.3 * np.eye(n1)) <= tol


def test_reference_known_answers_on_device(engine):
  """ gp/unittest_kernel.py:82-124 and utils/unittest_general_utils.py:27-35 run on the HIP path """
  from dragonfly_amd import kernel as K
  from dragonfly_amd.general_utils import dist_squared
  d1 = np.array([[1, 2], [3, 4.5]])
  d2 = np.array([[1, 2], [3, 4]])
  kern = K.SEKernel(2, 2, [0.1, 1])
  true_12 = 2 * np.array([[1, np.exp(-404/2)], [np.exp(-406.25/2), np.exp(-0.25/2)]])
  true_11 = 2 * np.array([[1, np.exp(-406.25/2)], [np.exp(-406.25/2), 1]])
  assert np.linalg.norm(true_12 - kern(d1, d2)) < 1e-10
  assert np.linalg.norm(true_11 - kern(d1)) < 1e-10
  dist12 = np.array([[0, np.sqrt(404)], [np.sqrt(406.25), np.sqrt(0.25)]])
  for nu in (0.5, 1.5, 2.5):
    km = K.MaternKernel(2, nu, 2.1, [0.1, 1])
    if nu == 0.5:
      ref = np.exp(-dist12)
    elif nu == 1.5:
      ref = np.exp(-np.sqrt(3) * dist12) * (1 + np.sqrt(3) * dist12)
    else:
      ref = np.exp(-np.sqrt(5) * dist12) * (1 + np.sqrt(5) * dist12 + (5/3.0) * dist12**2)
    assert np.linalg.norm(2.1 * ref - km(d1, d2)) < 1e-10
  X1 = np.array([[1, 2, 3], [1, 2, 4], [2, 3, 4.5]])
  X2 = np.array([[1, 2, 4], [1, 2, 5], [2, 3, 5]])
  assert (np.array([[1, 4, 6], [0, 1, 3], [2.25, 2.25, 0.25]]) == dist_squared(X1, X2)).all()
  with pytest.raises(ValueError):
    dist_squared(X1, X2[:, :2])


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 511, 512, 513, 1000, 2048])
def test_cholesky_and_solves(engine, n):
  from dragonfly_amd import general_utils as G
  from scipy.linalg import solve_triangular
  rs = np.random.RandomState(n)
  X = rs.rand(n, 4)
  M = O.se_kernel(X, X, 1.0, np.full(4, 0.4)) + 0.05 * np.eye(n)
  L = G.stable_cholesky(M)
  Lr = np.linalg.cholesky(M)
  assert relerr(L, Lr) < 1e-11 and np.array_equal(np.triu(L, 1), np.zeros_like(L))
  b = rs.randn(n)
  assert relerr(G.solve_lower_triangular(Lr, b), solve_triangular(Lr, b, lower=True)) < 1e-11
  assert relerr(G.solve_upper_triangular(Lr.T, b), solve_triangular(Lr.T, b, lower=False)) < 1e-11
  if n in (65, 1000):
    B = rs.randn(n, 19)
    assert relerr(G.solve_lower_triangular(Lr, B), solve_triangular(Lr, B, lower=True)) < 1e-11
    assert relerr(G.solve_upper_triangular(Lr.T, B), solve_triangular(Lr.T, B, lower=False)) < 1e-11


def test_cholesky_failure_modes(engine):
  from dragonfly_amd import general_utils as G
  rs = np.random.RandomState(0)
  # not positive definite: numpy raises LinAlgError (general_utils.py:178)
  X = rs.rand(300, 3)
  M = O.se_kernel(X, X, 1.0, np.full(3, 0.4)) + 0.05 * np.eye(300)
  M[150, 150] = -1.0
  with pytest.raises(np.linalg.LinAlgError):
    G.stable_cholesky(M, add_to_diag_till_psd=False)
  # rank deficient: ladder adds 10^p max(diag) (general_utils.py:183-203), same power as numpy
  A = rs.randn(200, 20)
  M = A.dot(A.T)
  L, p = engine.stable_cholesky(M, return_power=True)
  _, pr = O.stable_cholesky(M, return_power=True)
  assert p == pr and relerr(L.dot(L.T), M) < 1e-8      # (reconstruction of a rank-40 matrix + jitter 1e-11 max diag: not a parity assert)
  # hopeless: ValueError after p = 4 (general_utils.py:199-203)
  with pytest.raises(ValueError):
    G.stable_cholesky(-np.eye(70))
  with pytest.raises(ValueError):
    bad = np.eye(10); bad[3, 3] = np.nan
    G.stable_cholesky(bad)
  assert G.stable_cholesky(np.zeros((0, 0))).size == 0
  assert G.solve_lower_triangular(np.zeros((0, 0)), np.zeros((0,))).shape == (0,)


@pytest.mark.parametrize('kind,d,n,m', [('se', 32, 1500, 3000), ('matern', 6, 2048, 4096), ('se', 2, 200, 1000)])
def test_fit_and_posterior_against_oracle(engine, kind, d, n, m):
  rs = np.random.RandomState(d * 1000 + n)
  spec, ospec = _spec_pair(kind, d, rs, scale=1.0)
  X = rs.rand(n, d)
  Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
  mean_c, noise = float(np.median(Y)), float(Y.var() / 20)
  og = O.GPOracle(X, Y, ospec, mean_c, noise)
  gp = engine.gp_fit(spec, X, Y - mean_c, noise)
  assert gp.jitter_power is None
  assert relerr(gp.get_alpha(), og.alpha) < TOL
  assert abs(gp.lml - og.lml()) <= TOL * abs(og.lml())
  Xs = rs.rand(m, d)
  mu, sd = gp.predict(Xs)
  mur, sdr = og.eval_chunked(Xs, chunk=1024)
  assert relerr(mu + mean_c, mur) < TOL and relerr(sd, sdr) < TOL
  best = float(Y.max())
  for acq, params in (('ucb', (2.5, 0.0)), ('ei', (best, 0.0)), ('pi', (best, 0.0)), ('ttei', (best, 0.2))):
    bv, bi, vals = gp.acq_argmax(acq, Xs, params=params, mean_const=mean_c, return_vals=True)
    vr = O.acq_values(acq, mur, sdr, *params)
    assert relerr(vals, vr) < TOL
    assert bi == O.argmax_first(vr)[1]
  # blocked Thompson sampling, several blocks per call
  U = rs.randn(m)
  blk = 512
  bv, bi, samp, jps = gp.thompson(Xs[:2048], U[:2048], block=blk, mean_const=mean_c, return_samples=True)
  sr = og.draw_samples_blocked(Xs[:2048], U[:2048], blk)
  # the block covariance needs the jitter ladder (cond ~1e11 after it): rounding differences of
  # 1e-16 in Sigma are amplified by its Cholesky factor.  Per block: twice the oracle draw's own distance
  # from the truth built from the kernel in extended precision
  for b in range(2048 // blk):
    sl = slice(b * blk, (b + 1) * blk)
    _, cov_b = og.eval(Xs[sl], 'covar')
    tol_b = kernel_draw_bound(kind, ospec.nu, ospec.bandwidths, ospec.scale, X, Y - mean_c, noise, Xs[sl], mean_c, U[sl],
                              sr[sl], cov_b)
    assert relerr(samp[sl], sr[sl]) <= tol_b, (b, relerr(samp[sl], sr[sl]), tol_b)
  assert bi == int(np.argmax(sr))


def test_ragged_and_single_point_inputs(engine):
  rs = np.random.RandomState(5)
  spec, ospec = _spec_pair('se', 3, rs)
  X = rs.rand(7, 3)
  Y = rs.randn(7)
  og = O.GPOracle(X, Y, ospec, 0.0, 0.1)
  gp = engine.gp_fit(spec, X, Y, 0.1)
  for m in (1, 2, 129):
    Xs = rs.rand(m, 3)
    mu, sd = gp.predict(Xs)
    mur, sdr = og.eval(Xs, 'std')
    assert relerr(mu, mur) < TOL and relerr(sd, sdr) < TOL
    mu_only, none = gp.predict(Xs, want_std=False)
    assert none is None and np.array_equal(mu_only, mu)
  # single training point
  gp1 = engine.gp_fit(spec, X[:1], Y[:1], 0.1)
  og1 = O.GPOracle(X[:1], Y[:1], ospec, 0.0, 0.1)
  mu, sd = gp1.predict(X)
  mur, sdr = og1.eval(X, 'std')
  assert relerr(mu, mur) < TOL and relerr(sd, sdr) < TOL


def test_nan_variance_and_argmax_rule(engine):
  """ gp_core.py:187 does not clip: a negative variance gives NaN, and np.argmax (oper_utils.py:73)
      returns the first NaN.  Tiny noise + a candidate on a training point forces the case. """
  rs = np.random.RandomState(9)
  from dragonfly_amd.engine import KernelSpec
  d = 2
  spec = KernelSpec('se', d, 1.0, np.full(d, 2.0))
  X = rs.rand(60, d)
  Y = rs.randn(60)
  gp = engine.gp_fit(spec, X, Y, 1e-13)
  Xs = np.vstack((rs.rand(50, d), X, rs.rand(50, d)))
  bv, bi, vals = gp.acq_argmax('ucb', Xs, params=(2.0, 0.0), return_vals=True)
  nan_idx = np.flatnonzero(np.isnan(vals))
  if len(nan_idx):
    assert bi == nan_idx[0] and bv != bv
  else:
    assert bi == int(np.argmax(vals))
  # explicit tie: duplicated candidates -> the first index wins
  Xt = np.vstack((Xs[:10], Xs[:10]))
  gp2 = engine.gp_fit(spec, X, Y, 0.1)
  bv, bi, vals = gp2.acq_argmax('ucb', Xt, params=(2.0, 0.0), return_vals=True)
  assert bi == int(np.argmax(vals)) and bi < 10


def test_gemm_public_entry(engine):
  rs = np.random.RandomState(2)
  A, B, C = rs.randn(300, 70), rs.randn(129, 70), rs.randn(300, 129)
  assert relerr(engine.gemm(A, B, C, alpha=1.5, beta=0.5), 0.5 * C + 1.5 * A.dot(B.T)) < 1e-13
  Bn = rs.randn(70, 129)
  assert relerr(engine.gemm(A, Bn, alpha=-1.0, transb=True), -A.dot(Bn)) < 1e-13
  # transpose-detecting: asymmetric A = I check (cdna_hip_programming.md section 3)
  I = np.eye(64)
  Basym = np.arange(64 * 64, dtype=float).reshape(64, 64)
  assert np.array_equal(engine.gemm(I, Basym, transb=True), Basym)
  assert np.array_equal(engine.gemm(I, Basym), Basym.T)


def test_additive_gp_and_add_ucb_groups(engine):
  from dragonfly_amd.engine import KernelSpec
  rs = np.random.RandomState(21)
  d, n = 12, 600
  perm = list(rs.permutation(d))
  groups = [perm[i:i + 5] for i in range(0, d, 5)]
  bws = [0.4 + rs.rand(len(g)) for g in groups]
  spec = KernelSpec('additive', d, 2.2, groups=groups, sub_kinds=['se'] * len(groups),
                    sub_scales=[1.0] * len(groups), sub_nus=[0.0] * len(groups), sub_bandwidths=bws)
  subs = [O.KernelSpec('se', len(g), 1.0, b) for g, b in zip(groups, bws)]
  ospec = O.KernelSpec('additive', d, 2.2, groups=groups, subs=subs)
  X = rs.rand(n, d)
  Y = (X ** 2).sum(axis=1) + 0.05 * rs.randn(n)
  og = O.GPOracle(X, Y, ospec, 0.0, float(Y.var() / 20))
  gp = engine.gp_fit(spec, X, Y, float(Y.var() / 20))
  assert relerr(gp.get_alpha(), og.alpha) < TOL
  for j, grp in enumerate(groups):
    Xj = rs.rand(333, len(grp))
    beta = O.add_ucb_beta_th(len(grp), n)
    bv, bi, vals = gp.add_ucb_group(j, beta, Xj, return_vals=True)
    vr = O.add_ucb_group_values(og, j, Xj, n)
    assert relerr(vals, vr) < TOL and bi == int(np.argmax(vr))


@pytest.mark.parametrize('noise_frac', [1e-13, 1e-9])
def test_hallucination_when_the_augmented_matrix_needs_the_ladder(engine, noise_frac):
  """ gp_core.py:199-206 factors the WHOLE augmented matrix with stable_cholesky.  A tiny fixed
      noise variance with a pending point that duplicates a training point makes the augmented
      matrix numerically singular (and at 1e-13 the base fit already needs the ladder): the
      device re-factors the augmented matrix with the ladder as the reference does instead of
      giving up (round-1 advisor finding).  Values agree to what such a matrix allows. """
  from dragonfly_amd.engine import KernelSpec
  rs = np.random.RandomState(5)
  n, d, m = 60, 2, 40
  X = rs.rand(n, d)
  Y = np.sin(3 * X.sum(axis=1))
  scale, bw = float(Y.var()), np.full(d, 0.3)
  noise = noise_frac * scale
  og = O.GPOracle(X, Y, O.KernelSpec('se', d, scale, bw), 0.0, noise)
  gp = engine.gp_fit(KernelSpec('se', d, scale, bw), X, Y, noise)
  assert gp.jitter_power == og.jitter_power
  Xh = np.vstack([X[7], rs.rand(d)])            # one duplicate of a training point, one fresh point
  Xs = rs.rand(m, d)
  mu_o, sd_o = og.eval_with_hallucinated_observations(Xs, Xh)
  mu_d, sd_d = gp.predict(Xs, X_halluc=Xh)
  # The augmented matrix is numerically singular: the two sides' solves differ by cond x eps, and a literal 1e-10
  # cannot hold.  The bound is computed, as everywhere (tests/truth_bounds.py): both sides against the same linear
  # algebra in extended precision on the same Gram matrices (with whatever jitter the reference's ladder settled
  # on), the device held to max(1e-10, 8 x the reference's own distance from that truth).  (8, not the 2 of the other
  # cases: at noise 1e-13 x scale the base matrix has cond ~ 1e13 and is factored WITHOUT jitter on both sides; two
  # correct eliminations in different orders land a few cond x eps apart -- measured 1.8e-10 for the device against
  # 4.5e-11 for NumPy in the mean -- and a factor of 2 would be a statement about luck, not about correctness.)
  from oracle import ref_longdouble as T
  from truth_bounds import bound
  ks = O.KernelSpec('se', d, scale, bw)
  K = ks(X, X)
  jit_b = 0.0 if og.jitter_power is None else (10.0 ** og.jitter_power) * float(np.diag(K + noise * np.eye(n)).max())
  tr = T.gram_truth(K, noise + jit_b, Y, ks(Xs, X), np.diag(ks(Xs, Xs)).copy(), None, 0.0)
  Xa = np.vstack([X, Xh])
  Ka = ks(Xa, Xa)
  _, pa = O.stable_cholesky(Ka + noise * np.eye(n + 2), return_power=True)
  jit_a = 0.0 if pa is None else (10.0 ** pa) * float(np.diag(Ka + noise * np.eye(n + 2)).max())
  tra = T.gram_truth(Ka, noise + jit_a, np.zeros(n + 2), ks(Xs, Xa), np.diag(ks(Xs, Xs)).copy(), ks(Xs, Xs), 0.0)
  assert relerr(mu_d, tr['mu']) <= bound(mu_o, tr['mu'], factor=8.0)
  assert np.all(np.isfinite(sd_d) == np.isfinite(sd_o))
  ok = np.isfinite(sd_o) & np.isfinite(tra['sd'])
  assert relerr(sd_d[ok], tra['sd'][ok]) <= bound(sd_o[ok], tra['sd'][ok], factor=8.0)
  _, cov_o = og.eval_with_hallucinated_observations(Xs, Xh, 'covar')
  _, cov_d = gp.predict_covar(Xs, X_halluc=Xh)
  assert relerr(cov_d, tra['cov']) <= bound(cov_o, tra['cov'], factor=8.0)


def test_thompson_blocks_in_the_panel_strip_regime(engine):
  """ 64 blocks of 1100 candidates factored in lock step: 588 rows below each block's first 512-panel =
      10 row strips per block, 640 in the batch -> panel_strip_kernel (csrc/chol.hip), last strip
      ragged (588 = 9*64 + 12); against the oracle's blocked draw, block by block """
  rs = np.random.RandomState(77)
  n, d, blk, nblk = 300, 4, 1100, 64
  spec, ospec = _spec_pair('se', d, rs, scale=1.0)
  X = rs.rand(n, d)
  Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
  mean_c, noise = float(np.median(Y)), float(Y.var() / 20)
  og = O.GPOracle(X, Y, ospec, mean_c, noise)
  gp = engine.gp_fit(spec, X, Y - mean_c, noise)
  m = blk * nblk
  Xs, U = rs.rand(m, d), rs.randn(m)
  bv, bi, samp, jps = gp.thompson(Xs, U, block=blk, mean_const=mean_c, return_samples=True)
  assert samp.shape == (m,) and len(jps) == nblk
  for b in (0, 1, 31, 63):                       # a few blocks against the oracle (one 1100 x 1100 Cholesky each)
    sl = slice(b * blk, (b + 1) * blk)
    want = og.draw_samples_blocked(Xs[sl], U[sl], blk)
    mu_b, cov_b = og.eval(Xs[sl], 'covar')
    # 1100 x 1100 covariance behind a 300-point fit: rank-deficient, jittered -- the draw is sensitive to the
    # rounding of the covariance itself: bound = twice the oracle's distance from the truth built from the kernel
    tol_b = kernel_draw_bound('se', 0.0, ospec.bandwidths, ospec.scale, X, Y - mean_c, noise, Xs[sl], mean_c, U[sl], want, cov_b)
    assert relerr(samp[sl], want) <= tol_b, (b, relerr(samp[sl], want), tol_b)
    assert int(np.argmax(samp[sl])) == int(np.argmax(want))
  assert bi == int(np.argmax(samp)) and bv == samp[bi]
  # the same call with the strips switched off is checked in tools (DFH_CHOL_STRIPS=0): here, blocks
  # drawn one per call (a batch of one: pivot-step path) must agree with the lock-step batch
  for b in (5, 40):
    sl = slice(b * blk, (b + 1) * blk)
    _, _, one, _ = gp.thompson(Xs[sl], U[sl], block=blk, mean_const=mean_c, return_samples=True)
    mu_b, cov_b = og.eval(Xs[sl], 'covar')
    assert relerr(one, samp[sl]) <= kernel_draw_bound('se', 0.0, ospec.bandwidths, ospec.scale, X, Y - mean_c, noise, Xs[sl],
                                                      mean_c, U[sl], og.draw_samples_blocked(Xs[sl], U[sl], blk), cov_b)


@pytest.mark.parametrize('comb', ['additive', 'product'])
@pytest.mark.parametrize('n', [1, 63, 65, 300, 1100])
def test_symmetric_gram_of_grouped_kernels(engine, comb, n):
  """ kernel.py:484-494 / 578-589: the lower-triangle kernel for symmetric Gram matrices of grouped
      kernels -- groups of 1 ... 16 coordinates (packed widths 4, 8, 12, 16: one to four parts per
      LDS fill), SE and Matern groups mixed, ragged tile edges; and a group of 17 coordinates, which
      sends the whole matrix through the generic kernel instead """
  from dragonfly_amd.engine import KernelSpec
  rs = np.random.RandomState(n)
  for sizes in ([1, 3, 5, 7, 13, 2, 16, 4], [5] * 7, [3, 17, 2]):
    d = sum(sizes)
    perm = list(rs.permutation(d))
    groups, at = [], 0
    for s in sizes:
      groups.append(perm[at:at + s]); at += s
    kinds = ['se' if i % 3 else 'matern' for i in range(len(sizes))]
    nus = [[1.5, 2.5][i % 2] for i in range(len(sizes))]       # (nu = 0.5 on a symmetric matrix: test_kernel_matrix)
    scales = list(0.5 + rs.rand(len(sizes))) if comb == 'additive' else [1.0] * len(sizes)
    bws = [0.6 + rs.rand(s) for s in sizes]
    spec = KernelSpec(comb, d, 1.3, groups=groups, sub_kinds=kinds, sub_scales=scales, sub_nus=nus, sub_bandwidths=bws)
    subs = [O.KernelSpec(k, len(g), sc, b, nu=nu) for k, g, sc, b, nu in zip(kinds, groups, scales, bws, nus)]
    ospec = O.KernelSpec(comb, d, 1.3, groups=groups, subs=subs)
    X = rs.rand(n, d)
    K = engine.kernel_matrix(spec, X, None, diag_add=0.25)
    want = ospec(X, X) + 0.25 * np.eye(n)
    assert np.array_equal(K, K.T)
    assert relerr(K, want) < 1e-12, (comb, n, sizes, relerr(K, want))
    # the cross-matrix route (generic kernel) gives the same values off the diagonal
    Kc = engine.kernel_matrix(spec, X, X.copy())
    off = ~np.eye(n, dtype=bool)
    if n > 1:
      assert relerr(K[off], Kc[off]) < 1e-13


@pytest.mark.parametrize('kind,d,n', [('se', 6, 700), ('matern', 20, 1333), ('se', 32, 513), ('matern', 9, 64)])
def test_posterior_mean_from_the_cross_matrix_pass(engine, kind, d, n):
  """ gp_core.py:174 accumulated inside the cross-matrix kernel (blocks of 512 columns): against the
      oracle for ragged candidate counts, and bit for bit the same whatever rows the call holds """
  rs = np.random.RandomState(d + n)
  spec, ospec = _spec_pair(kind, d, rs, scale=1.2)
  X = rs.rand(n, d)
  Y = np.cos(2 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  noise = float(Y.var() / 10)
  og = O.GPOracle(X, Y, ospec, 0.0, noise)
  gp = engine.gp_fit(spec, X, Y, noise)
  Xs = rs.rand(2311, d)
  mu_all, sd_all = gp.predict(Xs)
  mur, sdr = og.eval_chunked(Xs, chunk=1024)
  assert relerr(mu_all, mur) < TOL and relerr(sd_all, sdr) < TOL
  for lo, hi in ((0, 1), (5, 36), (31, 64), (100, 1133), (2000, 2311)):
    mu, sd = gp.predict(Xs[lo:hi])
    assert np.array_equal(mu, mu_all[lo:hi]), (lo, hi)
    assert relerr(sd, sd_all[lo:hi]) < 1e-12          # (few-row solves take a different product kernel)
  bv, bi, vals = gp.acq_argmax('ucb', Xs, params=(1.5, 0.0), return_vals=True)
  assert relerr(vals, mur + 1.5 * sdr) < TOL and bi == int(np.argmax(mur + 1.5 * sdr))
