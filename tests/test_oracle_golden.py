"""CPU: pins the NumPy oracle (oracle/ref_numpy.py) against
  (a) the reference's own known-answer vectors (gp/unittest_kernel.py:38-53,82-124 and
      utils/unittest_general_utils.py:27-35,65-71) and
  (b) tests/golden/*.npz, outputs of the real reference (oracle/make_golden.py)."""
import numpy as np
import pytest

from conftest import GP_CASES, load_golden, relerr
from golden_kernels import oracle_kernel
from oracle import ref_numpy as O

# gp/unittest_kernel.py:38-53
DATA_1 = np.array([[1, 2], [3, 4.5]])
DATA_2 = np.array([[1, 2], [3, 4]])
BWS = [0.1, 1]
SE_SCALE = 2
MATERN_SCALE = 2.1


def test_se_known_answers():
  """ gp/unittest_kernel.py:82-91 """
  true_11 = SE_SCALE * np.array([[1, np.exp(-406.25/2)], [np.exp(-406.25/2), 1]])
  true_22 = SE_SCALE * np.array([[1, np.exp(-404/2)], [np.exp(-404/2), 1]])
  true_12 = SE_SCALE * np.array([[1, np.exp(-404/2)], [np.exp(-406.25/2), np.exp(-0.25/2)]])
  assert np.linalg.norm(true_11 - O.se_kernel(DATA_1, DATA_1, SE_SCALE, np.array(BWS))) < 1e-10
  assert np.linalg.norm(true_22 - O.se_kernel(DATA_2, DATA_2, SE_SCALE, np.array(BWS))) < 1e-10
  assert np.linalg.norm(true_12 - O.se_kernel(DATA_1, DATA_2, SE_SCALE, np.array(BWS))) < 1e-10


def _matern_closed_form(nu, scale, dist):
  """ gp/unittest_kernel.py:94-105 """
  if nu == 0.5:
    ret = np.exp(-dist)
  elif nu == 1.5:
    ret = np.exp(-np.sqrt(3) * dist) * (1 + np.sqrt(3) * dist)
  else:
    ret = np.exp(-np.sqrt(5) * dist) * (1 + np.sqrt(5) * dist + (5/3.0) * dist**2)
  return scale * ret


@pytest.mark.parametrize('nu', [0.5, 1.5, 2.5])
def test_matern_known_answers(nu):
  """ gp/unittest_kernel.py:107-124 """
  d11 = np.array([[0, np.sqrt(406.25)], [np.sqrt(406.25), 0]])
  d22 = np.array([[0, np.sqrt(404)], [np.sqrt(404), 0]])
  d12 = np.array([[0, np.sqrt(404)], [np.sqrt(406.25), np.sqrt(0.25)]])
  bws = np.array(BWS)
  for (A, B, dist) in ((DATA_1, DATA_1, d11), (DATA_2, DATA_2, d22), (DATA_1, DATA_2, d12)):
    K = O.matern_kernel(A, B, nu, MATERN_SCALE, bws)
    assert np.linalg.norm(_matern_closed_form(nu, MATERN_SCALE, dist) - K) < 1e-10


def test_dist_squared_known_answer():
  """ utils/unittest_general_utils.py:27-35: exact equality """
  X1 = np.array([[1, 2, 3], [1, 2, 4], [2, 3, 4.5]])
  X2 = np.array([[1, 2, 4], [1, 2, 5], [2, 3, 5]])
  true = np.array([[1, 4, 6], [0, 1, 3], [2.25, 2.25, 0.25]])
  assert (true == O.dist_squared(X1, X2)).all()


def test_stable_cholesky_reconstruction():
  """ utils/unittest_general_utils.py:65-71 """
  rs = np.random.RandomState(0)
  M = rs.normal(size=(5, 5))
  M = M.dot(M.T)
  L = O.stable_cholesky(M)
  assert np.linalg.norm(L.dot(L.T) - M) < 1e-5
  # rank-deficient matrix: the ladder kicks in and reports its power
  A = rs.normal(size=(30, 4))
  L, p = O.stable_cholesky(A.dot(A.T), return_power=True)
  assert p is not None and -11 <= p <= 4
  assert O.stable_cholesky(np.zeros((0, 0))).size == 0


@pytest.mark.parametrize('case', GP_CASES)
def test_oracle_matches_reference_outputs(case):
  """ the oracle reproduces the real reference bit for bit (same NumPy/SciPy/BLAS build) or to
      rounding (1e-13) where BLAS threading may reorder sums """
  g = load_golden('gp_' + case)
  kern = oracle_kernel(g)
  gp = O.GPOracle(g['X'], g['Y'], kern, float(g['mean_c']), float(g['noise']))
  tol = 1e-12
  assert relerr(gp.K_trtr_wo_noise, g['K']) < tol
  assert relerr(gp.L, g['L']) < tol
  assert relerr(gp.alpha, g['alpha']) < 1e-11
  assert abs(gp.lml() - float(g['lml'])) <= 1e-11 * abs(float(g['lml']))
  mu, sd = gp.eval(g['Xs'], 'std')
  _, cov = gp.eval(g['Xs'], 'covar')
  assert relerr(mu, g['mu']) < tol and relerr(sd, g['sd']) < 1e-11 and relerr(cov, g['cov']) < 1e-11
  _, sd_h = gp.eval_with_hallucinated_observations(g['Xs'], g['Xh'], 'std')
  assert relerr(sd_h, g['sd_h']) < 1e-11
  best = float(g['Y'].max())
  assert relerr(O.acq_values('ucb', mu, sd, float(g['beta_th'])), g['val_ucb']) < tol
  assert relerr(O.acq_values('ei', mu, sd, best), g['val_ei']) < 1e-11
  assert relerr(O.acq_values('pi', mu, sd, best), g['val_pi']) < 1e-11
  assert relerr(O.acq_values('ttei', mu, sd, best, 0.3), g['val_ttei']) < 1e-11
  assert abs(O.ucb_beta_th(kern.dim if kern.kind != 'additive' else g['X'].shape[1], len(g['Y']))
             - float(g['beta_th'])) < 1e-14
  ts = gp.draw_samples_blocked(g['Xs'], g['ts_U'], len(g['Xs']))
  assert relerr(ts, g['ts_sample']) < 1e-9


def test_oracle_chunked_eval_is_chunk_invariant():
  g = load_golden('gp_se_d32_n130')
  gp = O.GPOracle(g['X'], g['Y'], oracle_kernel(g), float(g['mean_c']), float(g['noise']))
  mu1, sd1 = gp.eval(g['Xs'], 'std')
  mu2, sd2 = gp.eval_chunked(g['Xs'], chunk=16)
  assert relerr(mu2, mu1) < 1e-13 and relerr(sd2, sd1) < 1e-12


def test_oracle_add_ucb_group_values():
  g = load_golden('gp_additive_d10_n80')
  kern = oracle_kernel(g)
  gp = O.GPOracle(g['X'], g['Y'], kern, float(g['mean_c']), float(g['noise']))
  rs = np.random.RandomState(3)
  for j, grp in enumerate(kern.groups):
    Xj = rs.random_sample((17, len(grp)))
    vals = O.add_ucb_group_values(gp, j, Xj, time_step=80)
    assert vals.shape == (17,) and np.all(np.isfinite(vals))


def test_argmax_first_semantics():
  assert O.argmax_first(np.array([1.0, 3.0, 3.0, 2.0])) == (3.0, 1)
  v, i = O.argmax_first(np.array([1.0, np.nan, 5.0, np.nan]))
  assert i == 1 and v != v


def _mfgp_oracle(g):
  fd, dd = g['ZZ'].shape[1], g['XX'].shape[1]
  subs = [O.KernelSpec('se', fd, 1.0, g['fbw']), O.KernelSpec('matern', dd, 1.0, g['dbw'], nu=2.5)]
  groups = [list(range(fd)), list(range(fd, fd + dd))]
  spec = O.KernelSpec('product', fd + dd, float(g['scale']), groups=groups, subs=subs)
  ZX = np.concatenate([g['ZZ'], g['XX']], axis=1)
  return spec, ZX, O.GPOracle(ZX, g['YY'], spec, float(g['mean_c']), float(g['noise']))


def test_oracle_product_kernel_matches_reference_mfgp():
  """ EuclideanMFGP of the real reference (coordinate-product kernel, euclidean_gp.py:347-412) """
  g = load_golden('mfgp_f1_d3_n70')
  spec, ZX, og = _mfgp_oracle(g)
  assert relerr(spec(ZX), g['K']) < 1e-12
  assert relerr(og.L, g['L']) < 1e-10 and relerr(og.alpha, g['alpha']) < 1e-10
  assert abs(og.lml() - float(g['lml'])) <= 1e-10 * abs(float(g['lml']))
  ZXs = np.concatenate([g['Zs'], g['Xs']], axis=1)
  mu, sd = og.eval(ZXs, 'std')
  assert relerr(mu, g['mu']) < 1e-10 and relerr(sd, g['sd']) < 1e-9
  _, sdh = og.eval_with_hallucinated_observations(ZXs, np.concatenate([g['Zh'], g['Xh']], axis=1), 'std')
  assert relerr(sdh, g['sdh']) < 1e-9


def test_oracle_product_kernel_with_an_additive_factor_matches_the_reference():
  """ the joint kernel of the reference's MF fitter with an additive domain model (a
      CoordinateProductKernel whose second kernel is an AdditiveKernel, euclidean_gp.py:696-707) and
      the GP it fits: the oracle's nested description against the reference's Gram matrix / outputs """
  g = load_golden('mf_fitter_additive_f1_d6_n40')
  cts = g['cts_hps']
  # hyper-parameter layout (euclidean_gp.py:454-486): mean? no -- [log scale] [fidelity bandwidths] [domain bandwidths]
  # preceded by the GP's own [noise] entries; the fixture also stores what they decode to
  sizes, flat = g['group_sizes'], g['groupings_flat']
  groups, at = [], 0
  for s in sizes:
    groups.append([int(c) for c in flat[at:at + s]]); at += s
  n_mean_noise = len(cts) - 1 - 1 - 6
  log_bw = cts[n_mean_noise + 1:]
  fid_bw, dom_bw = np.exp(log_bw[:1]), np.exp(log_bw[1:])
  dom = O.KernelSpec('additive', 6, 1.0, groups=groups, subs=[O.KernelSpec('se', len(grp), 1.0, dom_bw[grp]) for grp in groups])
  spec = O.KernelSpec('product', 7, float(g['scale']), groups=[[0], [1, 2, 3, 4, 5, 6]],
                      subs=[O.KernelSpec('se', 1, 1.0, fid_bw), dom])
  joint = np.concatenate((g['ZZ'], g['XX']), axis=1)
  assert relerr(spec(joint, joint), g['K']) < 1e-13
  joint_s = np.concatenate((g['Zs'], g['Xs']), axis=1)
  assert relerr(spec(joint_s, joint), g['K_cross']) < 1e-13
