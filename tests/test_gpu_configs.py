"""MI355X, BASELINE.json configs 2 and 5 at their full sizes: the fit is
checked against the oracle on the full training set, the candidate stage against the oracle on a
sub-sample of the candidates plus the chunk-invariance of the device arg-max over all of them."""
import numpy as np
import pytest

from conftest import relerr
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu
TOL = 1e-10


from bench_configs import hartmann6, park1      # noqa: E402,F401  (one definition of the synthetic inputs)


def test_config2_hartmann6_matern_ei_full_size(engine):
  from dragonfly_amd.engine import KernelSpec
  n, d, m = 4096, 6, 65536
  X = np.random.RandomState(102).random_sample((n, d))
  Y = hartmann6(X)
  scale, bw = float(Y.var()), 0.5 * np.ones(d)
  mean_c, noise = float(np.median(Y)), float(Y.var() / 20)
  spec = KernelSpec('matern', d, scale, bw, nu=2.5)
  ospec = O.KernelSpec('matern', d, scale, bw, nu=2.5)
  og = O.GPOracle(X, Y, ospec, mean_c, noise)
  gp = engine.gp_fit(spec, X, Y - mean_c, noise)
  assert gp.jitter_power is None
  assert relerr(gp.get_alpha(), og.alpha) < TOL
  assert abs(gp.lml - og.lml()) <= TOL * abs(og.lml())
  Xs = np.random.RandomState(202).random_sample((m, d))
  best = float(Y.max())
  bv, bi, vals = gp.acq_argmax('ei', Xs, params=(best, 0.0), mean_const=mean_c, return_vals=True)
  # oracle on the first 4096 candidates, on a stride through all of them, and around the winner
  for sel in (np.arange(4096), np.arange(0, m, 16), np.arange(max(0, bi - 2048), min(m, bi + 2048))):
    mur, sdr = og.eval_chunked(Xs[sel], chunk=2048)
    vr = O.acq_values('ei', mur, sdr, best, 0.0)
    assert relerr(vals[sel], vr) < TOL
    assert int(np.argmax(vals[sel])) == O.argmax_first(vr)[1]
  assert bi == int(np.argmax(vals)) and bv == vals[bi]
  # the device arg-max does not depend on how the candidates are chunked / sharded
  parts = [gp.acq_argmax('ei', Xs[lo:lo + 16384], params=(best, 0.0), mean_const=mean_c) for lo in range(0, m, 16384)]
  k = int(np.argmax([p[0] for p in parts]))
  assert parts[k][0] == bv and parts[k][1] + 16384 * k == bi


def test_config5_additive_d100_add_ucb_full_size(engine):
  from dragonfly_amd.engine import KernelSpec
  n, d, G = 4096, 100, 20
  X = np.random.RandomState(105).random_sample((n, d))
  Y = sum(park1(X[:, 4 * i:4 * i + 4]) for i in range(25))          # 'park1-100': 25 tiles of Park1
  perm = list(np.random.RandomState(405).permutation(d))
  groups = [perm[i:i + 5] for i in range(0, d, 5)]                   # euclidean_gp.py:733-735
  bws = [0.2 * np.sqrt(5) * np.ones(5) for _ in groups]
  scale = float(Y.var())
  noise = float(Y.var() / 20)
  mean_c = float(np.median(Y))
  spec = KernelSpec('additive', d, scale, groups=groups, sub_kinds=['se'] * G, sub_scales=[1.0] * G,
                    sub_nus=[0.0] * G, sub_bandwidths=bws)
  ospec = O.KernelSpec('additive', d, scale, groups=groups,
                       subs=[O.KernelSpec('se', 5, 1.0, b) for b in bws])
  og = O.GPOracle(X, Y, ospec, mean_c, noise)
  gp = engine.gp_fit(spec, X, Y - mean_c, noise)
  assert relerr(gp.get_alpha(), og.alpha) < TOL
  assert abs(gp.lml - og.lml()) <= TOL * abs(og.lml())
  m_j = 65536 // G
  # all 20 groups in one device call (one posterior solve) == the per-group calls
  cands = [np.random.RandomState(205 + j).random_sample((m_j, 5)) for j in range(G)]
  betas = [O.add_ucb_beta_th(5, n)] * G
  bvs, bis, vals_all = gp.add_ucb_all(betas, cands, return_vals=True)
  for j in (0, 7, 19):
    bv, bi, vals = gp.add_ucb_group(j, betas[j], cands[j], return_vals=True)
    assert bi == bis[j] and bv == bvs[j] and relerr(vals_all[j], vals) < 1e-12
  for j in (0, 7, 19):
    Xj = np.random.RandomState(205 + j).random_sample((m_j, 5))
    beta = O.add_ucb_beta_th(5, n)
    bv, bi, vals = gp.add_ucb_group(j, beta, Xj, return_vals=True)
    vr = O.add_ucb_group_values(og, j, Xj, n)
    assert relerr(vals, vr) < TOL and bi == int(np.argmax(vr))
