"""MI355X: the posterior row solve as one product per diagonal block (csrc/chol.hip: trsm_rows_shifted; gp_core.py:180
solve_lower_triangular(L, K_tetr.T)) against the oracle -- on the three routes that take it (GP.eval / the fused
acquisitions, the Thompson pipeline, the one-call add-UCB), at orders with a partial last block and row counts with a
ragged last tile, and against the two-launch form of the same library (DFH_TRSM_FUSED=0 in a child process)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import relerr
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu
TOL = 1e-10
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem(n, d=6, seed=0):
  rs = np.random.RandomState(1000 + n + seed)
  X = rs.random_sample((n, d))
  Y = np.sin(3.0 * X).sum(axis=1) + 0.1 * rs.randn(n)
  return X, Y, float(Y.var()), 0.4 * np.ones(d), float(np.median(Y)), float(Y.var() / 50)


@pytest.mark.parametrize('n,m', [(1024, 4096), (1152, 2000), (2048, 1024 + 77), (1536, 1024)])
def test_eval_and_acquisition_take_the_one_product_solve(engine, n, m):
  from dragonfly_amd.engine import KernelSpec
  X, Y, scale, bw, mean_c, noise = _problem(n)
  og = O.GPOracle(X, Y, O.KernelSpec('matern', 6, scale, bw, nu=2.5), mean_c, noise)
  gp = engine.gp_fit(KernelSpec('matern', 6, scale, bw, nu=2.5), X, Y - mean_c, noise)
  Xs = np.random.RandomState(7 + n).random_sample((m, 6))
  c0 = engine.counters()['trsm_one_product']
  mu, sd = gp.predict(Xs)
  assert engine.counters()['trsm_one_product'] == c0 + 1
  mur, sdr = og.eval_chunked(Xs, chunk=2048)
  assert relerr(mu + mean_c, mur) < TOL and relerr(sd, sdr) < TOL
  best = float(Y.max())
  bv, bi, vals = gp.acq_argmax('ei', Xs, params=(best, 0.0), mean_const=mean_c, return_vals=True)
  vr = O.acq_values('ei', mur, sdr, best, 0.0)
  assert relerr(vals, vr) < TOL and bi == O.argmax_first(vr)[1]
  # a row's value does not depend on where in the call it stands (full tiles and the ragged last one: same sums)
  lo = (m // 128) * 128 - 64
  _, sd_tail = gp.predict(np.vstack([Xs[lo:], Xs[:1024]]))
  assert np.array_equal(sd_tail[:m - lo], sd[lo:])
  # fewer than 1024 rows keep the two-launch form
  c1 = engine.counters()['trsm_one_product']
  _, sd_few = gp.predict(Xs[:512])
  assert engine.counters()['trsm_one_product'] == c1 and relerr(sd_few, sdr[:512]) < TOL
  # hallucinated observations ride on the same solve
  Xh = np.random.RandomState(3).random_sample((3, 6))
  _, sdh = gp.predict(Xs, X_halluc=Xh)
  _, sdhr = og.eval_with_hallucinated_observations(Xs[:1024], Xh, 'std')
  assert relerr(sdh[:1024], sdhr) < TOL
  gp.free()


def test_thompson_pipeline_on_the_wider_rows(engine):
  from dragonfly_amd.engine import KernelSpec
  n, m, block = 1152, 2304, 256
  X, Y, scale, bw, mean_c, noise = _problem(n, seed=1)
  og = O.GPOracle(X, Y, O.KernelSpec('se', 6, scale, bw), mean_c, noise)
  gp = engine.gp_fit(KernelSpec('se', 6, scale, bw), X, Y - mean_c, noise)
  Xs = np.random.RandomState(11).random_sample((m, 6))
  U = np.random.RandomState(12).randn(m)
  c0 = engine.counters()['trsm_one_product']
  bv, bi, samp, powers = gp.thompson(Xs, U, block=block, mean_const=mean_c, return_samples=True)
  assert engine.counters()['trsm_one_product'] > c0
  want = og.draw_samples_blocked(Xs, U, block)
  # (a draw is the factor of a difference of nearly equal matrices; its contract-level checks with computed bounds are
  #  tests/test_gpu_headline.py and the engine traces -- here the point is the row stride of V^T through the pipeline)
  assert relerr(samp, want) < 1e-7 and bi == int(np.argmax(samp)) and bv == samp[bi]
  assert powers == [None] * len(powers)
  gp.free()


def test_add_ucb_one_call_on_the_wider_rows(engine):
  from dragonfly_amd.engine import KernelSpec
  n, d, G = 1024, 12, 4
  rs = np.random.RandomState(5)
  X = rs.random_sample((n, d))
  Y = np.cos(2.0 * X).sum(axis=1)
  groups = [list(range(3 * i, 3 * i + 3)) for i in range(G)]
  bws = [0.5 * np.ones(3) for _ in groups]
  scale, noise, mean_c = float(Y.var()), float(Y.var() / 20), float(np.median(Y))
  spec = KernelSpec('additive', d, scale, groups=groups, sub_kinds=['se'] * G, sub_scales=[1.0] * G, sub_nus=[0.0] * G,
                    sub_bandwidths=bws)
  og = O.GPOracle(X, Y, O.KernelSpec('additive', d, scale, groups=groups, subs=[O.KernelSpec('se', 3, 1.0, b) for b in bws]),
                  mean_c, noise)
  gp = engine.gp_fit(spec, X, Y - mean_c, noise)
  m_j = 700                                              # 4 x 700 = 2800 rows: 21 full tiles + 112
  cands = [np.random.RandomState(30 + j).random_sample((m_j, 3)) for j in range(G)]
  betas = [O.add_ucb_beta_th(3, n)] * G
  c0 = engine.counters()['trsm_one_product']
  bvs, bis, vals_all = gp.add_ucb_all(betas, cands, return_vals=True)
  assert engine.counters()['trsm_one_product'] == c0 + 1
  for j in range(G):
    vr = O.add_ucb_group_values(og, j, cands[j], n)
    assert relerr(vals_all[j], vr) < TOL and bis[j] == int(np.argmax(vr))
  gp.free()


_CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
out = {}
for n, m in ((2048, 4096), (4096, 8192)):
  rs = np.random.RandomState(1000 + n)
  X = rs.random_sample((n, 6)); Y = np.sin(3.0 * X).sum(axis=1) + 0.1 * rs.randn(n)
  gp = eng.gp_fit(KernelSpec('matern', 6, float(Y.var()), 0.4 * np.ones(6), nu=2.5), X, Y - float(np.median(Y)), float(Y.var() / 50))
  Xs = np.random.RandomState(7 + n).random_sample((m, 6))
  mu, sd = gp.predict(Xs)
  out[str(n)] = {'sd': sd.tolist(), 'mu': mu.tolist(), 'count': eng.counters()['trsm_one_product']}
print('RESULT ' + json.dumps(out))
'''


def test_against_the_two_launch_form_of_the_same_library():
  res = {}
  for fused in ('1', '0'):
    env = dict(os.environ, DFH_TRSM_FUSED=fused)
    p = subprocess.run([sys.executable, '-c', _CHILD % {'root': ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res[fused] = json.loads([l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1][7:])
  for n in ('2048', '4096'):
    assert res['1'][n]['count'] > 0 and res['0'][n]['count'] == 0
    assert res['1'][n]['mu'] == res['0'][n]['mu']                       # the mean never sees the solve
    assert relerr(np.array(res['1'][n]['sd']), np.array(res['0'][n]['sd'])) < 1e-12
