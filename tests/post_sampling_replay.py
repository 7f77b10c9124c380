"""Checks the stand-alone posterior sampling of hyper-parameters (dragonfly_amd/hp_sampling.py through
EuclideanGPFitter: hp_tune_criterion='post_sampling') and the additive rand_exp_sampling against
the REAL reference's fitter under the same seed (tests/golden/post_sampling_d4_n30.npz,
oracle/make_golden.py: gen_post_sampling_cases): the same samples -- every slice / Metropolis
decision the same --, the same groupings, and the global random stream left where the reference
leaves it.  Shared by the CPU test (stand-in engine) and the MI355X test."""
from argparse import Namespace

import numpy as np

from conftest import load_golden
from oracle.make_golden import ADD_REXP_OPTS, POST_SAMPLING_CASES


def _flat(groupings):
  return np.array(sum([list(g) + [-1] for g in groupings], []), dtype=float)


def check_post_sampling(name, tol=1e-10):
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  g = load_golden('post_sampling_d4_n30')
  opts, num = POST_SAMPLING_CASES[name]
  np.random.seed(2718)
  fitter = EuclideanGPFitter(list(g['X']), list(g['Y']), options=Namespace(**opts))
  ret = fitter.fit_gp(num, 'post_sampling')
  assert ret[0] == str(g[name + '_kind'])
  if num == 1:
    _, gp, (cts, dscr) = ret
    others = [None]
    lml = float(g[name + '_lml'])
    assert abs(gp.compute_log_marginal_likelihood() - lml) <= tol * abs(lml)
  else:
    _, cts, dscr, others = ret
  # the chain's decisions are the reference's: the samples agree to rounding of the densities compared
  assert np.allclose(np.array(cts, dtype=float), g[name + '_cts'], rtol=0, atol=1e-12)
  assert np.array_equal(np.array(dscr, dtype=float), g[name + '_dscr'])
  for t, o in enumerate(others):
    key = name + '_grouping_%d' % t
    if key in g:
      assert np.array_equal(_flat(o.add_gp_groupings), g[key])
  assert np.random.random() == float(g[name + '_rand_after'])
  return fitter


def check_additive_rand_exp_sampling():
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  g = load_golden('post_sampling_d4_n30')
  np.random.seed(31415)
  fitter = EuclideanGPFitter(list(g['X']), list(g['Y']), options=Namespace(**ADD_REXP_OPTS))
  kind, cts, dscr, others, probs = fitter.fit_gp()
  assert kind == str(g['add_rexp_kind'])
  assert np.array_equal(np.array(cts, dtype=float), g['add_rexp_cts'])
  assert np.array_equal(np.array(dscr, dtype=float), g['add_rexp_dscr'])
  assert np.array_equal(_flat(others[0].add_gp_groupings), g['add_rexp_first_grouping'])
  assert np.random.random() == float(g['add_rexp_rand_after'])
  # weights exp(lml) / sum: a relative 1e-10 on a marginal likelihood is |lml| * 1e-10 on its weight
  # (they span 300 decades here), so the bound is computed per sample from the lml values themselves
  lml = np.array([fitter._tuning_objective_batch([c], [d], o)[0] for c, d, o in zip(cts, dscr, others)])
  bound = 1e-10 * (np.abs(lml) + np.abs(lml).max())
  ref = g['add_rexp_probs']
  pos = ref > 1e-300
  assert np.all(np.abs(np.asarray(probs)[pos] - ref[pos]) <= bound[pos] * ref[pos])
  assert np.all(np.asarray(probs)[~pos] <= 1e-300)
  # the bandit's use of the sample (gp_core.py:748-781) and of the adaptive method weights
  np.random.seed(5)
  fitter.fit_gp_for_gp_bandit(num_samples=3)
  kind, method, gp = fitter.get_next_gp()
  assert kind == 'sample_hps_with_probs' and method == 'ml' and gp.kernel is not None
  return fitter
