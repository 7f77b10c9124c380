"""CPU: stand-alone 'post_sampling' hyper-parameter tuning and additive rand_exp_sampling over the
stand-in engine against the real reference's fitter (same seed -> same samples), the Metropolis
walk against the reference's arithmetic, and the set-up / bookkeeping of the tuning methods."""
from argparse import Namespace

import numpy as np
import pytest

from post_sampling_replay import POST_SAMPLING_CASES, check_additive_rand_exp_sampling, check_post_sampling


@pytest.mark.parametrize('name', sorted(POST_SAMPLING_CASES))
def test_post_sampling_draws_the_reference_samples(monkeypatch, name):
  from oracle_engine import patch_engine
  from dragonfly_amd import euclidean_gp    # noqa: F401
  patch_engine(monkeypatch)
  check_post_sampling(name)


def test_additive_rand_exp_sampling_draws_the_reference_samples(monkeypatch):
  from oracle_engine import patch_engine
  from dragonfly_amd import euclidean_gp    # noqa: F401
  patch_engine(monkeypatch)
  check_additive_rand_exp_sampling()


def test_metropolis_integer_walk_and_priors():
  from dragonfly_amd.hp_sampling import CategoricalPrior, UniformPrior, metropolis_integer_walk
  u = UniformPrior(-1.0, 3.0)
  assert u.get_mean() == 1.0 and u.logp(3.5) == -np.inf and u.logp(0.0) == -np.log(4.0)
  c = CategoricalPrior([0.5, 1.5, 2.5], np.repeat(1.0 / 3, 3))
  assert c.get_id(1.5) == 1 and c.get_id(None) == -1 and c.get_id(float('nan')) == -1
  assert c.get_category(3) is None and c.logp(-1) == -np.inf and c.logp(2) == np.log(1.0 / 3)
  # a walk on {0..4} with p ~ (1, 2, 3, 2, 1): stays inside the support, visits every state
  w = np.log(np.array([1.0, 2.0, 3.0, 2.0, 1.0]))
  logp = lambda q: w[int(q[0])] if 0 <= int(q[0]) < 5 else -np.inf
  np.random.seed(3)
  chain = metropolis_integer_walk(lambda q, q0: (logp(q), logp(q0)), 2, 400)
  assert chain.shape == (400, 1) and set(np.unique(chain)) == {0.0, 1.0, 2.0, 3.0, 4.0}
  assert abs(np.mean(chain) - 2.0) < 0.4


def test_tuning_method_bookkeeping(monkeypatch):
  from oracle_engine import patch_engine
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  patch_engine(monkeypatch)
  rs = np.random.RandomState(2)
  X, Y = list(rs.random_sample((14, 2))), list(rs.randn(14))
  with pytest.raises(NotImplementedError):
    EuclideanGPFitter(X, Y, options=Namespace(hp_tune_criterion='post_mean'))
  with pytest.raises(ValueError):
    EuclideanGPFitter(X, Y, options=Namespace(hp_tune_criterion='cv'))
  f = EuclideanGPFitter(X, Y, options=Namespace(kernel_type='se', hp_tune_criterion='ml-post_sampling', hp_tune_probs='adaptive',
                                                ml_hp_tune_opt='rand', hp_tune_max_evals=20, post_hp_tune_burn=2))
  assert f.methods_to_use == ['ml', 'post_sampling'] and np.allclose(f.hp_tune_probs, [0.5, 0.5])
  f.update_hp_tune_method_weight('post_sampling', 3)
  p = f._get_adaptive_hp_tune_probs()          # pylint: disable=protected-access
  assert np.isclose(p.sum(), 1.0) and p[1] > p[0]
  np.random.seed(9)
  f.fit_gp_for_gp_bandit(num_samples=1)
  assert f.hp_tune_results['ml'][0] == 'fitted_gp' and f.hp_tune_results['post_sampling'][0] == 'post_fitted_gp'
  kind, method, gp = f.get_next_gp()
  assert method in ('ml', 'post_sampling') and kind in ('fitted_gp', 'post_fitted_gp') and gp is not None
  with pytest.raises(NotImplementedError):
    g = EuclideanGPFitter(X, Y, options=Namespace(hp_tune_criterion='post_sampling', post_hp_tune_method='nuts'))
    g.fit_gp(1, 'post_sampling')
