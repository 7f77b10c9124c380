"""MI355X: the mirrors composed into a small Bayesian-optimisation loop, stand-alone (no reference
needed): EuclideanGPFitter (batched random-search tuning) -> GP -> fused acquisition -> new
observation -> incremental posterior update.  Checks the plumbing end to end and that the loop does
what BO should do on Branin."""
from argparse import Namespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def branin(x):
  """ Branin on [-5, 10] x [0, 15], negated: maximum -0.397887 (euclidean_synthetic_functions.py:108) """
  a, b, c, r, s, t = 1.0, 5.1 / (4 * np.pi ** 2), 5 / np.pi, 6.0, 10.0, 1 / (8 * np.pi)
  return -(a * (x[1] - b * x[0] ** 2 + c * x[0] - r) ** 2 + s * (1 - t) * np.cos(x[0]) + s)


@pytest.mark.parametrize('acq', ['ucb', 'ei', 'ts'])
def test_bo_loop_on_branin(engine, acq):
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  from dragonfly_amd import gpb_acquisitions as A
  from dragonfly_amd.oper_utils import EuclideanDomain
  np.random.seed({'ucb': 11, 'ei': 12, 'ts': 13}[acq])
  lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
  to_raw = lambda u: lo + u * (hi - lo)
  domain = EuclideanDomain([[0, 1], [0, 1]])                 # Dragonfly normalises domains
  X = [np.random.random(2) for _ in range(8)]
  Y = [branin(to_raw(x)) for x in X]
  init_best = max(Y)
  gp = None
  for it in range(24):
    if it % 6 == 0:                                          # re-tune the hyper-parameters
      opts = Namespace(kernel_type='se', ml_hp_tune_opt='rand', hp_tune_max_evals=200, hp_tune_criterion='ml')
      _, gp, _ = EuclideanGPFitter(X, Y, options=opts).fit_gp()
    anc = Namespace(max_evals=4000, t=len(Y), domain=domain, acq_opt_method='rand', curr_max_val=max(Y),
                    handle_parallel='halluc', eval_points_in_progress=[], is_mf=False)
    x_next = np.asarray(getattr(A.asy, acq)(gp, anc), dtype=float)
    assert x_next.shape == (2,) and np.all(x_next >= 0) and np.all(x_next <= 1)
    y_next = branin(to_raw(x_next))
    n_before = gp.num_tr_data
    gp.add_data_single(x_next, y_next)                       # block-row append of the cached factor
    assert gp.num_tr_data == n_before + 1 and gp.device_gp.n == n_before + 1
    X, Y = list(gp.X), list(gp.Y)
  assert len(Y) == 32
  assert max(Y) > init_best or init_best > -1.0
  assert max(Y) > -2.5, max(Y)                               # global maximum is -0.398
