"""Replays tests/golden/trajectory_branin.npz -- every seam-crossing call of a multi-step run of the
REAL reference's EuclideanGPBandit, recorded in order with the global RNG state before and after
(oracle/make_golden.py: gen_trajectory_case) -- through the mirrors of dragonfly_amd.  Shared by
the CPU plumbing test (stand-in engine) and the MI355X test (libdfhip.so)."""
import json
from argparse import Namespace

import numpy as np

from conftest import load_golden


def _set_state(s):
  np.random.set_state(('MT19937', np.array(s[0], dtype=np.uint32), int(s[1]), int(s[2]), float(s[3])))


def _state_equal(s):
  cur = np.random.get_state()
  return np.array_equal(cur[1], np.array(s[0], dtype=np.uint32)) and int(cur[2]) == int(s[1]) and \
      int(cur[3]) == int(s[2]) and float(cur[4]) == float(s[3])


def _same_gp(gp, want):
  k = gp.kernel
  assert type(k).__name__ == want['kernel'], (type(k).__name__, want['kernel'])
  assert float(k.hyperparams['scale']) == want['scale']
  assert [float(b) for b in np.ravel(k.hyperparams['dim_bandwidths'])] == want['bw']
  assert float(gp.noise_var) == want['noise']
  assert float(gp.mean_func([np.zeros(k.dim)])[0]) == want['mean']
  assert int(gp.num_tr_data) == want['n']


def replay():
  """ Returns the number of (fits, acquisitions) replayed; raises AssertionError on the first
      call whose result or random-number consumption differs from the reference's. """
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  from dragonfly_amd import gpb_acquisitions as A
  from dragonfly_amd.oper_utils import EuclideanDomain
  g = load_golden('trajectory_branin')
  events = json.loads(bytes(g['events_json']).decode('utf-8'))
  fitter, gp = None, None
  n_fit = n_acq = 0
  for i, ev in enumerate(events):
    kind = ev['type']
    if kind == 'fit':
      _set_state(ev['before'])
      fitter = EuclideanGPFitter([np.array(x) for x in ev['X']], list(ev['Y']), options=Namespace(**ev['options']))
      fitter.fit_gp_for_gp_bandit(ev['num_samples'])
      assert _state_equal(ev['after']), 'event %d: the fit consumed different random numbers' % i
      n_fit += 1
    elif kind == 'next_gp':
      _set_state(ev['before'])
      fit_type, method, gp = fitter.get_next_gp()
      assert (fit_type, method) == (ev['fit_type'], ev['method'])
      assert _state_equal(ev['after']), 'event %d: get_next_gp consumed different random numbers' % i
      _same_gp(gp, ev['gp'])
    elif kind == 'add':
      gp.add_data_multiple([np.array(x) for x in ev['X']], list(ev['Y']))
    elif kind == 'acq':
      _same_gp(gp, ev['gp'])
      a = ev['anc']
      anc = Namespace(max_evals=a['max_evals'], t=a['t'], curr_max_val=a['curr_max_val'],
                      acq_opt_method=a['acq_opt_method'], handle_parallel=a['handle_parallel'],
                      domain=EuclideanDomain(a['bounds']), is_mf=False,
                      eval_points_in_progress=[np.array(x) for x in a['in_progress']])
      _set_state(ev['before'])
      pt = np.asarray(getattr(A.asy, ev['acq'])(gp, anc), dtype=float)
      assert np.array_equal(pt, np.array(ev['point'])), 'event %d (%s): %s != %s' % (i, ev['acq'], pt, ev['point'])
      assert _state_equal(ev['after']), 'event %d (%s): consumed different random numbers' % (i, ev['acq'])
      n_acq += 1
  return n_fit, n_acq
