"""Random-search maximisers, mirror of dragonfly/utils/oper_utils.py:59-80.  Candidate
generation uses the global np.random state with the reference's exact call, so a seeded run
draws the same candidates as the reference."""
from argparse import Namespace

import numpy as np

from .general_utils import map_to_bounds


def random_sample(obj, bounds, max_evals, vectorised=True):
  """ oper_utils.py:59-67 """
  dim = len(bounds)
  rand_pts = map_to_bounds(np.random.random((int(max_evals), dim)), bounds)
  if vectorised:
    obj_vals = obj(rand_pts)
  else:
    obj_vals = np.array([obj(x) for x in rand_pts])
  return rand_pts, obj_vals


def random_maximise(obj, bounds, max_evals, return_history=False, vectorised=True):
  """ oper_utils.py:70-80 """
  rand_pts, obj_vals = random_sample(obj, bounds, max_evals, vectorised)
  max_idx = obj_vals.argmax()
  max_val = obj_vals[max_idx]
  max_pt = rand_pts[max_idx]
  if return_history:
    history = Namespace(query_vals=obj_vals, query_points=rand_pts)
  else:
    history = None
  return max_val, max_pt, history


def random_sample_cts_dscr(obj, cts_bounds, dscr_vals, max_evals, vectorised=True):
  """ oper_utils.py:random_sample_cts_dscr -- continuous + discrete random sampling used by
      the rand_exp_sampling hyper-parameter tuner. """
  dim = len(cts_bounds)
  cts_rand_pts = map_to_bounds(np.random.random((int(max_evals), dim)), cts_bounds) \
                 if dim > 0 else np.zeros((int(max_evals), 0))
  dscr_rand_pts = [[np.random.choice(vals) for vals in dscr_vals] for _ in range(int(max_evals))]
  if vectorised:
    obj_vals = obj(cts_rand_pts, dscr_rand_pts)
  else:
    obj_vals = np.array([obj(c, d) for c, d in zip(cts_rand_pts, dscr_rand_pts)])
  return cts_rand_pts, dscr_rand_pts, obj_vals


class EuclideanDomain(object):
  """ Minimal stand-in for dragonfly/exd/domains.py:EuclideanDomain (get_type, bounds, dim). """

  def __init__(self, bounds):
    self.bounds = np.array(bounds, dtype=float)
    self.dim = len(self.bounds)

  def get_type(self):
    return 'euclidean'

  def get_dim(self):
    return self.dim

  def is_a_member(self, point):
    point = np.asarray(point)
    return bool(np.all(point >= self.bounds[:, 0]) and np.all(point <= self.bounds[:, 1]))
