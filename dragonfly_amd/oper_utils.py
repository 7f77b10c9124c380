"""Random-search maximisers, mirror of dragonfly/utils/oper_utils.py:59-80.  Candidate
generation uses the global np.random state with the reference's exact call, so a seeded run
draws the same candidates as the reference."""
from argparse import Namespace

import numpy as np

from .doo import pdoo_maximise, pdoo_maximise_batched, pdoo_minimise    # pylint: disable=unused-import
from . import gaplog
from .general_utils import map_to_bounds


def _evaluate(obj, pts, vectorised):
  """ Objective values at the rows of pts: one call, or one call per point. """
  if vectorised:
    return obj(pts)
  return np.array([obj(pt) for pt in pts])


def random_sample(obj, bounds, max_evals, vectorised=True):
  """ max_evals uniform points in the box `bounds` and the objective there (oper_utils.py:59-67).
      The draw is np.random.random((max_evals, dim)) mapped to the bounds -- the reference's call,
      so a seeded run sees the same points. """
  pts = map_to_bounds(np.random.random((int(max_evals), len(bounds))), bounds)
  return pts, _evaluate(obj, pts, vectorised)


def random_maximise(obj, bounds, max_evals, return_history=False, vectorised=True):
  """ Best of a random sample (oper_utils.py:70-80): np.argmax's rule -- the first maximum, a NaN
      beats everything -- picks the winner.  Returns (value, point, history or None). """
  pts, vals = random_sample(obj, bounds, max_evals, vectorised)
  gaplog.top2('random_maximise', vals)
  best = vals.argmax()
  history = Namespace(query_vals=vals, query_points=pts) if return_history else None
  return vals[best], pts[best], history


def random_sample_cts_dscr(obj, cts_bounds, dscr_vals, max_evals, vectorised=True):
  """ Random continuous x discrete samples for the rand_exp_sampling hyper-parameter tuner
      (oper_utils.py:100-112): the continuous block is drawn first, then one np.random.choice per
      discrete parameter and sample. """
  num = int(max_evals)
  if len(cts_bounds) > 0:
    cts_pts = map_to_bounds(np.random.random((num, len(cts_bounds))), cts_bounds)
  else:
    cts_pts = np.zeros((num, 0))
  dscr_pts = [[np.random.choice(vals) for vals in dscr_vals] for _ in range(num)]
  if vectorised:
    obj_vals = obj(cts_pts, dscr_pts)
  else:
    obj_vals = np.array([obj(c, d) for c, d in zip(cts_pts, dscr_pts)])
  return cts_pts, dscr_pts, obj_vals


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
