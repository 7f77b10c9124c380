"""TEST INFRASTRUCTURE ONLY: closed-form objectives shared by oracle/make_golden.py (which runs the
reference's maximisers on them) and the tests (which run ours on the same functions)."""
import numpy as np


def wavy_bowl(x):
  """ A multi-modal bowl on any dimension; x is one point. """
  x = np.asarray(x, dtype=np.float64).ravel()
  return float(-np.sum((x - 0.3) ** 2) + 0.3 * np.sin(9 * x[0]) * np.cos(7 * x[-1]))


def ridge(x):
  """ Plateaus and exact ties (values repeat across symmetric boxes): exercises the tie rules. """
  x = np.asarray(x, dtype=np.float64).ravel()
  return float(np.round(np.cos(3 * x).sum(), 1))


PDOO_CASES = [
  # name, objective, bounds, budget
  ('wavy_d2', wavy_bowl, [[-1.0, 2.0]] * 2, 200),
  ('wavy_d3', wavy_bowl, [[-1.0, 2.0], [0.0, 1.0], [-3.0, 0.5]], 500),
  ('wavy_d6', wavy_bowl, [[-1.0, 2.0]] * 6, 1200),
  ('wavy_d1', wavy_bowl, [[-1.0, 2.0]], 50),
  ('ridge_d2', ridge, [[-2.0, 2.0]] * 2, 300),
  ('ridge_d4', ridge, [[-2.0, 2.0], [-1.0, 1.0], [0.0, 4.0], [-2.0, 2.0]], 700),
]


def bimodal_logp(x):
  """ An unnormalised two-bump log density with hard support [-4, 6] (-inf outside, as a bounded
      prior gives): exercises stepping out over the support edge and long shrink chains. """
  x = float(np.ravel(x)[0])
  if x < -4.0 or x > 6.0:
    return -np.inf
  return float(np.logaddexp(-0.5 * ((x + 1.0) / 0.3) ** 2, -0.5 * ((x - 2.5) / 0.8) ** 2 - 0.7))


def heavy_tail_logp(x):
  """ Student-t like, unbounded support: the slice is often much wider than w. """
  x = float(np.ravel(x)[0])
  return float(-2.0 * np.log1p(x * x / 3.0))


SLICE_CASES = [
  # name, log density, start, kept samples, burn, seed
  ('bimodal', bimodal_logp, 0.3, 200, 50, 11),
  ('heavy_tail', heavy_tail_logp, -7.0, 300, 20, 12),
]
