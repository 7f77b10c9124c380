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
