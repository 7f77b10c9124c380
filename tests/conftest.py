import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950); run with -m gpu')


def load_golden(name):
  return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


def relerr(a, b):
  """ norm-wise relative error max|a-b| / max|b| """
  a = np.asarray(a, dtype=float)
  b = np.asarray(b, dtype=float)
  den = np.max(np.abs(b)) if b.size else 1.0
  return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0)) if b.size else 0.0


ELEM_MASK = 1e-3


def relerr_elem(a, b):
  """ element-wise relative error max |a_i - b_i| / |b_i| over the entries with |b_i| >= 1e-3 max|b|
      (reported beside the norm-wise figure) """
  a = np.asarray(a, dtype=float).ravel()
  b = np.asarray(b, dtype=float).ravel()
  if not b.size:
    return 0.0
  keep = (np.abs(b) >= ELEM_MASK * np.max(np.abs(b))) & (b != 0)
  return float(np.max(np.abs(a[keep] - b[keep]) / np.abs(b[keep]))) if keep.any() else 0.0


def pytest_sessionfinish(session, exitstatus):
  """ tests/truth_bounds.py: which parity bounds of this run were wider than 1e-10, and by how much. """
  del session, exitstatus
  import json
  tb = sys.modules.get('truth_bounds')
  if tb is None or not tb.APPLIED:
    return
  summary = tb.applied_summary()
  print('\n[truth_bounds] %d bounds above 1e-10 in %d tests; largest applied %.2e (ceiling %.0e)'
        % (summary['bounds_above_floor'], summary['tests_with_relaxed_bounds'], summary['largest_bound_applied'], summary['ceiling']))
  try:
    out_dir = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, 'truth_bounds_applied.json'), 'w') as f:
      json.dump(summary, f, indent=1)
  except OSError:
    pass


@pytest.fixture(scope='session')
def engine():
  """ The process-wide Engine on cuda:0 / LOCAL_RANK; GPU tests fail loudly without a device. """
  from dragonfly_amd.engine import get_engine
  return get_engine()


GP_CASES = ['se_d2_n40', 'se_ard_d5_n50', 'matern25_d6_n60', 'matern15_d3_n45', 'matern05_d2_n30',
            'se_d32_n130', 'additive_d10_n80']
