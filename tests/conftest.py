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


@pytest.fixture(scope='session')
def engine():
  """ The process-wide Engine on cuda:0 / LOCAL_RANK; GPU tests fail loudly without a device. """
  from dragonfly_amd.engine import get_engine
  return get_engine()


GP_CASES = ['se_d2_n40', 'se_ard_d5_n50', 'matern25_d6_n60', 'matern15_d3_n45', 'matern05_d2_n30',
            'se_d32_n130', 'additive_d10_n80']
