"""MI355X: from n = 2048 on the fit writes the 64 x 64 tiles on and below the diagonal of K + noise I only (round 5;
csrc/api.hip: dfh_gp_fit, _get_training_kernel_matrix + stable_cholesky of dragonfly/gp/gp_core.py:155-160, 827-847) and
the tiles above keep what the recycled buffer held.  Nothing may read them: with the buffer filled with NaN first
(DFH_TEST_POISON_L=1) the factor, alpha, lml, GP.eval, the hallucinated posterior, a joint Thompson block and an append
are the same as with the full symmetric build (DFH_KM_LOWER_ONLY=0) -- at a size that is a multiple of the tiles and at
sizes that are not a multiple of 64 or 128 (advisor, round 5)."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _run(n, path, env_extra):
  env = dict(os.environ)
  env.update(env_extra)
  res = subprocess.run([sys.executable, os.path.join(HERE, 'upper_triangle_check.py'), str(n), path], env=env,
                       capture_output=True, text=True, timeout=600)
  assert res.returncode == 0 and res.stdout.strip().endswith('OK'), (res.stdout[-2000:], res.stderr[-4000:])
  return dict(np.load(path))


@pytest.mark.parametrize('n', [2048, 2111, 2625, 4160])
def test_poisoned_upper_triangle_changes_nothing(tmp_path, n):
  full = _run(n, str(tmp_path / 'full.npz'), {'DFH_KM_LOWER_ONLY': '0'})
  poisoned = _run(n, str(tmp_path / 'poisoned.npz'), {'DFH_TEST_POISON_L': '1'})
  for key, want in full.items():
    got = poisoned[key]
    assert np.all(np.isfinite(got)), key
    assert np.array_equal(got, want), (key, float(np.max(np.abs(got - want))))
