"""MI355X: a multi-step ask/tell trajectory of the REAL reference's EuclideanGPBandit
(opt/gp_bandit.py:551; Branin, Matern kernel, acquisitions ucb / ei / ts / ttei chosen by its own
ensemble rule, a new model every four steps, data appended in between), recorded call by call with
the RNG states (oracle/make_golden.py: gen_trajectory_case), replayed through the mirrors on
libdfhip.so: the fitter picks the reference's hyper-parameters, every acquisition returns the
reference's point, and every call leaves the global RNG where the reference left it -- the S3 / S4
seams pinned on hardware, not only against the CPU stand-in."""
import pytest

from trajectory_replay import replay

pytestmark = pytest.mark.gpu


def test_reference_trajectory_replayed_on_the_device(engine):
  n_fit, n_acq = replay()
  assert n_fit == 4 and n_acq == 14
