"""MI355X: the multi-fidelity fitter mirror (dragonfly_amd.mf_gp.EuclideanMFGPFitter) with its
tuning objective evaluated by dfh_gp_lml_batch -- product kernels with SE / Matern / exponential-decay
factors -- against the real reference's fitter under the same seed."""
import pytest

from mf_fitter_replay import CASES, check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', sorted(CASES))
def test_mf_fitter_picks_the_reference_hyperparameters(engine, name):
  check(name)
