"""MI355X: the multi-fidelity fitter mirror (dragonfly_amd.mf_gp.EuclideanMFGPFitter) with its
tuning objective evaluated by dfh_gp_lml_batch -- product kernels with SE / Matern / exponential-decay
factors -- against the real reference's fitter under the same seed."""
import pytest

from mf_fitter_replay import CASES, check, check_additive_domain

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', sorted(CASES))
def test_mf_fitter_picks_the_reference_hyperparameters(engine, name):
  check(name)


def test_mf_fitter_with_an_additive_domain_model(engine):
  """ euclidean_gp.py:696-707: the joint kernel is a product with an ADDITIVE factor -- flattened into
      struct dfh_kernel_desc (group_factor / factor_is_sum / factor_scale) and evaluated on the device """
  check_additive_domain()
