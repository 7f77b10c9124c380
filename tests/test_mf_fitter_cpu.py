"""CPU: the multi-fidelity fitter mirror over the stand-in engine against the real reference's
fitter (same seed, same choices), and its set-up / error behaviour."""
from argparse import Namespace

import numpy as np
import pytest

from mf_fitter_replay import CASES, check, check_additive_domain


@pytest.mark.parametrize('name', sorted(CASES))
def test_mf_fitter_picks_the_reference_hyperparameters(monkeypatch, name):
  from oracle_engine import patch_engine
  from dragonfly_amd import mf_gp    # noqa: F401  (imports gp_core before patching)
  patch_engine(monkeypatch)
  fitter, gp = check(name)
  assert fitter.fidel_dim == 2 and fitter.domain_dim == 3 and fitter.dim == 5
  assert gp.fidel_kernel.dim == 2 and gp.domain_kernel.dim == 3
  assert 'scale' in str(gp)


def test_mf_fitter_with_an_additive_domain_model(monkeypatch):
  """ the additive-domain fitter over the stand-in engine: the reference's groupings, candidates and
      winner; the nested kernel description (a product with an additive factor) round-trips """
  from oracle_engine import patch_engine, to_oracle_spec
  from dragonfly_amd import mf_gp    # noqa: F401
  patch_engine(monkeypatch)
  fitter, gp = check_additive_domain()
  spec = gp.kernel.to_spec()
  assert spec.kind == 'product' and list(spec.group_factors) == [0, 1, 1, 1] and list(spec.factor_sums) == [False, True]
  assert [list(g) for g in spec.groups] == [[0], [5, 4], [6, 3], [2, 1]]       # absolute columns: fidelity first
  joint = np.concatenate((fitter.ZZ, fitter.XX), axis=1)
  assert np.array_equal(to_oracle_spec(spec)(joint, joint), gp.kernel._host_compose(joint, joint))


def test_mf_fitter_set_up_errors_and_bandit_interface(monkeypatch):
  from oracle_engine import patch_engine
  from dragonfly_amd.mf_gp import EuclideanMFGPFitter
  patch_engine(monkeypatch)
  rs = np.random.RandomState(1)
  ZZ, XX = list(rs.random_sample((12, 1))), list(rs.random_sample((12, 2)))
  YY = list(rs.randn(12))
  with pytest.raises(ValueError):
    EuclideanMFGPFitter(ZZ, XX, YY, options=Namespace(fidel_kernel_type='spline'))
  with pytest.raises(ValueError):
    EuclideanMFGPFitter(ZZ, XX, YY, options=Namespace(domain_kernel_type='expdecay'))
  with pytest.raises(NotImplementedError):          # the reference has no bounds for polynomial kernels either
    EuclideanMFGPFitter(ZZ, XX, YY, options=Namespace(fidel_kernel_type='poly'))
  # hyper-parameter layout with the exponential-decay fidelity kernel: scale, offset, power, 2 bandwidths
  f = EuclideanMFGPFitter(ZZ, XX, YY, options=Namespace(fidel_kernel_type='expdecay', mean_func_type='median',
                                                        noise_var_type='label', ml_hp_tune_opt='rand',
                                                        hp_tune_max_evals=20))
  assert len(f.cts_hp_bounds) == 1 + 2 + 2 and f.num_hps == 5
  assert np.allclose(f.cts_hp_bounds[2], [np.log(0.1), np.log(50)])
  # the bandit-facing calls (gp_core.py:728-781)
  np.random.seed(3)
  f.fit_gp_for_gp_bandit(num_samples=1)
  fit_type, method, gp = f.get_next_gp()
  assert fit_type == 'fitted_gp' and method == 'ml' and len(gp.ZZ) == 12
  # additive domain kernel (euclidean_gp.py:696-707): a product kernel with an additive factor, on the device too
  np.random.seed(4)
  fa = EuclideanMFGPFitter(ZZ, XX, YY, options=Namespace(domain_use_additive_gp=True, ml_hp_tune_opt='rand',
                                                         hp_tune_max_evals=6, domain_num_groups_per_group_size=1))
  kind, gpa, hps = fa.fit_gp()
  assert kind == 'fitted_gp' and not gpa._generic and len(hps[1]) == 1
