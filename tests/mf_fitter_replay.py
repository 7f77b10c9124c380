"""Checks the mirror of EuclideanMFGPFitter (dragonfly_amd/mf_gp.py; reference
dragonfly/gp/euclidean_gp.py:418-710) against the real reference's fitter under the same seed
(tests/golden/mf_fitter_f2_d3_n48.npz, oracle/make_golden.py: gen_mf_fitter_case): the same
hyper-parameter boxes, the same random candidates, the same winner, the same fitted GP, and the
global random stream left where the reference leaves it.  Shared by the CPU plumbing test
(stand-in engine) and the MI355X test."""
from argparse import Namespace

import numpy as np

from conftest import load_golden, relerr

CASES = {
  'se_se': dict(fidel_kernel_type='se', domain_kernel_type='se'),
  'expdecay_se': dict(fidel_kernel_type='expdecay', domain_kernel_type='se'),
  'matern_matern': dict(fidel_kernel_type='matern', domain_kernel_type='matern', fidel_matern_nu=-1.0,
                        domain_matern_nu=1.5),
}


def check(name, tol=1e-10):
  from dragonfly_amd.mf_gp import EuclideanMFGPFitter, EuclideanMFGP
  g = load_golden('mf_fitter_f2_d3_n48')
  opts = Namespace(ml_hp_tune_opt='rand', hp_tune_max_evals=60, hp_tune_criterion='ml', **CASES[name])
  np.random.seed(1618)
  fitter = EuclideanMFGPFitter(list(g['ZZ']), list(g['XX']), list(g['YY']), options=opts)
  assert np.array_equal(np.array(fitter.cts_hp_bounds, dtype=float), g[name + '_bounds'])
  kind, gp, hps = fitter.fit_gp()
  assert kind == 'fitted_gp' and isinstance(gp, EuclideanMFGP)
  assert np.array_equal(np.array(hps[0], dtype=float), g[name + '_cts_hps'])
  assert np.array_equal(np.array(hps[1], dtype=float), g[name + '_dscr_hps'])
  assert abs(gp.compute_log_marginal_likelihood() - float(g[name + '_lml'])) <= tol * abs(float(g[name + '_lml']))
  assert gp.noise_var == float(g[name + '_noise']) and gp.kernel.hyperparams['scale'] == float(g[name + '_scale'])
  mu, sd = gp.eval_at_fidel(list(g['Zs']), list(g['Xs']), 'std')
  assert relerr(mu, g[name + '_mu']) < tol and relerr(sd, g[name + '_sd']) < tol
  assert np.random.random() == float(g[name + '_rand_after'])
  return fitter, gp


MF_ADDITIVE_OPTS = dict(ml_hp_tune_opt='rand', hp_tune_max_evals=40, hp_tune_criterion='ml', fidel_kernel_type='se',
                        domain_kernel_type='se', domain_use_additive_gp=True, domain_add_max_group_size=3,
                        domain_num_groups_per_group_size=2)


def check_additive_domain(tol=1e-10):
  """ The fitter with an additive domain model (euclidean_gp.py:480-486, 622-633, 696-707): a
      product kernel with an additive factor, evaluated on the device like any other; same random
      groupings and candidates, same winner, same joint Gram matrix and predictions as the reference
      (tests/golden/mf_fitter_additive_f1_d6_n40.npz, oracle/make_golden.py). """
  from dragonfly_amd.mf_gp import EuclideanMFGPFitter, EuclideanMFGP
  g = load_golden('mf_fitter_additive_f1_d6_n40')
  np.random.seed(99)
  fitter = EuclideanMFGPFitter(list(g['ZZ']), list(g['XX']), list(g['YY']), options=Namespace(**MF_ADDITIVE_OPTS))
  assert np.array_equal(np.array(fitter.cts_hp_bounds, dtype=float), g['bounds'])
  kind, gp, hps = fitter.fit_gp()
  assert kind == 'fitted_gp' and isinstance(gp, EuclideanMFGP) and not gp._generic     # on the device path
  assert np.array_equal(np.array(hps[0], dtype=float), g['cts_hps'])
  assert np.array_equal(np.array(hps[1], dtype=float), g['dscr_hps'])
  dk = gp.kernel.kernel_list[1]
  assert type(dk).__name__ == 'AdditiveKernel'
  assert np.array_equal([len(grp) for grp in dk.groupings], g['group_sizes'])
  assert np.array_equal([int(c) for grp in dk.groupings for c in grp], g['groupings_flat'])
  assert gp.kernel.has_device_spec()
  joint = np.concatenate((g['ZZ'], g['XX']), axis=1)
  assert relerr(gp.kernel(joint, joint), g['K']) < 1e-13
  assert relerr(gp.kernel(np.concatenate((g['Zs'], g['Xs']), axis=1), joint), g['K_cross']) < 1e-13
  assert abs(gp.compute_log_marginal_likelihood() - float(g['lml'])) <= tol * abs(float(g['lml']))
  assert gp.noise_var == float(g['noise']) and gp.kernel.hyperparams['scale'] == float(g['scale'])
  mu, sd = gp.eval_at_fidel(list(g['Zs']), list(g['Xs']), 'std')
  assert relerr(mu, g['mu']) < tol and relerr(sd, g['sd']) < tol
  assert np.random.random() == float(g['rand_after'])
  return fitter, gp
