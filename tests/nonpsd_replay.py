"""Checks the mirrors' GP / CPGP for kernels that are not guaranteed PSD against the outputs of the
real reference (tests/golden/nonpsd_gp.npz, oracle/make_golden.py: gen_nonpsd_cases).  Shared by
the CPU plumbing test (stand-in engine) and the MI355X test."""
import numpy as np

from conftest import load_golden, relerr


class SigmoidKernel(object):
  """ k(x, y) = tanh(a x.y + b): the indefinite test kernel of the fixture, evaluated on the host
      (a caller's kernel object: the GP runs in host-kernel mode). """

  def __init__(self, a, b):
    self.hyperparams = dict(a=a, b=b)

  def is_guaranteed_psd(self):
    return False

  def __call__(self, X1, X2=None):
    X2 = X1 if X2 is None else X2
    X1, X2 = np.array(X1, dtype=float), np.array(X2, dtype=float)
    return np.tanh(self.hyperparams['a'] * X1.dot(X2.T) + self.hyperparams['b'])

  def __str__(self):
    return 'sigmoid'


class ProductOverParts(object):
  """ scale * prod_i k_i(part i): what the reference's CartesianProductKernel computes
      (gp/kernel.py:504-538), over the caller's part kernels. """

  def __init__(self, scale, kernel_list):
    self.kernel_list, self.num_kernels = kernel_list, len(kernel_list)
    self.hyperparams = dict(scale=scale)

  def is_guaranteed_psd(self):
    return all(k.is_guaranteed_psd() for k in self.kernel_list)

  def __call__(self, X1, X2=None):
    X2 = X1 if X2 is None else X2
    out = self.hyperparams['scale'] * np.ones((len(X1), len(X2)))
    for idx, kern in enumerate(self.kernel_list):
      out *= kern([x[idx] for x in X1], [x[idx] for x in X2])
    return out

  def __str__(self):
    return 'DomProd'


def _reference_cpgp_module():
  """ dragonfly.gp.cartesian_product_gp where a checkout is present (the build container), else None """
  import os
  import sys
  ref = os.environ.get('DRAGONFLY_REFERENCE', '/root/reference')
  if not os.path.isdir(os.path.join(ref, 'dragonfly')):
    return None
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
  import make_golden
  make_golden.import_reference()
  import dragonfly.gp.cartesian_product_gp as ref_cpgp
  return ref_cpgp


def check(tol=1e-10):
  from dragonfly_amd.gp_core import GP
  from dragonfly_amd.cartesian_product_gp import device_cpgp_class
  from dragonfly_amd import kernel as K
  g = load_golden('nonpsd_gp')
  assert float(g['min_eig_K']) < -1.0 and float(g['cp_min_eig_K']) < -1.0      # genuinely indefinite
  mean_c, noise = float(g['mean_c']), float(g['noise'])
  for mode in ('project_first', 'try_before_project'):
    gp = GP(list(g['X']), list(g['Y']), SigmoidKernel(float(g['a']), float(g['b'])),
            lambda x, _c=mean_c: np.array([_c] * len(x)), noise, handle_non_psd_kernels=mode)
    assert relerr(gp.K_trtr_wo_noise, g[mode + '_K']) < 1e-13
    assert relerr(gp.L, g[mode + '_L']) < tol and relerr(gp.alpha, g[mode + '_alpha']) < tol
    assert abs(gp.compute_log_marginal_likelihood() - float(g[mode + '_lml'])) < tol * abs(float(g[mode + '_lml']))
    mu, sd = gp.eval(list(g['Xs']), 'std')
    assert relerr(mu, g[mode + '_mu']) < tol and relerr(sd, g[mode + '_sd']) < tol
    _, cov = gp.eval(list(g['Xs']), 'covar')
    assert relerr(cov, g[mode + '_cov']) < tol
    _, sdh = gp.eval_with_hallucinated_observations(list(g['Xs']), list(g['Xh']), 'std')
    assert relerr(sdh, g[mode + '_sdh']) < tol
  # CPGP: R^2 x R^1, SE part on the device kernels, sigmoid part on the host
  lists = lambda A, B: [[A[i], B[i]] for i in range(len(A))]
  kern = ProductOverParts(float(g['cp_scale']), [K.SEKernel(2, 1.0, g['cp_bw0']),
                                                 SigmoidKernel(float(g['cp_a']), float(g['cp_b']))])
  mean2 = float(g['cp_mean'])
  ref_cpgp = _reference_cpgp_module()
  if ref_cpgp is not None:
    # the class install(cartesian_product=True) binds: the reference's own class body over the device GP
    CPGP = device_cpgp_class(ref_cpgp)
    assert CPGP.__mro__[1] is GP and CPGP._get_training_kernel_matrix.__code__ is ref_cpgp.CPGP._get_training_kernel_matrix.__code__
    gp = CPGP(lists(g['cp_P0'], g['cp_P1']), list(g['cp_Y']), kern, lambda x: np.array([mean2] * len(x)),
              float(g['cp_noise']))
  else:
    # no Dragonfly checkout (the GPU box): the same device route -- host-kernel mode with 'project_first', which is
    # all the class adds to the device GP -- with the Gram matrix of the same product kernel
    gp = GP(lists(g['cp_P0'], g['cp_P1']), list(g['cp_Y']), kern, lambda x: np.array([mean2] * len(x)),
            float(g['cp_noise']), handle_non_psd_kernels='project_first')
  assert gp.handle_non_psd_kernels == 'project_first'
  assert relerr(gp.K_trtr_wo_noise, g['cp_K']) < 1e-12
  assert relerr(gp.L, g['cp_L']) < tol and relerr(gp.alpha, g['cp_alpha']) < tol
  assert abs(gp.compute_log_marginal_likelihood() - float(g['cp_lml'])) < tol * abs(float(g['cp_lml']))
  Xt = lists(g['cp_T0'], g['cp_T1'])
  mu, sd = gp.eval(Xt, 'std')
  assert relerr(mu, g['cp_mu']) < tol and relerr(sd, g['cp_sd']) < tol
  _, cov = gp.eval(Xt, 'covar')
  assert relerr(cov, g['cp_cov']) < tol
  _, sdh = gp.eval_with_hallucinated_observations(Xt, lists(g['cp_H0'], g['cp_H1']), 'std')
  assert relerr(sdh, g['cp_sdh']) < tol
  assert ref_cpgp is None or 'DomProd' in str(gp)
