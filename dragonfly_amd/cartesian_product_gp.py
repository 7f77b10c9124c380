"""GPs on Cartesian-product domains on the MI355X -- host-side mirror of the GP class of
dragonfly/gp/cartesian_product_gp.py:208-248 (class CPGP).

A CPGP's kernel is a product over the parts of the domain (Euclidean, integral, discrete, neural-
network, ...), not guaranteed to be positive semi-definite, so the reference builds its posterior
through _get_cholesky_decomp's 'project_first' branch (gp_core.py:838-841: eigen-projection of the
Gram matrix onto the PSD cone) and projects every posterior covariance as well (gp_core.py:849-857).
Here the kernel object stays whatever the caller built (the reference's CartesianProductKernel and
its parts are evaluated on the host, also from pre-computed distance lists), while the projection
(csrc/psdproj.hip), the factorisation, the solves and the posterior run on the device: the GP of
gp_core.py in host-kernel mode with handle_non_psd_kernels='project_first'.

dragonfly_amd.install(cartesian_product=True) rebinds dragonfly.gp.cartesian_product_gp.CPGP to
this class: the reference's CPGPFitter constructs its GPs through that module global
(cartesian_product_gp.py: `CPGP(X, Y, kernel, mean_func, noise_var, ...)`), so Cartesian-product
runs -- the reference's default for every non-Euclidean domain -- get the device posterior.
"""
import numpy as np

from .gp_core import GP


def get_idxs_from_list_of_lists(list_of_lists, idx):
  """ utils/general_utils.py:43-45 """
  return [elem[idx] for elem in list_of_lists]


class CPGP(GP):
  """ cartesian_product_gp.py:208-248 """
  # pylint: disable=attribute-defined-outside-init

  def __init__(self, X, Y, kernel, mean_func, noise_var, domain_lists_of_dists=None,
               build_posterior=True, reporter=None, handle_non_psd_kernels='project_first'):
    if domain_lists_of_dists is None:
      domain_lists_of_dists = [None] * kernel.num_kernels
    self.domain_lists_of_dists = domain_lists_of_dists
    super(CPGP, self).__init__(X, Y, kernel, mean_func, noise_var, build_posterior, reporter,
                               handle_non_psd_kernels)

  def set_domain_lists_of_dists(self, domain_lists_of_dists):
    """ cartesian_product_gp.py:226-228 """
    self.domain_lists_of_dists = domain_lists_of_dists

  def _child_str(self):
    """ cartesian_product_gp.py:230-236 """
    mean_str = 'mu[#0]=%0.4f, ' % (self.mean_func([self.X[0]])[0]) if len(self.X) > 0 else ''
    return mean_str + str(self.kernel)

  def _get_training_kernel_matrix(self):
    """ cartesian_product_gp.py:238-248: scale * prod_parts k_part, a part from its distance list
        when one was given.  (Overriding this documented hook is what puts the GP into host-kernel
        mode: dfh_gp_fit_gram with DFH_FIT_PROJECT_FIRST.) """
    n = len(self.X)
    ret = self.kernel.hyperparams['scale'] * np.ones((n, n))
    for idx, kern in enumerate(self.kernel.kernel_list):
      if self.domain_lists_of_dists[idx] is not None:
        ret *= kern.evaluate_from_dists(self.domain_lists_of_dists[idx])
      else:
        curr_X = get_idxs_from_list_of_lists(self.X, idx)
        ret *= kern(curr_X, curr_X)
    return ret
