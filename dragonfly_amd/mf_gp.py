"""Multi-fidelity GP on the MI355X -- host-side mirror of dragonfly/gp/mf_gp.py:18-107 (class MFGP)
and dragonfly/gp/euclidean_gp.py:347-412 (class EuclideanMFGP).  A multi-fidelity GP is a GP on the
joint (fidelity, domain) points with a product kernel; everything numerical is the same device
path as dragonfly_amd.gp_core.GP (kernel matrix, Cholesky, posterior)."""
import numpy as np

from . import kernel as gp_kernel
from .gp_core import GP


def get_ZX_from_ZZ_XX(ZZ, XX):
  """ Joint representation of fidelity and domain data (mf_gp.py:18-23): paired point by point
      when the two are sequences of equal length, otherwise one (fidelity, domain) pair. """
  paired = hasattr(ZZ, '__iter__') and len(ZZ) == len(XX)
  return list(zip(ZZ, XX)) if paired else (ZZ, XX)


class MFGP(GP):
  """ A GP to be used in multi-fidelity settings (mf_gp.py:26-107): a GP on the joint points with
      the fidelity / domain split kept alongside.  mf_kernel must be a combined kernel on the joint
      points (the Namespace form of the reference builds a CartesianProductKernel over arbitrary
      spaces, which is outside the Euclidean device path). """

  def __init__(self, ZZ, XX, YY, mf_kernel, mean_func, noise_var, *args, **kwargs):
    if not isinstance(mf_kernel, gp_kernel.Kernel) and not hasattr(mf_kernel, 'is_guaranteed_psd'):
      raise NotImplementedError('dragonfly_amd.MFGP needs a combined kernel object '
                                '(e.g. CoordinateProductKernel); there is no CPU fallback.')
    self._keep_mf_data(ZZ, XX, YY)
    super(MFGP, self).__init__(self.get_ZX_from_ZZ_XX(ZZ, XX), YY, mf_kernel, mean_func, noise_var,
                               *args, **kwargs)

  def _keep_mf_data(self, ZZ, XX, YY):
    self.ZZ, self.XX, self.YY = list(ZZ), list(XX), list(YY)

  def get_ZX_from_ZZ_XX(self, ZZ, XX):
    """ Overridden by subclasses with a more compact joint representation. """
    return get_ZX_from_ZZ_XX(ZZ, XX)

  # evaluation / sampling at (fidelity, domain) points: mf_gp.py:57-68, 89-92
  def eval_at_fidel(self, ZZ_test, XX_test, *args, **kwargs):
    return self.eval(self.get_ZX_from_ZZ_XX(ZZ_test, XX_test), *args, **kwargs)

  def eval_at_fidel_with_hallucinated_observations(self, ZZ_test, XX_test, ZZ_halluc, XX_halluc,
                                                   *args, **kwargs):
    joint_test = self.get_ZX_from_ZZ_XX(ZZ_test, XX_test)
    joint_halluc = self.get_ZX_from_ZZ_XX(ZZ_halluc, XX_halluc)
    return self.eval_with_hallucinated_observations(joint_test, joint_halluc, *args, **kwargs)

  def draw_mf_samples(self, num_samples, ZZ_test=None, XX_test=None, *args, **kwargs):
    joint = None if ZZ_test is None else self.get_ZX_from_ZZ_XX(ZZ_test, XX_test)
    return self.draw_samples(num_samples, joint, *args, **kwargs)

  # data: mf_gp.py:70-87
  def set_mf_data(self, ZZ, XX, YY, build_posterior=True):
    self._keep_mf_data(ZZ, XX, YY)
    super(MFGP, self).set_data(self.get_ZX_from_ZZ_XX(ZZ, XX), YY, build_posterior)

  def add_mf_data_multiple(self, ZZ_new, XX_new, YY_new, *args, **kwargs):
    joint_new = self.get_ZX_from_ZZ_XX(ZZ_new, XX_new)
    self.ZZ += list(ZZ_new)
    self.XX += list(XX_new)
    self.add_data_multiple(joint_new, YY_new, *args, **kwargs)

  def add_mf_data_single(self, zz_new, xx_new, yy_new, *args, **kwargs):
    self.add_mf_data_multiple([zz_new], [xx_new], [yy_new], *args, **kwargs)

  def get_fidel_kernel(self):
    return self.fidel_kernel

  def get_domain_kernel(self):
    return self.domain_kernel

  def _child_str(self):
    return 'scale: %0.3f, %s'%(self.kernel.hyperparams['scale'], str(self.kernel))


class EuclideanMFGP(MFGP):
  """ An MFGP for Euclidean fidelity and domain spaces (euclidean_gp.py:347-412): the joint point
      is the concatenation [z, x] and the kernel scale * k_fidel(z, z') * k_domain(x, x'). """

  def __init__(self, ZZ, XX, YY, mf_kernel, kernel_scale, fidel_kernel, domain_kernel,
               mean_func, noise_var, *args, **kwargs):
    if len(ZZ) != 0:
      self.fidel_dim = len(ZZ[0])
      self.domain_dim = len(XX[0])
    if fidel_kernel is not None and domain_kernel is not None:
      self.fidel_kernel = fidel_kernel
      self.domain_kernel = domain_kernel
      self.fidel_dim = fidel_kernel.dim
      self.domain_dim = domain_kernel.dim
    elif 'fidel_dim' in kwargs and 'domain_dim' in kwargs:
      self.fidel_dim = kwargs.pop('fidel_dim')
      self.domain_dim = kwargs.pop('domain_dim')
    else:
      raise Exception('Specify fidel_dim and domain_dim.')
    self.fidel_coords = list(range(self.fidel_dim))
    self.domain_coords = list(range(self.fidel_dim, self.fidel_dim + self.domain_dim))
    if mf_kernel is None:
      mf_kernel = gp_kernel.CoordinateProductKernel(self.fidel_dim + self.domain_dim,
                                                    kernel_scale, [fidel_kernel, domain_kernel],
                                                    [self.fidel_coords, self.domain_coords])
    super(EuclideanMFGP, self).__init__(ZZ, XX, YY, mf_kernel, mean_func, noise_var,
                                        *args, **kwargs)

  def _test_fidel_domain_dims(self, test_fidel_dim, test_domain_dim):
    """ euclidean_gp.py:379-385 """
    if test_fidel_dim != self.fidel_dim or test_domain_dim != self.domain_dim:
      raise ValueError('ZZ, XX dimensions should be (%d, %d). Given (%d, %d)'%( \
                       self.fidel_dim, self.domain_dim, test_fidel_dim, test_domain_dim))

  def get_ZX_from_ZZ_XX(self, ZZ, XX):
    """ euclidean_gp.py:387-403 """
    ordering = np.argsort(self.fidel_coords + self.domain_coords)
    if hasattr(ZZ, '__iter__') and len(ZZ) == 0:
      return []
    if hasattr(ZZ[0], '__iter__'):
      self._test_fidel_domain_dims(len(ZZ[0]), len(XX[0]))
      ZX_unordered = np.concatenate((np.array(ZZ), np.array(XX)), axis=1)
      return list(ZX_unordered[:, ordering])
    self._test_fidel_domain_dims(len(ZZ), len(XX))
    return np.concatenate((ZZ, XX))[ordering]

  def get_domain_pts(self, data_idxs=None):
    data_idxs = data_idxs if data_idxs is not None else range(self.num_tr_data)
    return [self.XX[i] for i in data_idxs]

  def get_fidel_pts(self, data_idxs=None):
    data_idxs = data_idxs if data_idxs is not None else range(self.num_tr_data)
    return [self.ZZ[i] for i in data_idxs]
