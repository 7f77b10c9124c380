"""Euclidean kernels evaluated on the MI355X: SE, Matern and Additive.

Host-side mirror of dragonfly/gp/kernel.py (reference lines in the docstrings): same class
names, constructor arguments, `hyperparams` dictionary and error behaviour, so code written
against the reference's kernels runs unchanged.  The Gram / cross matrices themselves are
produced by libdfhip.so (csrc/kernmat.hip) -- there is no NumPy evaluation path here.
"""
import numpy as np

from .engine import KernelSpec, get_engine


def _as_2d_array(X):
  """ The reference passes lists of 1-D arrays (gp_core.py:129-130) or 2-D arrays. """
  X = np.asarray(X, dtype=np.float64)
  if X.ndim == 1:
    X = X.reshape((len(X), -1)) if X.size else X.reshape((0, 0))
  return np.ascontiguousarray(X)


def _get_se_matern_scale_bw_strs(kern):
  """ kernel.py:48-57 """
  if kern.dim > 6:
    bw_str = 'avg-bw: %0.4f'%(kern.hyperparams['dim_bandwidths'].mean())
  else:
    bw_str = 'bws:[' + ' '.join(['%0.2f'%(dbw) for dbw in
                                 kern.hyperparams['dim_bandwidths']]) + ']'
  scale_str = 'sc:%0.4f'%(kern.hyperparams['scale'])
  return scale_str, bw_str


class Kernel(object):
  """ A kernel class (kernel.py:60-129). """

  def __init__(self):
    super(Kernel, self).__init__()
    self.hyperparams = {}

  def is_guaranteed_psd(self):
    raise NotImplementedError('Implement in a child class.')

  def __call__(self, X1, X2=None):
    return self.evaluate(X1, X2)

  def evaluate(self, X1, X2=None):
    """ kernel.py:76-83: n1 x n2 kernel matrix; empty inputs give an empty matrix. """
    X2 = X1 if X2 is None else X2
    if len(X1) == 0 or len(X2) == 0:
      return np.zeros((len(X1), len(X2)))
    return self._child_evaluate(X1, X2)

  def _child_evaluate(self, X1, X2):
    raise NotImplementedError('Implement in a child class.')

  def set_hyperparams(self, **kwargs):
    self.hyperparams = kwargs

  def add_hyperparams(self, **kwargs):
    for key, value in kwargs.items():
      self.hyperparams[key] = value

  def to_spec(self, in_dim=None):
    """ Description handed to the C-ABI (struct dfh_kernel_desc). in_dim: number of columns of
        the inputs (only an additive kernel cannot tell it from its own parameters). """
    raise NotImplementedError('Implement in a child class.')

  def __str__(self):
    return '%s:: %s'%(type(self), str(self.hyperparams))


class _EuclideanDeviceKernel(Kernel):
  """ Shared device evaluation of SE / Matern / Additive kernels. """

  def has_device_spec(self):
    """ False for a grouped kernel with a factor the device does not evaluate (e.g. the reference's
        PolyKernel inside a product): it is then composed on the host from its factors' own
        evaluations, and GPs using it run in host-kernel mode (gp_core.GP._generic). """
    return True

  def _child_evaluate(self, X1, X2):
    X1a = _as_2d_array(X1)
    same = X2 is X1
    X2a = X1a if same else _as_2d_array(X2)
    if X1a.shape[1] != X2a.shape[1]:
      raise ValueError('Second dimension of X1 and X2 should be equal.')   # general_utils.py:64
    if not self.has_device_spec():
      return self._host_compose(X1a, X2a)
    return get_engine().kernel_matrix(self.to_spec(in_dim=X1a.shape[1]), X1a,
                                      None if same else X2a)

  def get_scaled_repr(self, X):
    """ kernel.py:179-181 / 255-257 """
    return X/self.hyperparams['dim_bandwidths']

  def change_smoothness(self, factor):
    """ kernel.py:198-200 """
    self.hyperparams['dim_bandwidths'] *= factor

  def get_effective_norm(self, X, order=None, is_single=True):
    """ kernel.py:183-190 """
    scaled_X = self.get_scaled_repr(X)
    if is_single:
      return np.linalg.norm(scaled_X, ord=order)
    return np.array([np.linalg.norm(sx, ord=order) for sx in scaled_X])

  def compute_std_slack(self, X1, X2):
    """ kernel.py:192-196 """
    k_12 = np.array([float(self.evaluate(X1[i].reshape(1, -1), X2[i].reshape(1, -1)))
                     for i in range(len(X1))])
    return np.sqrt(self.hyperparams['scale'] - k_12)


class SEKernel(_EuclideanDeviceKernel):
  """ Squared exponential kernel (kernel.py:132-222). """

  def __init__(self, dim, scale=None, dim_bandwidths=None):
    super(SEKernel, self).__init__()
    self.dim = dim
    self.set_se_hyperparams(scale, dim_bandwidths)

  def is_guaranteed_psd(self):
    return True

  def set_dim_bandwidths(self, dim_bandwidths):
    if dim_bandwidths is not None:
      if len(dim_bandwidths) != self.dim:
        raise ValueError('Dimension of dim_bandwidths should be the same as dimension.')
      dim_bandwidths = np.array(dim_bandwidths).T
    self.add_hyperparams(dim_bandwidths=dim_bandwidths)

  def set_single_bandwidth(self, bandwidth):
    dim_bandwidths = None if bandwidth is None else [bandwidth] * self.dim
    self.set_dim_bandwidths(dim_bandwidths)

  def set_scale(self, scale):
    self.add_hyperparams(scale=scale)

  def set_se_hyperparams(self, scale, dim_bandwidths):
    self.set_scale(scale)
    if hasattr(dim_bandwidths, '__len__'):
      self.set_dim_bandwidths(dim_bandwidths)
    else:
      self.set_single_bandwidth(dim_bandwidths)

  def to_spec(self, in_dim=None):
    return KernelSpec('se', self.dim, self.hyperparams['scale'],
                      np.ravel(np.asarray(self.hyperparams['dim_bandwidths'], dtype=float)))

  def __str__(self):
    scale_str, bw_str = _get_se_matern_scale_bw_strs(self)
    return 'SE: ' + scale_str + ' ' + bw_str


class MaternKernel(_EuclideanDeviceKernel):
  """ The Matern class of kernels (kernel.py:225-328), nu = p + 1/2. """

  def __init__(self, dim, nu=None, scale=None, dim_bandwidths=None):
    super(MaternKernel, self).__init__()
    self.dim = dim
    self.p = None
    self.norm_constant = None
    self.set_matern_hyperparams(nu, scale, dim_bandwidths)

  def is_guaranteed_psd(self):
    return True

  def set_matern_hyperparams(self, nu, scale, dim_bandwidths):
    """ kernel.py:242-253 """
    if nu%1 != 0.5:
      raise ValueError('Matern kernel: nu has to be p + 0.5 where p is an integer.')
    self.add_hyperparams(nu=nu)
    self.add_hyperparams(scale=scale)
    dim_bandwidths = dim_bandwidths if hasattr(dim_bandwidths, '__len__') else \
                     [dim_bandwidths] * self.dim
    dim_bandwidths = np.array(dim_bandwidths).T
    self.add_hyperparams(dim_bandwidths=dim_bandwidths)
    self.p = int(nu)
    self.norm_constant = 1.0     # 1/value(0); the library evaluates it with the reference's formula

  def to_spec(self, in_dim=None):
    bws = np.ravel(np.asarray(self.hyperparams['dim_bandwidths'], dtype=float))
    if bws.size != self.dim:
      raise ValueError('Dimension of dim_bandwidths should be the same as dimension.')
    return KernelSpec('matern', self.dim, self.hyperparams['scale'], bws,
                      nu=self.hyperparams['nu'])

  def __str__(self):
    scale_str, bw_str = _get_se_matern_scale_bw_strs(self)
    nu_str = 'nu=%0.1f'%(self.hyperparams['nu'])
    return 'Matern: ' + nu_str + ' ' + scale_str + ' ' + bw_str


def _grouped_spec(kind, kernel, groupings, in_dim):
  """ KernelSpec of a kernel made of SE / Matern sub-kernels on coordinate groups. """
  kinds, scales, nus, bws = [], [], [], []
  for kern in kernel.kernel_list:
    if isinstance(kern, SEKernel):
      kinds.append('se')
      nus.append(0.0)
    elif isinstance(kern, MaternKernel):
      kinds.append('matern')
      nus.append(kern.hyperparams['nu'])
    else:
      raise TypeError('%s on the device supports SE/Matern sub-kernels only, got %s.'
                      % (type(kernel).__name__, type(kern)))
    scales.append(kern.hyperparams['scale'])
    bws.append(np.ravel(np.asarray(kern.hyperparams['dim_bandwidths'], dtype=float)))
  groups = [[int(i) for i in grp] for grp in groupings]
  if in_dim is None:
    in_dim = max(max(max(g) for g in groups) + 1, kernel.dim)
  return KernelSpec(kind, in_dim, kernel.hyperparams['scale'], groups=groups,
                    sub_kinds=kinds, sub_scales=scales, sub_nus=nus, sub_bandwidths=bws)


class AdditiveKernel(_EuclideanDeviceKernel):
  """ Additive kernel on Euclidean spaces with non-overlapping groups (kernel.py:461-501). The
      sub-kernels must be SEKernel / MaternKernel objects. """

  def __init__(self, scale, kernel_list, groupings):
    if len(kernel_list) != len(groupings):
      raise ValueError("number of kernels do not correspond to number of groups.")
    super(AdditiveKernel, self).__init__()
    self.kernel_list = kernel_list
    self.groupings = groupings
    self.add_hyperparams(scale=scale)
    self.dim = sum([kern.dim for kern in self.kernel_list])

  def is_guaranteed_psd(self):
    return all([kern.is_guaranteed_psd() for kern in self.kernel_list])

  def get_scaled_repr(self, X):
    raise NotImplementedError('Not defined for additive kernels.')

  def has_device_spec(self):
    return all(isinstance(k, (SEKernel, MaternKernel)) for k in self.kernel_list)

  def _host_compose(self, X1, X2):
    """ kernel.py:484-494 with each factor evaluated by its own class """
    result = np.zeros((X1.shape[0], X2.shape[0]))
    for kern, group in zip(self.kernel_list, self.groupings):
      result += kern(X1[:, group], X2[:, group])
    return self.hyperparams['scale'] * result

  def to_spec(self, in_dim=None):
    return _grouped_spec('additive', self, self.groupings, in_dim)

  def __str__(self):
    kernels_str_list = ['%s(%s)'%(grp, kern) for (grp, kern) in
                        zip(self.groupings, self.kernel_list)]
    kernels_str = ', '.join(kernels_str_list)
    return 'ADD scale=%0.2f, '%(self.hyperparams['scale']) + kernels_str


class CoordinateProductKernel(_EuclideanDeviceKernel):
  """ Coordinate-wise product kernel scale * prod_i k_i(X[:, coordinate_list[i]])
      (kernel.py:541-591); the kernel of a Euclidean multi-fidelity GP (fidelity x domain).
      The sub-kernels must be SEKernel / MaternKernel objects. """

  def __init__(self, dim, scale, kernel_list=None, coordinate_list=None):
    super(CoordinateProductKernel, self).__init__()
    self.dim = dim
    self.add_hyperparams(scale=scale)
    self.kernel_list = kernel_list
    self.coordinate_list = coordinate_list

  def set_kernel_list(self, kernel_list):
    self.kernel_list = kernel_list

  def is_guaranteed_psd(self):
    return all([kern.is_guaranteed_psd() for kern in self.kernel_list])

  def set_new_kernel(self, kernel_idx, new_kernel):
    self.kernel_list[kernel_idx] = new_kernel

  def set_kernel_hyperparams(self, kernel_idx, **kwargs):
    self.kernel_list[kernel_idx].set_hyperparams(**kwargs)

  def get_scaled_repr(self, X):
    raise NotImplementedError('Not defined for product kernels.')

  def has_device_spec(self):
    return all(isinstance(k, (SEKernel, MaternKernel)) for k in self.kernel_list)

  def _host_compose(self, X1, X2):
    """ kernel.py:578-589 with each factor evaluated by its own class """
    K = self.hyperparams['scale'] * np.ones((X1.shape[0], X2.shape[0]))
    for kern, coords in zip(self.kernel_list, self.coordinate_list):
      K *= kern(X1[:, coords], X2[:, coords])
    return K

  def to_spec(self, in_dim=None):
    if len(self.kernel_list) != len(self.coordinate_list):
      raise ValueError("number of kernels do not correspond to number of coordinate groups.")
    return _grouped_spec('product', self, self.coordinate_list, in_dim if in_dim is not None else self.dim)

  def __str__(self):
    kernels_str_list = ['%s(%s)'%(grp, kern) for (grp, kern) in
                        zip(self.coordinate_list, self.kernel_list)]
    kernels_str = ', '.join(kernels_str_list)
    return 'CoordProd scale=%0.2f, '%(self.hyperparams['scale']) + kernels_str
