"""Euclidean kernels evaluated on the MI355X: SE, Matern, Polynomial, Exponential-decay, Additive
and coordinate-wise Product.

Host-side counterpart of dragonfly/gp/kernel.py: the class names, constructor arguments, the
`hyperparams` dictionary, the printed form and the error behaviour are the reference's (its lines
are quoted in the docstrings), so code written against the reference's kernels runs unchanged.
The Gram / cross matrices themselves are produced by libdfhip.so (csrc/kernmat.hip) -- there is no
NumPy evaluation path here, except for grouped kernels with a factor the device does not know,
which are composed from their factors' own evaluations.
"""
import numpy as np

from .engine import KernelSpec, get_engine


def _as_2d_array(X):
  """ The reference passes lists of 1-D arrays (gp_core.py:129-130) or 2-D arrays. """
  X = np.asarray(X, dtype=np.float64)
  if X.ndim == 1:
    X = X.reshape((len(X), -1)) if X.size else X.reshape((0, 0))
  return np.ascontiguousarray(X)


def _scale_and_bandwidth_text(kern):
  """ The two fragments the reference prints for SE / Matern kernels (kernel.py:48-57): the mean
      bandwidth above six dimensions, the individual ones otherwise. """
  bws = kern.hyperparams['dim_bandwidths']
  if kern.dim <= 6:
    bw_text = 'bws:[%s]' % (' '.join('%0.2f' % (b) for b in bws))
  else:
    bw_text = 'avg-bw: %0.4f' % (bws.mean())
  return 'sc:%0.4f' % (kern.hyperparams['scale']), bw_text


def _bandwidth_vector(dim, bandwidths, allow_scalar):
  """ Per-dimension bandwidths as the reference stores them (kernel.py:151-160, 247-250): a
      vector of length dim (checked), or one value repeated; None stays None. """
  if bandwidths is None:
    return None
  if hasattr(bandwidths, '__len__'):
    if len(bandwidths) != dim:
      raise ValueError('Dimension of dim_bandwidths should be the same as dimension.')
    return np.array(bandwidths).T
  if not allow_scalar:
    raise ValueError('Dimension of dim_bandwidths should be the same as dimension.')
  return np.array([bandwidths] * dim).T


class Kernel(object):
  """ Base class (kernel.py:60-129): a callable returning the n1 x n2 kernel matrix, with its
      parameters in the dictionary `hyperparams`. """

  def __init__(self):
    self.hyperparams = dict()

  def is_guaranteed_psd(self):
    raise NotImplementedError('Implement in a child class.')

  def evaluate(self, X1, X2=None):
    """ kernel.py:76-83; empty inputs give an empty matrix without touching the device. """
    if X2 is None:
      X2 = X1
    n1, n2 = len(X1), len(X2)
    if n1 == 0 or n2 == 0:
      return np.zeros((n1, n2))
    return self._child_evaluate(X1, X2)

  __call__ = evaluate

  def _child_evaluate(self, X1, X2):
    raise NotImplementedError('Implement in a child class.')

  def set_hyperparams(self, **kwargs):
    """ Replaces the whole dictionary (kernel.py:110-112). """
    self.hyperparams = dict(kwargs)

  def add_hyperparams(self, **kwargs):
    """ Adds / overwrites entries (kernel.py:114-117). """
    self.hyperparams.update(kwargs)

  def to_spec(self, in_dim=None):
    """ Description handed to the C-ABI (struct dfh_kernel_desc). in_dim: number of columns of
        the inputs (only a grouped kernel cannot tell it from its own parameters). """
    raise NotImplementedError('Implement in a child class.')

  def __str__(self):
    return '%s:: %s'%(type(self), str(self.hyperparams))


class _EuclideanDeviceKernel(Kernel):
  """ Shared device evaluation of the Euclidean kernels. """

  def has_device_spec(self):
    """ False for a grouped kernel with a factor the device does not evaluate (e.g. the reference's
        PolyKernel inside a product): it is then composed on the host from its factors' own
        evaluations, and GPs using it run in host-kernel mode (gp_core.GP._generic). """
    return True

  def _child_evaluate(self, X1, X2):
    A = _as_2d_array(X1)
    B = None if X2 is X1 else _as_2d_array(X2)
    if B is not None and A.shape[1] != B.shape[1]:
      raise ValueError('Second dimension of X1 and X2 should be equal.')   # general_utils.py:64
    if not self.has_device_spec():
      return self._host_compose(A, A if B is None else B)
    return get_engine().kernel_matrix(self.to_spec(in_dim=A.shape[1]), A, B)

  # helpers of the reference's SE kernel that other parts of Dragonfly call (kernel.py:179-200)
  def get_scaled_repr(self, X):
    return X/self.hyperparams['dim_bandwidths']

  def get_effective_norm(self, X, order=None, is_single=True):
    scaled = self.get_scaled_repr(X)
    if is_single:
      return np.linalg.norm(scaled, ord=order)
    return np.array([np.linalg.norm(row, ord=order) for row in scaled])

  def compute_std_slack(self, X1, X2):
    pairwise = [float(self.evaluate(x1.reshape(1, -1), x2.reshape(1, -1))[0, 0]) for x1, x2 in zip(X1, X2)]
    return np.sqrt(self.hyperparams['scale'] - np.array(pairwise))

  def change_smoothness(self, factor):
    self.hyperparams['dim_bandwidths'] *= factor


class SEKernel(_EuclideanDeviceKernel):
  """ Squared exponential kernel scale * exp(-|x/bw - y/bw|^2 / 2) (kernel.py:132-222). """

  def __init__(self, dim, scale=None, dim_bandwidths=None):
    super(SEKernel, self).__init__()
    self.dim = dim
    self.set_se_hyperparams(scale, dim_bandwidths)

  def is_guaranteed_psd(self):
    return True

  def set_scale(self, scale):
    self.hyperparams['scale'] = scale

  def set_dim_bandwidths(self, dim_bandwidths):
    self.hyperparams['dim_bandwidths'] = _bandwidth_vector(self.dim, dim_bandwidths, allow_scalar=False)

  def set_single_bandwidth(self, bandwidth):
    self.hyperparams['dim_bandwidths'] = _bandwidth_vector(self.dim, bandwidth, allow_scalar=True)

  def set_se_hyperparams(self, scale, dim_bandwidths):
    """ kernel.py:167-173: a vector is taken per dimension, anything else as one common value. """
    self.set_scale(scale)
    self.hyperparams['dim_bandwidths'] = _bandwidth_vector(self.dim, dim_bandwidths, allow_scalar=True)

  def to_spec(self, in_dim=None):
    return KernelSpec('se', self.dim, self.hyperparams['scale'],
                      np.ravel(np.asarray(self.hyperparams['dim_bandwidths'], dtype=float)))

  def __str__(self):
    return 'SE: %s %s' % _scale_and_bandwidth_text(self)


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
    """ kernel.py:242-253; the half-integer check and its message are the reference's. """
    if nu%1 != 0.5:
      raise ValueError('Matern kernel: nu has to be p + 0.5 where p is an integer.')
    if hasattr(dim_bandwidths, '__len__'):
      bws = np.array(dim_bandwidths).T            # length is checked when the kernel is evaluated
    else:
      bws = np.array([dim_bandwidths] * self.dim).T
    self.hyperparams.update(nu=nu, scale=scale, dim_bandwidths=bws)
    self.p = int(nu)
    self.norm_constant = 1.0     # 1/value(0); the library evaluates it with the reference's formula

  def to_spec(self, in_dim=None):
    bws = np.ravel(np.asarray(self.hyperparams['dim_bandwidths'], dtype=float))
    if bws.size != self.dim:
      raise ValueError('Dimension of dim_bandwidths should be the same as dimension.')
    return KernelSpec('matern', self.dim, self.hyperparams['scale'], bws,
                      nu=self.hyperparams['nu'])

  def __str__(self):
    return 'Matern: nu=%0.1f %s %s' % ((self.hyperparams['nu'],) + _scale_and_bandwidth_text(self))


class PolyKernel(_EuclideanDeviceKernel):
  """ The polynomial kernel scale * ((x*s).(y*s) + 1)^order (kernel.py:331-395).  Not stationary:
      the prior variance k(x, x) depends on x (the library evaluates it per point). """

  def __init__(self, dim, order, scale, dim_scalings=None):
    super(PolyKernel, self).__init__()
    self.dim = dim
    self.set_poly_hyperparams(order, scale, dim_scalings)

  def is_guaranteed_psd(self):
    return True

  def set_order(self, order):
    self.add_hyperparams(order=order)

  def set_scale(self, scale):
    self.add_hyperparams(scale=scale)

  def set_dim_scalings(self, dim_scalings):
    """ kernel.py:356-364 """
    if dim_scalings is not None:
      if len(dim_scalings) != self.dim:
        raise ValueError('Dimension of dim_scalings should be dim.')
      dim_scalings = np.array(dim_scalings)
    self.add_hyperparams(dim_scalings=dim_scalings)

  def set_single_scaling(self, scaling):
    self.set_dim_scalings(None if scaling is None else [scaling] * self.dim)

  def set_poly_hyperparams(self, order, scale, dim_scalings):
    """ kernel.py:373-380 """
    self.set_order(order)
    self.set_scale(scale)
    if hasattr(dim_scalings, '__len__'):
      self.set_dim_scalings(dim_scalings)
    else:
      self.set_single_scaling(dim_scalings)

  def get_scaled_repr(self, X):
    return X * self.hyperparams['dim_scalings']

  def to_spec(self, in_dim=None):
    return _poly_spec(self)

  def __str__(self):
    return 'Poly: d=%d, scale=%0.2f, %s'%(self.hyperparams['order'], self.hyperparams['scale'],
                                          ','.join('%0.2f'%(elem) for elem in self.hyperparams['dim_scalings']))


class ExpDecayKernel(_EuclideanDeviceKernel):
  """ The kernel for exponentially decaying functions of Freeze-Thaw Bayesian optimisation,
      scale * prod_d (1 + x_d + y_d)^(-powers_d) + offset (kernel.py:398-437): the fidelity kernel of
      the reference's multi-fidelity GPs (euclidean_gp.py:881-887).  Not stationary. """

  def __init__(self, dim, scale=None, offset=None, powers=None):
    super(ExpDecayKernel, self).__init__()
    self.dim = dim
    if not hasattr(powers, '__iter__'):
      powers = [powers] * dim
    self.set_hyperparams(scale=scale, offset=offset, powers=powers)

  def is_guaranteed_psd(self):
    return True

  def get_scaled_repr(self, X):
    raise NotImplementedError('Not defined for the exponential-decay kernel.')

  def to_spec(self, in_dim=None):
    return _expdecay_spec(self)

  def __str__(self):
    return 'ExpDec: sc=%0.3f, offset=%0.3f, pow=%s'%(self.hyperparams['scale'], self.hyperparams['offset'],
        '[' + ', '.join('%0.3f'%(b) for b in np.ravel(self.hyperparams['powers'])) + ']')


def _factor_kind(kern):
  """ 'se' | 'matern' | 'poly' | 'expdecay' | None for a factor of a grouped kernel.  Decided by the
      class NAME and the hyper-parameters it carries, so that the reference's own PolyKernel /
      ExpDecayKernel objects (dragonfly/gp/kernel.py), which `install()` may leave in place inside a
      product kernel, reach the device as well. """
  hps = getattr(kern, 'hyperparams', None)
  if not isinstance(hps, dict):
    return None
  name = type(kern).__name__
  if isinstance(kern, SEKernel):
    return 'se'
  if isinstance(kern, MaternKernel):
    return 'matern'
  if name == 'PolyKernel' and all(k in hps for k in ('order', 'scale', 'dim_scalings')):
    order = hps['order']
    if hps['dim_scalings'] is not None and float(order) == int(order) and 0 <= int(order) <= 64:
      return 'poly'
  if name == 'ExpDecayKernel' and all(k in hps for k in ('scale', 'offset', 'powers')):
    if len(np.ravel(hps['powers'])) <= 8:
      return 'expdecay'
  return None


def _factor_fields(kern, kind):
  """ (scale, nu field, per-column field) of one factor, as struct dfh_kernel_desc carries them """
  hps = kern.hyperparams
  if kind == 'se':
    return hps['scale'], 0.0, np.ravel(np.asarray(hps['dim_bandwidths'], dtype=float))
  if kind == 'matern':
    return hps['scale'], hps['nu'], np.ravel(np.asarray(hps['dim_bandwidths'], dtype=float))
  if kind == 'poly':
    return hps['scale'], float(hps['order']), np.ravel(np.asarray(hps['dim_scalings'], dtype=float))
  return hps['scale'], float(hps['offset']), np.ravel(np.asarray(hps['powers'], dtype=float))


def _poly_spec(kern):
  scale, order, scalings = _factor_fields(kern, 'poly')
  if scalings.size != kern.dim:
    raise ValueError('Dimension of dim_scalings should be dim.')
  return KernelSpec('poly', kern.dim, scale, scalings, nu=order)


def _expdecay_spec(kern):
  scale, offset, powers = _factor_fields(kern, 'expdecay')
  if powers.size != kern.dim:
    raise ValueError('Dimension of powers should be dim.')
  return KernelSpec('expdecay', kern.dim, scale, powers, nu=offset)


class _GroupedKernel(_EuclideanDeviceKernel):
  """ A kernel assembled from SE / Matern factors on groups of coordinates: the additive kernel
      (sum) and the coordinate-wise product kernel.  `_groups()` gives the coordinate lists. """
  _kind = None
  _label = None

  def _groups(self):
    raise NotImplementedError

  def is_guaranteed_psd(self):
    return all(kern.is_guaranteed_psd() for kern in self.kernel_list)

  def get_scaled_repr(self, X):
    raise NotImplementedError('Not defined for grouped kernels.')

  _factor_kinds = ('se', 'matern', 'poly')  # what the device evaluates inside this grouped kernel (poly: euclidean_gp.py:870-879)

  def has_device_spec(self):
    return all(_factor_kind(kern) in self._factor_kinds for kern in self.kernel_list)

  def to_spec(self, in_dim=None):
    groups = [[int(c) for c in grp] for grp in self._groups()]
    if len(groups) != len(self.kernel_list):
      raise ValueError("number of kernels do not correspond to number of groups.")
    kinds, scales, nus, bws = [], [], [], []
    for kern in self.kernel_list:
      kind = _factor_kind(kern)
      if kind not in self._factor_kinds:
        raise TypeError('%s on the device supports %s sub-kernels only, got %s.'
                        % (type(self).__name__, '/'.join(self._factor_kinds), type(kern)))
      scale, nu, per_col = _factor_fields(kern, kind)
      kinds.append(kind)
      nus.append(nu)
      scales.append(scale)
      bws.append(per_col)
    if in_dim is None:
      in_dim = max(1 + max(max(grp) for grp in groups), self.dim)
    return KernelSpec(self._kind, in_dim, self.hyperparams['scale'], groups=groups, sub_kinds=kinds,
                      sub_scales=scales, sub_nus=nus, sub_bandwidths=bws)

  def __str__(self):
    parts = ', '.join('%s(%s)' % (grp, kern) for grp, kern in zip(self._groups(), self.kernel_list))
    return '%s scale=%0.2f, %s' % (self._label, self.hyperparams['scale'], parts)


class AdditiveKernel(_GroupedKernel):
  """ Additive kernel scale * sum_g k_g(X[:, group g]) with non-overlapping groups
      (kernel.py:461-501). """
  _kind = 'additive'
  _label = 'ADD'

  def __init__(self, scale, kernel_list, groupings):
    if len(kernel_list) != len(groupings):
      raise ValueError("number of kernels do not correspond to number of groups.")
    super(AdditiveKernel, self).__init__()
    self.hyperparams['scale'] = scale
    self.kernel_list, self.groupings = kernel_list, groupings
    self.dim = sum(kern.dim for kern in kernel_list)

  def _groups(self):
    return self.groupings

  def _host_compose(self, X1, X2):
    """ kernel.py:484-494 with each factor evaluated by its own class """
    total = np.zeros((X1.shape[0], X2.shape[0]))
    for kern, grp in zip(self.kernel_list, self.groupings):
      total += kern(X1[:, grp], X2[:, grp])
    return self.hyperparams['scale'] * total


class CoordinateProductKernel(_GroupedKernel):
  """ Coordinate-wise product kernel scale * prod_i k_i(X[:, coordinate_list[i]])
      (kernel.py:541-591); the kernel of a Euclidean multi-fidelity GP (fidelity x domain). """
  _kind = 'product'
  _label = 'CoordProd'
  _factor_kinds = ('se', 'matern', 'poly', 'expdecay')

  def __init__(self, dim, scale, kernel_list=None, coordinate_list=None):
    super(CoordinateProductKernel, self).__init__()
    self.dim = dim
    self.hyperparams['scale'] = scale
    self.kernel_list, self.coordinate_list = kernel_list, coordinate_list

  def _groups(self):
    return self.coordinate_list

  def set_kernel_list(self, kernel_list):
    self.kernel_list = kernel_list

  def set_new_kernel(self, kernel_idx, new_kernel):
    self.kernel_list[kernel_idx] = new_kernel

  def set_kernel_hyperparams(self, kernel_idx, **kwargs):
    self.kernel_list[kernel_idx].set_hyperparams(**kwargs)

  @staticmethod
  def _additive_factor(kern):
    """ An AdditiveKernel among the product's kernels (ours or, left in place by `install()`, the
        reference's): the multi-fidelity GP with an additive domain model, euclidean_gp.py:696-707 """
    return type(kern).__name__ == 'AdditiveKernel' and hasattr(kern, 'kernel_list') and hasattr(kern, 'groupings') \
           and isinstance(getattr(kern, 'hyperparams', None), dict) and 'scale' in kern.hyperparams

  def has_device_spec(self):
    for kern in self.kernel_list:
      if self._additive_factor(kern):
        if not all(_factor_kind(k) in AdditiveKernel._factor_kinds for k in kern.kernel_list):
          return False
      elif _factor_kind(kern) not in self._factor_kinds:
        return False
    return True

  def to_spec(self, in_dim=None):
    in_dim = self.dim if in_dim is None else in_dim
    if not any(self._additive_factor(kern) for kern in self.kernel_list):
      return super(CoordinateProductKernel, self).to_spec(in_dim)
    # additive factors: their groups are listed with the product's own, in absolute coordinates
    if len(self.coordinate_list) != len(self.kernel_list):
      raise ValueError("number of kernels do not correspond to number of groups.")
    groups, kinds, scales, nus, bws, gfac, fsum, fscale = [], [], [], [], [], [], [], []
    for f, (kern, coords) in enumerate(zip(self.kernel_list, self.coordinate_list)):
      coords = [int(c) for c in coords]
      if self._additive_factor(kern):
        if len(kern.kernel_list) != len(kern.groupings):
          raise ValueError("number of kernels do not correspond to number of groups.")
        members = [(k, [coords[int(c)] for c in grp]) for k, grp in zip(kern.kernel_list, kern.groupings)]
        allowed = AdditiveKernel._factor_kinds
        fsum.append(True)
        fscale.append(kern.hyperparams['scale'])
      else:
        members = [(kern, coords)]
        allowed = self._factor_kinds
        fsum.append(False)
        fscale.append(1.0)
      for k, cols in members:
        kind = _factor_kind(k)
        if kind not in allowed:
          raise TypeError('%s on the device supports %s sub-kernels only, got %s.'
                          % (type(self).__name__, '/'.join(allowed), type(k)))
        scale, nu, per_col = _factor_fields(k, kind)
        groups.append(cols); kinds.append(kind); scales.append(scale); nus.append(nu); bws.append(per_col)
        gfac.append(f)
    return KernelSpec('product', in_dim, self.hyperparams['scale'], groups=groups, sub_kinds=kinds,
                      sub_scales=scales, sub_nus=nus, sub_bandwidths=bws, group_factors=gfac, factor_sums=fsum,
                      factor_scales=fscale)

  def _host_compose(self, X1, X2):
    """ kernel.py:578-589 with each factor evaluated by its own class """
    K = self.hyperparams['scale'] * np.ones((X1.shape[0], X2.shape[0]))
    for kern, coords in zip(self.kernel_list, self.coordinate_list):
      K *= kern(X1[:, coords], X2[:, coords])
    return K
