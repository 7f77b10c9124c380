"""One description of how a kernel's hyper-parameters sit in the fitters' vectors.

The reference's fitters tune two flat vectors -- continuous hyper-parameters in log space
(`gp_cts_hps`) and discrete ones (`gp_dscr_hps`) -- and its kernel factory peels a kernel's share
off the front of each (dragonfly/gp/euclidean_gp.py:808-900; the fitters lay the vectors out
accordingly, :215-276 and :433-619).  Here that contract is data: per kernel type, the ordered
fields a kernel consumes, each with

    arg     constructor argument of the kernel class it feeds,
    source  'cts' (exponentiated log values off the continuous vector) or 'dscr' (one value off the
            discrete vector -- unless the fitter options pin it, in which case nothing is consumed),
    shape   'per_dim'  one value per input dimension, or a single shared value when the fitter ties
                       them (use_same_bandwidth); a group of an additive kernel takes its own
                       coordinates' values,
            'scalar'   one value,
            'vector'   one value per input dimension, handed to every group whole.

`consume_kernel_hps` (building kernels: the factory below it, both fitters) and `describe_kernel_hps`
(sizing the hyper-parameter boxes and the `param_order` bookkeeping the posterior sampler walks)
read the same table, so the two cannot drift apart.
"""
from collections import namedtuple

import numpy as np

HPField = namedtuple('HPField', ['arg', 'source', 'shape', 'pinned_by', 'param_name'])

KERNEL_HP_LAYOUT = {
  'se': ('SEKernel', [HPField('dim_bandwidths', 'cts', 'per_dim', None, 'dim_bandwidths')]),
  'matern': ('MaternKernel', [HPField('dim_bandwidths', 'cts', 'per_dim', None, 'dim_bandwidths'),
                              HPField('nu', 'dscr', 'scalar', 'nu', 'nu')]),
  'poly': ('PolyKernel', [HPField('dim_scalings', 'cts', 'per_dim', None, 'dim_bandwidths'),
                          HPField('order', 'dscr', 'scalar', 'order', 'order')]),
  'expdecay': ('ExpDecayKernel', [HPField('offset', 'cts', 'scalar', None, 'expdecay_offset'),
                                  HPField('powers', 'cts', 'vector', None, 'expdecay_powers')]),
}


def _is_pinned(field, kernel_hyperparams):
  """ a discrete hyper-parameter the options fix (a non-negative value) is not in the vector """
  return field.pinned_by is not None and field.pinned_by in kernel_hyperparams and \
         kernel_hyperparams[field.pinned_by] >= 0


def consume_kernel_hps(kernel_type, dim, kernel_hyperparams, cts_hps, dscr_hps, tied):
  """ Takes the kernel's fields off the fronts of the two vectors.  Returns ({arg: value}, the
      kernel class name, the left-over continuous and discrete vectors); per_dim values come back as
      a length-dim sequence (a tied value repeated). """
  if kernel_type not in KERNEL_HP_LAYOUT:
    raise Exception('Unknown kernel type %s!' % (kernel_type))
  cls_name, fields = KERNEL_HP_LAYOUT[kernel_type]
  values = {}
  for field in fields:
    if field.source == 'cts':
      if field.shape == 'per_dim' and tied:
        values[field.arg] = [np.exp(cts_hps[0])] * dim
        cts_hps = cts_hps[1:]
      elif field.shape == 'scalar':
        values[field.arg] = np.exp(cts_hps[0])
        cts_hps = cts_hps[1:]
      else:
        values[field.arg] = np.exp(cts_hps[0:dim])
        cts_hps = cts_hps[dim:]
    elif _is_pinned(field, kernel_hyperparams):
      values[field.arg] = kernel_hyperparams[field.pinned_by]
    else:
      values[field.arg] = dscr_hps[0]
      dscr_hps = dscr_hps[1:]
  return values, cls_name, cts_hps, dscr_hps


def group_kernel_args(kernel_type, values, group):
  """ the constructor arguments of one group's kernel: per_dim fields restricted to the group """
  _, fields = KERNEL_HP_LAYOUT[kernel_type]
  out = {}
  for field in fields:
    v = values[field.arg]
    out[field.arg] = [v[idx] for idx in group] if field.shape == 'per_dim' else v
  return out


def describe_kernel_hps(kernel_type, dim, kernel_hyperparams, tied):
  """ [(param_name, 'cts' | 'dscr', count)] in vector order: what a fitter has to provide boxes /
      value lists for, and the names the posterior sampler's `param_order` carries. """
  _, fields = KERNEL_HP_LAYOUT[kernel_type]
  out = []
  for field in fields:
    if field.source == 'dscr':
      if not _is_pinned(field, kernel_hyperparams):
        out.append((field.param_name, 'dscr', 1))
    elif field.shape == 'per_dim' and tied:
      out.append(('same_' + field.param_name, 'cts', 1))
    elif field.shape == 'scalar':
      out.append((field.param_name, 'cts', 1))
    else:
      out.append((field.param_name, 'cts', dim))
  return out
