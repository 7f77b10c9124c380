"""Rebuild the kernel of a golden GP case for (a) the NumPy oracle and (b) dragonfly_amd."""
import numpy as np


def _case_kind(g):
  if 'kern_groups' in g:
    return 'additive'
  return 'matern' if 'kern_nu' in g else 'se'


def _additive_parts(g):
  groups = [[int(i) for i in row if i >= 0] for row in g['kern_groups']]
  bws = [np.asarray(b[:len(grp)], dtype=float) for b, grp in zip(g['kern_sub_bws'], groups)]
  kinds = ['se' if int(k) == 0 else 'matern' for k in g['kern_sub_kinds']]
  return groups, bws, kinds


def oracle_kernel(g):
  from oracle import ref_numpy as O
  kind = _case_kind(g)
  d = g['X'].shape[1]
  scale = float(g['kern_scale'])
  if kind == 'additive':
    groups, bws, kinds = _additive_parts(g)
    subs = [O.KernelSpec(k, len(grp), 1.0, bw, nu=(2.5 if k == 'matern' else None))
            for k, grp, bw in zip(kinds, groups, bws)]
    return O.KernelSpec('additive', d, scale, groups=groups, subs=subs)
  if kind == 'se':
    return O.KernelSpec('se', d, scale, g['kern_bw'])
  return O.KernelSpec('matern', d, scale, g['kern_bw'], nu=float(g['kern_nu']))


def device_kernel(g):
  from dragonfly_amd import kernel as K
  kind = _case_kind(g)
  d = g['X'].shape[1]
  scale = float(g['kern_scale'])
  if kind == 'additive':
    groups, bws, kinds = _additive_parts(g)
    subs = [K.SEKernel(len(grp), 1.0, bw) if k == 'se' else K.MaternKernel(len(grp), 2.5, 1.0, bw)
            for k, grp, bw in zip(kinds, groups, bws)]
    return K.AdditiveKernel(scale, subs, groups)
  if kind == 'se':
    return K.SEKernel(d, scale, g['kern_bw'])
  return K.MaternKernel(d, float(g['kern_nu']), scale, g['kern_bw'])
