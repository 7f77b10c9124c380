"""Acquisition functions for Bayesian optimisation on the MI355X -- host-side mirror of
dragonfly/opt/gpb_acquisitions.py (UCB :215, PI :230, EI :251, TTEI :282, TS :119, add-UCB :139;
hallucination wrappers :43-87; synchronous batches :90-115; namespaces asy / syn / seq :443-471).

Same callables, same `anc_data` fields (gp_bandit.py:462-484): `asy.<acq>(gp, anc_data) -> point`,
`syn.<acq>(num_workers, gps, anc_datas) -> [points]`.

With acq_opt_method == 'rand' on a Euclidean domain -- the batched path, oper_utils.py:59-80 --
the whole evaluation is ONE fused device call: candidates are drawn from the global np.random
state exactly as the reference draws them, then posterior mean / std, the acquisition formula
and the arg-max (first NaN, else first maximum) run in libdfhip.so and only the winning index
comes back.  Other acq_opt_method values are per-point serial tree searches in the reference
(DIRECT / PDOO, out of scope): they are served through Dragonfly's own maximisers when this
package is installed under Dragonfly (dragonfly_amd.install), each callback hitting GP.eval on
the device.
"""
import os
from argparse import Namespace
from copy import copy

import numpy as np

from . import gaplog
from .general_utils import map_to_bounds
from .kernel import AdditiveKernel, _as_2d_array
from .doo import pdoo_maximise_batched
from .oper_utils import random_maximise

# Candidates of the fused 'rand' path are generated in HBM (bit-identical to the host draw, same
# generator state afterwards); DFH_HOST_CANDIDATES=1 keeps the draw on the host.
DEVICE_CANDIDATES = os.environ.get('DFH_HOST_CANDIDATES', '0') != '1'

# Open leaves whose halves are prefetched with every cache miss of the batched PDOO (0: one point
# per device call, the reference's access pattern).
PDOO_FRONTIER = int(os.environ.get('DFH_PDOO_FRONTIER', '32'))

# A maximiser for non-'rand' methods; dragonfly_amd.install points this at
# dragonfly.exd.exd_utils.maximise_with_method.
external_maximise_with_method = None


def _candidates(anc_data, max_evals=None):
  """ random_sample's draw (oper_utils.py:61-62): map_to_bounds(np.random.random((m, d)), b). """
  bounds = np.asarray(anc_data.domain.bounds, dtype=np.float64)
  m = int(anc_data.max_evals if max_evals is None else max_evals)
  return map_to_bounds(np.random.random((m, len(bounds))), bounds)


def _device_candidates(gp, anc_data):
  """ The same draw generated in HBM (Engine.random_candidates): the global np.random state is
      continued bit for bit and left where the host draw would have left it, so the candidates
      are the reference's -- without the host generation and the m x d copy.  Returns the
      DeviceArray and what acq_argmax / thompson need to add the prior mean: a constant when the
      mean function is the fitter's constant, else its values on a host copy of the candidates. """
  bounds = np.asarray(anc_data.domain.bounds, dtype=np.float64)
  cands = gp.device_gp.engine.random_candidates(int(anc_data.max_evals), len(bounds), bounds=bounds)
  const = getattr(gp.mean_func, 'constant_value', None)
  if const is not None:
    return cands, dict(mean_const=float(const))
  return cands, dict(mean_vals=gp.mean_func(cands.download()))


def _is_rand_euclidean(anc_data):
  return anc_data.domain.get_type() == 'euclidean' and \
         str(anc_data.acq_opt_method).lower().startswith('rand')


def _is_device_gp(gp):
  """ A fitted mirror GP with a device kernel.  Anything else offering eval() -- e.g. the
      Namespace BOCA hands the acquisitions for the GP restricted to the target fidelity
      (gpb_acquisitions.py:384-395) -- is served through its eval. """
  from .gp_core import GP
  return isinstance(gp, GP) and gp.num_tr_data > 0 and not gp._generic    # pylint: disable=protected-access


def _can_fuse(gp, anc_data):
  """ The fused candidates -> posterior -> acquisition -> arg-max call needs a device kernel; GPs
      whose kernel the host evaluates take the reference's closure route (batched gp.eval). """
  return _is_rand_euclidean(anc_data) and _is_device_gp(gp) and not getattr(anc_data, 'is_mf', False)


def _fortran_direct_available():
  """ True when a Dragonfly with its compiled DIRECT is importable (oper_utils.py:21-25). """
  if external_maximise_with_method is None:      # not installed under a Dragonfly
    return False
  try:
    from dragonfly.utils import oper_utils as ref_oper_utils
    return ref_oper_utils.direct_ft_wrap is not None
  except Exception:     # pylint: disable=broad-except
    return False


def maximise_acquisition(acq_fn, anc_data, *args, **kwargs):
  """ gpb_acquisitions.py:23-40 for acquisition callables over rows of points ([m x d] -> [m]).
      'rand': one vectorised call on the random candidates.  'pdoo', and 'direct' where the
      reference itself falls back to PDOO because its Fortran DIRECT is not built
      (oper_utils.py:130-133): the tree search of dragonfly_amd.doo, which visits the boxes the
      reference visits but fetches the values a frontier per device call.  Anything else goes to
      Dragonfly's own maximiser when installed under it. """
  acq_opt_method = str(anc_data.acq_opt_method).lower()
  if anc_data.domain.get_type() != 'euclidean':
    raise NotImplementedError('dragonfly_amd acquisitions handle Euclidean domains.')
  if acq_opt_method.startswith('rand'):
    kwargs.pop('vectorised', None)
    _, opt_pt, _ = random_maximise(acq_fn, anc_data.domain.bounds, anc_data.max_evals)
    return opt_pt
  if acq_opt_method.startswith('pdoo') or \
     (acq_opt_method.startswith('direct') and not _fortran_direct_available()):
    frontier = int(getattr(anc_data, 'pdoo_frontier', PDOO_FRONTIER))
    _, opt_pt, _ = pdoo_maximise_batched(acq_fn, anc_data.domain.bounds, anc_data.max_evals,
                                         frontier=frontier, depth=2 if frontier > 0 else 0)
    return opt_pt
  if external_maximise_with_method is None:
    raise NotImplementedError(
        'acq_opt_method=%s is served by Dragonfly\'s own maximiser; install under Dragonfly '
        '(dragonfly_amd.install) to use it.' % (anc_data.acq_opt_method))
  acquisition = lambda x: acq_fn(x.reshape((1, -1)))
  _, opt_pt = external_maximise_with_method(anc_data.acq_opt_method, acquisition, anc_data.domain,
                                            anc_data.max_evals, *args, **kwargs)
  return opt_pt


class _BoxDomain(object):
  """ The Euclidean sub-domain of one additive group (EuclideanDomain(domain_bounds[group_j]),
      gpb_acquisitions.py:180). """

  def __init__(self, bounds):
    self.bounds = np.asarray(bounds, dtype=np.float64)

  def get_type(self):
    return 'euclidean'

  def get_dim(self):
    return len(self.bounds)


def _in_progress(anc_data):
  """ What the acquisitions hallucinate (gpb_acquisitions.py:56-64): the evaluations in progress
      when handle_parallel == 'halluc' -- (fidelity, point) pairs in multi-fidelity runs, where the
      GP handed in is BOCA's view at the target fidelity -- or None. """
  if getattr(anc_data, 'handle_parallel', None) != 'halluc' or \
     len(getattr(anc_data, 'eval_points_in_progress', [])) == 0:
    return None
  if getattr(anc_data, 'is_mf', False):
    return anc_data.eval_fidel_points_in_progress
  return anc_data.eval_points_in_progress


def _halluc_points(anc_data):
  """ The in-progress points as an array for the device calls (single-fidelity GPs only). """
  pts = _in_progress(anc_data)
  if pts is None or getattr(anc_data, 'is_mf', False):
    return None
  return _as_2d_array(pts)


def _get_gp_eval_for_parallel_strategy(gp, anc_data, uncert_form='std'):
  """ gpb_acquisitions.py:43-64 """
  pts = _in_progress(anc_data)
  if pts is not None:
    pts = pts if getattr(anc_data, 'is_mf', False) else _as_2d_array(pts)
    return lambda x: gp.eval_with_hallucinated_observations(x, pts, uncert_form=uncert_form)
  return lambda x: gp.eval(x, uncert_form=uncert_form)


def get_gp_sampler_for_parallel_strategy(gp, anc_data):
  """ gpb_acquisitions.py:67-87 """
  pts = _in_progress(anc_data)
  if pts is not None:
    pts = pts if getattr(anc_data, 'is_mf', False) else _as_2d_array(pts)
    return lambda x: gp.draw_samples_with_hallucinated_observations(1, x, pts).ravel()
  return lambda x: gp.draw_samples(1, x).ravel()


def _fused_argmax(gp, acq, params, anc_data):
  """ Candidates -> posterior -> acquisition -> arg-max on the device; returns the point. """
  Xh = _halluc_points(anc_data)
  if DEVICE_CANDIDATES:
    cands, mean = _device_candidates(gp, anc_data)
    if gaplog.ENABLED:
      _, idx, vals = gp.device_gp.acq_argmax(acq, cands, params=params, X_halluc=Xh, return_vals=True, **mean)
      gaplog.top2('acq_argmax', vals)
      return cands.row(idx)
    _, idx = gp.device_gp.acq_argmax(acq, cands, params=params, X_halluc=Xh, **mean)
    return cands.row(idx)
  cands = _candidates(anc_data)
  if gaplog.ENABLED:
    _, idx, vals = gp.device_gp.acq_argmax(acq, cands, params=params, mean_vals=gp.mean_func(cands), X_halluc=Xh, return_vals=True)
    gaplog.top2('acq_argmax', vals)
    return cands[idx]
  _, idx = gp.device_gp.acq_argmax(acq, cands, params=params, mean_vals=gp.mean_func(cands), X_halluc=Xh)
  return cands[idx]


def _get_syn_recommendations_from_asy(asy_acq, num_workers, list_of_gps, anc_datas):
  """ A synchronous batch is the asynchronous rule applied num_workers times in a row, every
      earlier recommendation counting as an evaluation in progress for the later ones
      (gpb_acquisitions.py:90-115); a single gp / anc_data serves all workers, lists are used
      round-robin.  The objects are shallow-copied so that the caller's anc_data keeps its own
      eval_points_in_progress. """
  def _per_worker(obj):
    seq = list(obj) if hasattr(obj, '__iter__') else [obj]
    return [copy(seq[i % len(seq)]) for i in range(num_workers)]
  recommendations = []
  for worker, (worker_gp, worker_anc) in enumerate(zip(_per_worker(list_of_gps), _per_worker(anc_datas))):
    if worker > 0:
      worker_anc.eval_points_in_progress = recommendations
    recommendations.append(asy_acq(worker_gp, worker_anc))
  return recommendations


def _host_acquisition(gp, anc_data, formula):
  """ The reference's closure route: `formula(mu, sigma)` on the posterior of whatever points the
      maximiser asks for (hallucinated observations included), maximised by maximise_acquisition. """
  gp_eval = _get_gp_eval_for_parallel_strategy(gp, anc_data, 'std')
  return maximise_acquisition(lambda x: formula(*gp_eval(x)), anc_data)


def _acquire(gp, anc_data, acq, params, formula):
  """ One acquisition step: fused on the device when possible, the closure route otherwise. """
  if _can_fuse(gp, anc_data):
    return _fused_argmax(gp, acq, params, anc_data)
  return _host_acquisition(gp, anc_data, formula)


def _ndtr(x):
  from scipy.stats import norm as normal_distro
  return normal_distro.cdf(x)


def _expected_improvement_for_norm_diff(norm_diff):
  """ z Phi(z) + phi(z)  (gpb_acquisitions.py:247-249; host form, closure route only) """
  from scipy.stats import norm as normal_distro
  return norm_diff * normal_distro.cdf(norm_diff) + normal_distro.pdf(norm_diff)


# Thompson sampling ---------------------------------------------------------------------------
def asy_ts(gp, anc_data):
  """ gpb_acquisitions.py:119-127: TS always works on random candidates with one vectorised joint
      sample; a different configured method only multiplies the number of candidates by four. """
  anc_data = copy(anc_data)
  if anc_data.acq_opt_method != 'rand':
    anc_data.max_evals *= 4
    anc_data.acq_opt_method = 'rand'
  Xh = _halluc_points(anc_data)
  fused = Xh is None and anc_data.domain.get_type() == 'euclidean' and _is_device_gp(gp)
  if not fused:
    return maximise_acquisition(get_gp_sampler_for_parallel_strategy(gp, anc_data), anc_data,
                                vectorised=True)
  # covariance, stable_cholesky, L u and the arg-max stay on the device; the standard normals are
  # np.random.normal(size=(m, 1)) as in draw_gaussian_samples (general_utils.py:230) -- drawn in HBM
  # as well (Engine.random_normals continues the global state bit for bit): nothing of size m
  # crosses PCIe in either direction
  if DEVICE_CANDIDATES:
    cands, mean = _device_candidates(gp, anc_data)
    normals = gp.device_gp.engine.random_normals(cands.shape[0])
    _, idx = gp.device_gp.thompson(cands, normals, block=cands.shape[0], **mean)
    normals.free()
    return cands.row(idx)
  cands = _candidates(anc_data)
  normals = np.random.normal(size=(len(cands), 1)).ravel()
  if gaplog.ENABLED:
    _, idx, samples, _ = gp.device_gp.thompson(cands, normals, block=len(cands), mean_vals=gp.mean_func(cands), return_samples=True)
    gaplog.top2('thompson', samples)
    return cands[idx]
  _, idx = gp.device_gp.thompson(cands, normals, block=len(cands), mean_vals=gp.mean_func(cands))
  return cands[idx]


# Add-UCB -------------------------------------------------------------------------------------
def _get_add_ucb_beta_th(dim, time_step):
  """ gpb_acquisitions.py:135-137 """
  return np.sqrt(0.2 * dim * np.log(2 * dim * time_step + 1))


def _add_ucb(gp, add_kernel, mean_funcs, anc_data):
  """ gpb_acquisitions.py:139-189: one UCB maximisation per additive group, each over its own
      candidate set (max_evals // number of groups points, drawn group by group from the global
      np.random state as the reference's loop draws them; the acquisition itself consumes no random
      numbers), the winners assembled coordinate-wise.  With 'rand' all groups go to the device in
      ONE call: the per-group posteriors share the factor L and alpha in HBM and one triangular
      solve.  Other maximisers, and per-group mean functions, take the reference's loop. """
  if not isinstance(add_kernel, AdditiveKernel):
    raise TypeError('add_ucb needs a GP with an AdditiveKernel.')
  groupings = add_kernel.groupings
  all_bounds = np.asarray(anc_data.domain_bounds, dtype=np.float64)
  per_group_evals = int(anc_data.max_evals // len(add_kernel.kernel_list))
  betas = [_get_add_ucb_beta_th(len(grp), anc_data.t) for grp in groupings]
  point = np.zeros((sum(len(grp) for grp in groupings),))
  if mean_funcs is not None or not _is_rand_euclidean(anc_data):
    # the reference's loop shape (gpb_acquisitions.py:159-183): one maximisation per group over
    # the group's own box, by whatever maximiser is configured (the tree search for 'pdoo' /
    # 'direct'), the group's posterior evaluated on the device for the rows the maximiser asks for
    if mean_funcs is None:
      mean_funcs = lambda x: np.array([0] * len(x))
    if not hasattr(mean_funcs, '__iter__'):
      mean_funcs = [mean_funcs] * len(groupings)
    for j, (grp, beta, mean_func_j) in enumerate(zip(groupings, betas, mean_funcs)):
      def _group_acq(X_j, _j=j, _beta=beta, _mean=mean_func_j):
        X_j = _as_2d_array(X_j)
        return gp.device_gp.add_ucb_group(_j, _beta, X_j, return_vals=True)[2] + _mean(X_j)
      anc_j = copy(anc_data)
      anc_j.max_evals = per_group_evals
      anc_j.domain = _BoxDomain(all_bounds[grp])
      point[grp] = maximise_acquisition(_group_acq, anc_j)
    return point
  if DEVICE_CANDIDATES:
    engine = gp.device_gp.engine
    widths = [len(grp) for grp in groupings]
    starts = np.concatenate([[0], np.cumsum([per_group_evals * w for w in widths])])
    flat = engine.empty((int(starts[-1]),))
    for grp, start in zip(groupings, starts):
      engine.random_candidates(per_group_evals, len(grp), bounds=all_bounds[grp], out=flat.offset(start))
    _, winners = gp.device_gp.add_ucb_all(betas, flat, sizes=[per_group_evals] * len(groupings))
    for grp, start, idx in zip(groupings, starts, winners):
      point[grp] = flat.slice(int(start) + int(idx) * len(grp), len(grp))
    return point
  cands = [map_to_bounds(np.random.random((per_group_evals, len(grp))), all_bounds[grp])
           for grp in groupings]
  _, winners = gp.device_gp.add_ucb_all(betas, cands)
  for grp, cands_g, idx in zip(groupings, cands, winners):
    point[grp] = cands_g[int(idx)]
  return point


def asy_add_ucb(gp, anc_data):
  return _add_ucb(gp, gp.kernel, None, anc_data)


# UCB ------------------------------------------------------------------------------------------
def _get_gp_ucb_dim(gp):
  """ The dimension that enters beta_t (gpb_acquisitions.py:202-209): an explicit gp.ucb_dim, the
      kernel's dimension, or 3. """
  explicit = getattr(gp, 'ucb_dim', None)
  if explicit is not None:
    return explicit
  return getattr(gp.kernel, 'dim', 3.0)


def _get_ucb_beta_th(dim, time_step):
  """ gpb_acquisitions.py:211-213 """
  return np.sqrt(0.5 * dim * np.log(2 * dim * time_step + 1))


def asy_ucb(gp, anc_data):
  """ mu + beta_t sigma  (gpb_acquisitions.py:215-223) """
  beta_th = _get_ucb_beta_th(_get_gp_ucb_dim(gp), anc_data.t)
  return _acquire(gp, anc_data, 'ucb', (beta_th, 0.0), lambda mu, sigma: mu + beta_th * sigma)


# PI -------------------------------------------------------------------------------------------
def asy_pi(gp, anc_data):
  """ Phi((mu - best) / sigma)  (gpb_acquisitions.py:230-239) """
  best = anc_data.curr_max_val
  return _acquire(gp, anc_data, 'pi', (best, 0.0), lambda mu, sigma: _ndtr((mu - best) / sigma))


# EI -------------------------------------------------------------------------------------------
def asy_ei(gp, anc_data):
  """ sigma (z Phi(z) + phi(z)), z = (mu - best) / sigma  (gpb_acquisitions.py:251-261) """
  best = anc_data.curr_max_val
  return _acquire(gp, anc_data, 'ei', (best, 0.0),
                  lambda mu, sigma: sigma * _expected_improvement_for_norm_diff((mu - best) / sigma))


# TTEI -----------------------------------------------------------------------------------------
def _ttei(gp_eval, anc_data, ref_point, gp=None):
  """ Expected improvement over a reference point, with the combined standard deviation
      (gpb_acquisitions.py:269-280). """
  ref_mean, ref_std = [float(np.ravel(v)[0]) for v in gp_eval([ref_point])]
  if gp is not None and _can_fuse(gp, anc_data):
    return _fused_argmax(gp, 'ttei', (ref_mean, ref_std), anc_data)
  def _tt_ei_acq(x):
    mu, sigma = gp_eval(x)
    comb_std = np.sqrt(ref_std ** 2 + sigma ** 2)
    return comb_std * _expected_improvement_for_norm_diff((mu - ref_mean) / comb_std)
  return maximise_acquisition(_tt_ei_acq, anc_data)


def asy_ttei(gp, anc_data):
  """ Top-two EI (gpb_acquisitions.py:282-294): with probability 1/2 plain EI; otherwise EI with
      half the budget gives the reference point and the other half maximises the improvement over
      it.  The coin is the first np.random.random() call, as in the reference. """
  if np.random.random() < 0.5:
    return asy_ei(gp, anc_data)
  halved = copy(anc_data)
  halved.max_evals = anc_data.max_evals // 2
  ei_argmax = asy_ei(gp, halved)
  return _ttei(_get_gp_eval_for_parallel_strategy(gp, halved, 'std'), halved, ei_argmax, gp)


# Random ---------------------------------------------------------------------------------------
def asy_rand(_, anc_data):
  """ gpb_acquisitions.py:301-306: maximise a random acquisition.  The reference's objective
      returns ONE random number per call and is called point by point by its tree searches; the
      batched searches here hand over k points per call and get k numbers back (the 'rand'
      maximiser's single call with all candidates still draws one, as in the reference). """
  vectorised = anc_data.acq_opt_method == 'rand'
  return maximise_acquisition(lambda x: np.random.random((1,) if vectorised else (len(x),)), anc_data)


# Synchronous versions (gpb_acquisitions.py:129, 191, 225, 241, 263, 296, 308) -------------------
def _synchronous(asy_acq):
  def syn_acq(num_workers, list_of_gps, anc_datas):
    return _get_syn_recommendations_from_asy(asy_acq, num_workers, list_of_gps, anc_datas)
  syn_acq.__name__ = asy_acq.__name__.replace('asy_', 'syn_')
  return syn_acq


syn_ts, syn_add_ucb, syn_ucb = _synchronous(asy_ts), _synchronous(asy_add_ucb), _synchronous(asy_ucb)
syn_pi, syn_ei, syn_ttei, syn_rand = _synchronous(asy_pi), _synchronous(asy_ei), _synchronous(asy_ttei), \
                                     _synchronous(asy_rand)

_ASY = dict(ucb=asy_ucb, add_ucb=asy_add_ucb, ei=asy_ei, pi=asy_pi, ttei=asy_ttei, ts=asy_ts, rand=asy_rand)
asy = Namespace(**_ASY)
seq = Namespace(**_ASY)
syn = Namespace(ucb=syn_ucb, add_ucb=syn_add_ucb, ei=syn_ei, pi=syn_pi, ttei=syn_ttei, ts=syn_ts,
                rand=syn_rand)
