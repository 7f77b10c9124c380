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
from argparse import Namespace
from copy import copy

import numpy as np

from .general_utils import map_to_bounds
from .kernel import AdditiveKernel, _as_2d_array
from .oper_utils import random_maximise

# A maximiser for non-'rand' methods; dragonfly_amd.install points this at
# dragonfly.exd.exd_utils.maximise_with_method.
external_maximise_with_method = None


def _candidates(anc_data, max_evals=None):
  """ random_sample's draw (oper_utils.py:61-62): map_to_bounds(np.random.random((m, d)), b). """
  bounds = np.asarray(anc_data.domain.bounds, dtype=np.float64)
  m = int(anc_data.max_evals if max_evals is None else max_evals)
  return map_to_bounds(np.random.random((m, len(bounds))), bounds)


def _is_rand_euclidean(anc_data):
  return anc_data.domain.get_type() == 'euclidean' and \
         str(anc_data.acq_opt_method).lower().startswith('rand')


def _can_fuse(gp, anc_data):
  """ The fused candidates -> posterior -> acquisition -> arg-max call needs a device kernel; GPs
      whose kernel the host evaluates take the reference's closure route (batched gp.eval). """
  return _is_rand_euclidean(anc_data) and gp.num_tr_data > 0 and not getattr(gp, '_generic', False)


def maximise_acquisition(acq_fn, anc_data, *args, **kwargs):
  """ gpb_acquisitions.py:23-40 for host-evaluated acquisition callables. """
  acq_opt_method = anc_data.acq_opt_method
  if anc_data.domain.get_type() != 'euclidean':
    raise NotImplementedError('dragonfly_amd acquisitions handle Euclidean domains.')
  if str(acq_opt_method).lower().startswith('rand'):
    kwargs.pop('vectorised', None)
    _, opt_pt, _ = random_maximise(acq_fn, anc_data.domain.bounds, anc_data.max_evals)
    return opt_pt
  if external_maximise_with_method is None:
    raise NotImplementedError(
        'acq_opt_method=%s is a per-point serial search in the reference (DIRECT/PDOO); only '
        '"rand" is fused on the device. Install under Dragonfly (dragonfly_amd.install) to use '
        'its maximisers.' % (acq_opt_method))
  acquisition = lambda x: acq_fn(x.reshape((1, -1)))
  _, opt_pt = external_maximise_with_method(acq_opt_method, acquisition, anc_data.domain,
                                            anc_data.max_evals, *args, **kwargs)
  return opt_pt


def _halluc_points(anc_data):
  """ gpb_acquisitions.py:56-64: the in-progress points when handle_parallel == 'halluc'. """
  if getattr(anc_data, 'handle_parallel', None) == 'halluc' and \
     len(getattr(anc_data, 'eval_points_in_progress', [])) > 0:
    if getattr(anc_data, 'is_mf', False):
      raise NotImplementedError('Multi-fidelity acquisitions are out of scope.')
    return _as_2d_array(anc_data.eval_points_in_progress)
  return None


def _get_gp_eval_for_parallel_strategy(gp, anc_data, uncert_form='std'):
  """ gpb_acquisitions.py:43-64 """
  Xh = _halluc_points(anc_data)
  if Xh is not None:
    return lambda x: gp.eval_with_hallucinated_observations(x, Xh, uncert_form=uncert_form)
  return lambda x: gp.eval(x, uncert_form=uncert_form)


def get_gp_sampler_for_parallel_strategy(gp, anc_data):
  """ gpb_acquisitions.py:67-87 """
  Xh = _halluc_points(anc_data)
  if Xh is not None:
    return lambda x: gp.draw_samples_with_hallucinated_observations(1, x, Xh).ravel()
  return lambda x: gp.draw_samples(1, x).ravel()


def _fused_argmax(gp, acq, params, anc_data, max_evals=None):
  """ Candidates -> posterior -> acquisition -> arg-max in one device call; returns the point. """
  cands = _candidates(anc_data, max_evals)
  test_mean = gp.mean_func(cands)
  Xh = _halluc_points(anc_data)
  _, idx = gp.device_gp.acq_argmax(acq, cands, params=params, mean_vals=test_mean, X_halluc=Xh)
  return cands[idx]


def _get_syn_recommendations_from_asy(asy_acq, num_workers, list_of_gps, anc_datas):
  """ gpb_acquisitions.py:90-115: earlier recommendations become hallucinated points. """
  def _get_next_and_append(_list_of_objects):
    ret = _list_of_objects.pop(0)
    _list_of_objects = _list_of_objects + [ret]
    return ret, _list_of_objects
  if not hasattr(list_of_gps, '__iter__'):
    list_of_gps = [list_of_gps] * num_workers
  if not hasattr(anc_datas, '__iter__'):
    anc_datas = [anc_datas] * num_workers
  list_of_gps = [copy(gp) for gp in list_of_gps]
  anc_datas = [copy(ad) for ad in anc_datas]
  next_gp, list_of_gps = _get_next_and_append(list_of_gps)
  next_anc_data, anc_datas = _get_next_and_append(anc_datas)
  recommendations = [asy_acq(next_gp, next_anc_data)]
  for _ in range(1, num_workers):
    next_gp, list_of_gps = _get_next_and_append(list_of_gps)
    next_anc_data, anc_datas = _get_next_and_append(anc_datas)
    next_anc_data.eval_points_in_progress = recommendations
    recommendations.append(asy_acq(next_gp, next_anc_data))
  return recommendations


# Thompson sampling ---------------------------------------------------------------------------
def asy_ts(gp, anc_data):
  """ gpb_acquisitions.py:119-127: always random candidates + a vectorised joint sample. """
  anc_data = copy(anc_data)
  if anc_data.acq_opt_method != 'rand':
    anc_data.acq_opt_method = 'rand'
    anc_data.max_evals = 4 * anc_data.max_evals
  Xh = _halluc_points(anc_data)
  if Xh is None and gp.num_tr_data > 0 and anc_data.domain.get_type() == 'euclidean' and \
     not getattr(gp, '_generic', False):
    # fused: covariance, stable_cholesky, L u and the arg-max stay on the device
    cands = _candidates(anc_data)
    test_mean = gp.mean_func(cands)
    U = np.random.normal(size=(len(cands), 1))          # general_utils.py:230
    _, idx = gp.device_gp.thompson(cands, U.ravel(), block=len(cands), mean_vals=test_mean)
    return cands[idx]
  gp_sample = get_gp_sampler_for_parallel_strategy(gp, anc_data)
  return maximise_acquisition(gp_sample, anc_data, vectorised=True)


def syn_ts(num_workers, list_of_gps, anc_datas):
  return _get_syn_recommendations_from_asy(asy_ts, num_workers, list_of_gps, anc_datas)


# Add-UCB -------------------------------------------------------------------------------------
def _get_add_ucb_beta_th(dim, time_step):
  """ gpb_acquisitions.py:135-137 """
  return np.sqrt(0.2 * dim * np.log(2 * dim * time_step + 1))


def _add_ucb(gp, add_kernel, mean_funcs, anc_data):
  """ gpb_acquisitions.py:139-189: one UCB maximisation per additive group, each over its own
      candidate set; the per-group posterior uses the shared factor L and alpha in HBM. """
  if not isinstance(add_kernel, AdditiveKernel):
    raise TypeError('add_ucb needs a GP with an AdditiveKernel.')
  if mean_funcs is not None:
    raise NotImplementedError('Per-group mean functions are not used by the reference (None).')
  if not str(anc_data.acq_opt_method).lower().startswith('rand'):
    raise NotImplementedError('add_ucb on the device engine uses acq_opt_method="rand".')
  groupings = add_kernel.groupings
  total_max_evals = anc_data.max_evals
  domain_bounds = np.asarray(anc_data.domain_bounds, dtype=np.float64)
  num_groups = len(add_kernel.kernel_list)
  group_points = []
  num_coordinates = 0
  anc_data.max_evals = total_max_evals//num_groups
  # The candidate sets are drawn group by group exactly as the reference's loop does (the
  # acquisition itself consumes no random numbers), then all groups go to the device in one call.
  betas, cands = [], []
  for j, group_j in enumerate(groupings):
    betas.append(_get_add_ucb_beta_th(len(group_j), anc_data.t))
    bounds_j = domain_bounds[group_j]
    cands.append(map_to_bounds(np.random.random((int(anc_data.max_evals), len(bounds_j))), bounds_j))
  _, idxs = gp.device_gp.add_ucb_all(betas, cands)
  for cands_j, idx in zip(cands, idxs):
    point_j = cands_j[int(idx)]
    group_points.append(point_j)
    num_coordinates += len(point_j)
  anc_data.max_evals = total_max_evals
  ret = np.zeros((num_coordinates,))
  for point_j, group_j in zip(group_points, groupings):
    ret[group_j] = point_j
  return ret


def asy_add_ucb(gp, anc_data):
  return _add_ucb(gp, gp.kernel, None, anc_data)


def syn_add_ucb(num_workers, list_of_gps, anc_datas):
  return _get_syn_recommendations_from_asy(asy_add_ucb, num_workers, list_of_gps, anc_datas)


# UCB ------------------------------------------------------------------------------------------
def _get_gp_ucb_dim(gp):
  """ gpb_acquisitions.py:202-209 """
  if hasattr(gp, 'ucb_dim') and gp.ucb_dim is not None:
    return gp.ucb_dim
  elif hasattr(gp.kernel, 'dim'):
    return gp.kernel.dim
  else:
    return 3.0


def _get_ucb_beta_th(dim, time_step):
  """ gpb_acquisitions.py:211-213 """
  return np.sqrt(0.5 * dim * np.log(2 * dim * time_step + 1))


def asy_ucb(gp, anc_data):
  """ gpb_acquisitions.py:215-223 """
  beta_th = _get_ucb_beta_th(_get_gp_ucb_dim(gp), anc_data.t)
  if _can_fuse(gp, anc_data):
    return _fused_argmax(gp, 'ucb', (beta_th, 0.0), anc_data)
  gp_eval = _get_gp_eval_for_parallel_strategy(gp, anc_data, 'std')
  def _ucb_acq(x):
    mu, sigma = gp_eval(x)
    return mu + beta_th * sigma
  return maximise_acquisition(_ucb_acq, anc_data)


def syn_ucb(num_workers, list_of_gps, anc_datas):
  return _get_syn_recommendations_from_asy(asy_ucb, num_workers, list_of_gps, anc_datas)


# PI -------------------------------------------------------------------------------------------
def _ndtr(x):
  from scipy.stats import norm as normal_distro
  return normal_distro.cdf(x)


def asy_pi(gp, anc_data):
  """ gpb_acquisitions.py:230-239 """
  curr_best = anc_data.curr_max_val
  if _can_fuse(gp, anc_data):
    return _fused_argmax(gp, 'pi', (curr_best, 0.0), anc_data)
  gp_eval = _get_gp_eval_for_parallel_strategy(gp, anc_data, 'std')
  def _pi_acq(x):
    mu, sigma = gp_eval(x)
    return _ndtr((mu - curr_best) / sigma)
  return maximise_acquisition(_pi_acq, anc_data)


def syn_pi(num_workers, list_of_gps, anc_datas):
  return _get_syn_recommendations_from_asy(asy_pi, num_workers, list_of_gps, anc_datas)


# EI -------------------------------------------------------------------------------------------
def _expected_improvement_for_norm_diff(norm_diff):
  """ gpb_acquisitions.py:247-249 (host form, used on the per-point fallback route only) """
  from scipy.stats import norm as normal_distro
  return norm_diff * normal_distro.cdf(norm_diff) + normal_distro.pdf(norm_diff)


def asy_ei(gp, anc_data):
  """ gpb_acquisitions.py:251-261 """
  curr_best = anc_data.curr_max_val
  if _can_fuse(gp, anc_data):
    return _fused_argmax(gp, 'ei', (curr_best, 0.0), anc_data)
  gp_eval = _get_gp_eval_for_parallel_strategy(gp, anc_data, 'std')
  def _ei_acq(x):
    mu, sigma = gp_eval(x)
    norm_diff = (mu - curr_best) / sigma
    return sigma * _expected_improvement_for_norm_diff(norm_diff)
  return maximise_acquisition(_ei_acq, anc_data)


def syn_ei(num_workers, list_of_gps, anc_datas):
  return _get_syn_recommendations_from_asy(asy_ei, num_workers, list_of_gps, anc_datas)


# TTEI -----------------------------------------------------------------------------------------
def _ttei(gp_eval, anc_data, ref_point, gp=None):
  """ gpb_acquisitions.py:269-280 """
  ref_mean, ref_std = gp_eval([ref_point])
  ref_mean = float(np.ravel(ref_mean)[0])
  ref_std = float(np.ravel(ref_std)[0])
  if gp is not None and _can_fuse(gp, anc_data):
    return _fused_argmax(gp, 'ttei', (ref_mean, ref_std), anc_data)
  def _tt_ei_acq(x):
    mu, sigma = gp_eval(x)
    comb_std = np.sqrt(ref_std**2 + sigma**2)
    norm_diff = (mu - ref_mean)/comb_std
    return comb_std * _expected_improvement_for_norm_diff(norm_diff)
  return maximise_acquisition(_tt_ei_acq, anc_data)


def asy_ttei(gp, anc_data):
  """ gpb_acquisitions.py:282-294 """
  if np.random.random() < 0.5:
    return asy_ei(gp, anc_data)
  else:
    max_acq_opt_evals = anc_data.max_evals
    anc_data = copy(anc_data)
    anc_data.max_evals = max_acq_opt_evals//2
    ei_argmax = asy_ei(gp, anc_data)
    gp_eval = _get_gp_eval_for_parallel_strategy(gp, anc_data, 'std')
    return _ttei(gp_eval, anc_data, ei_argmax, gp)


def syn_ttei(num_workers, list_of_gps, anc_data):
  return _get_syn_recommendations_from_asy(asy_ttei, num_workers, list_of_gps, anc_data)


# Random ---------------------------------------------------------------------------------------
def asy_rand(_, anc_data):
  """ gpb_acquisitions.py:301-306: maximise a random acquisition. """
  def _rand_eval(_):
    return np.random.random((1,))
  return maximise_acquisition(_rand_eval, anc_data)


def syn_rand(num_workers, list_of_gps, anc_datas):
  return _get_syn_recommendations_from_asy(asy_rand, num_workers, list_of_gps, anc_datas)


syn = Namespace(ucb=syn_ucb, add_ucb=syn_add_ucb, ei=syn_ei, pi=syn_pi, ttei=syn_ttei, ts=syn_ts,
                rand=syn_rand)
asy = Namespace(ucb=asy_ucb, add_ucb=asy_add_ucb, ei=asy_ei, pi=asy_pi, ttei=asy_ttei, ts=asy_ts,
                rand=asy_rand)
seq = Namespace(ucb=asy_ucb, add_ucb=asy_add_ucb, ei=asy_ei, pi=asy_pi, ttei=asy_ttei, ts=asy_ts,
                rand=asy_rand)
