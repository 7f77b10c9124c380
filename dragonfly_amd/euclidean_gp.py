"""GPs on Euclidean spaces -- host-side mirror of dragonfly/gp/euclidean_gp.py:134-340,718-900
(EuclideanGP, EuclideanGPFitter, the kernel factory) and of the parts of GPFitter
(dragonfly/gp/gp_core.py:306-821) the maximum-likelihood route needs.

Every GP the fitter builds, and every log-marginal-likelihood it evaluates while tuning
hyper-parameters, is one dfh_gp_fit call on the MI355X (the training inputs are uploaded once per
fitter and stay in HBM: SURVEY.md section 8f item 1).  Stand-alone the fitter offers the
reference's maximum-likelihood strategies ('rand', 'rand_exp_sampling', the tree searches) and its
posterior sampling of the hyper-parameters ('post_sampling' with the slice sampler,
dragonfly_amd/hp_sampling.py) -- the bandit's default criterion is 'ml-post_sampling'
(opt/gp_bandit.py:51); only NUTS, which needs the gradient of the marginal likelihood, is left to
dragonfly_amd.install under a real Dragonfly.
"""
from argparse import Namespace
from itertools import product as itertools_product

import numpy as np

from . import kernel as gp_kernel
from .engine import get_engine
from .general_utils import map_to_bounds
from .gp_core import GP, ConstantMean
from .kernel import _as_2d_array
from .doo import pdoo_maximise_batched
from .oper_utils import random_maximise, random_sample_cts_dscr
from .hp_sampling import PosteriorHPSampler
from .hp_layout import consume_kernel_hps, describe_kernel_hps, group_kernel_args
from .option_handler import get_option_specs, load_options

_DFLT_KERNEL_TYPE = 'matern'

# gp_core.py:31-68 (mandatory_gp_args)
mandatory_gp_args = [
  get_option_specs('hp_tune_criterion', False, 'ml', 'ml | post_sampling'),
  get_option_specs('hp_tune_probs', False, 'uniform', ''),
  get_option_specs('ml_hp_tune_opt', False, 'default', 'rand | rand_exp_sampling | direct | pdoo'),
  get_option_specs('hp_tune_max_evals', False, -1, ''),
  get_option_specs('handle_non_psd_kernels', False, 'guaranteed_psd', ''),
  get_option_specs('mean_func', False, None, ''),
  get_option_specs('mean_func_type', False, 'tune', 'mean | median | const | zero | tune'),
  get_option_specs('mean_func_const', False, 0.0, ''),
  get_option_specs('noise_var_type', False, 'tune', 'tune | label | value'),
  get_option_specs('noise_var_label', False, 0.05, ''),
  get_option_specs('noise_var_value', False, 0.1, ''),
  get_option_specs('post_hp_tune_method', False, 'slice', ''),
  get_option_specs('post_hp_tune_burn', False, -1, ''),
  get_option_specs('post_hp_tune_offset', False, 25, ''),
  get_option_specs('rand_exp_sampling_replace', False, False, ''),
]
# euclidean_gp.py:27-73
basic_euc_gp_args = [
  get_option_specs('kernel_type', False, 'default', 'se | matern'),
  get_option_specs('use_same_bandwidth', False, False, ''),
]
matern_gp_args = [get_option_specs('matern_nu', False, 2.5, '')]
add_gp_args = [
  get_option_specs('use_additive_gp', False, False, ''),
  get_option_specs('add_max_group_size', False, 6, ''),
  get_option_specs('add_grouping_criterion', False, 'randomised_ml', ''),
  get_option_specs('num_groups_per_group_size', False, -1, ''),
  get_option_specs('add_group_size_criterion', False, 'sampled', ''),
]
euclidean_gp_args = mandatory_gp_args + basic_euc_gp_args + matern_gp_args + add_gp_args


class EuclideanGP(GP):
  """ euclidean_gp.py:134-202 """

  def __init__(self, X, Y, kernel, mean_func, noise_var,
               kernel_hyperparams=None, build_posterior=True, reporter=None):
    if isinstance(kernel, str):
      kernel = self._get_kernel_from_type(kernel, kernel_hyperparams)
    super(EuclideanGP, self).__init__(X, Y, kernel, mean_func, noise_var,
                                      build_posterior, reporter)

  @classmethod
  def _get_kernel_from_type(cls, kernel_type, kernel_hyperparams):
    """ A kernel given by name and a dictionary of its parameters (euclidean_gp.py:154-175); the
        two Euclidean kernels that run on the device. """
    hp = kernel_hyperparams
    if kernel_type == 'se':
      return gp_kernel.SEKernel(hp['dim'], hp['scale'], hp['dim_bandwidths'])
    if kernel_type == 'matern':
      return gp_kernel.MaternKernel(hp['dim'], hp['nu'], hp['scale'], hp['dim_bandwidths'])
    raise ValueError('Cannot construct kernel from kernel_type %s.' % (kernel_type))

  def _child_str(self):
    """ 'scale: .., <kernel>, mu(0)=..' as the reference prints it (euclidean_gp.py:177-183) """
    origin = [np.zeros(len(self.X[0]) if len(self.X) > 0 else 0)]
    return 'scale: %0.3f, %s, mu(0)=%0.3f' % (self.kernel.hyperparams['scale'],
                                             self._get_kernel_str(self.kernel),
                                             self.mean_func(origin)[0])

  @classmethod
  def _get_kernel_str(cls, kern):
    """ euclidean_gp.py:185-202 """
    if isinstance(kern, gp_kernel.AdditiveKernel):
      return str(kern)
    if isinstance(kern, (gp_kernel.SEKernel, gp_kernel.MaternKernel)):
      kern_name = 'se' if isinstance(kern, gp_kernel.SEKernel) else \
        'matern(%0.1f)' % (kern.hyperparams['nu'])
      bws = kern.hyperparams['dim_bandwidths']
      if kern.dim > 6:
        ret = '%0.4f(avg)' % (bws.mean())
      else:
        ret = '[' + ', '.join(['%0.3f'%(b) for b in np.ravel(bws)]) + ']'
      return kern_name + '-' + ret
    return ''


# Kernel factory ---------------------------------------------------------------------------------
def get_sublist_from_indices(orig_list, idxs):
  """ general_utils.py:30-33 """
  return [orig_list[idx] for idx in idxs]


def prep_euclidean_integral_kernel_hyperparams(kernel_type, gp_fitter_options, domain_dim):
  """ euclidean_gp.py:777-792 """
  hyperparams = {}
  hyperparams['dim'] = domain_dim
  if kernel_type == 'matern' and gp_fitter_options.matern_nu > 0:
    hyperparams['nu'] = gp_fitter_options.matern_nu
  return hyperparams


def get_euclidean_integral_gp_kernel(kernel_type, kernel_hyperparams, gp_cts_hps,
                                     gp_dscr_hps, use_same_bandwidth,
                                     add_gp_groupings=None, esp_kernel_type=None):
  """ euclidean_gp.py:796-805: the scale is the first continuous hyper-parameter. """
  scale = np.exp(gp_cts_hps[0])
  gp_cts_hps = gp_cts_hps[1:]
  return get_euclidean_integral_gp_kernel_with_scale(kernel_type, scale, \
    kernel_hyperparams, gp_cts_hps, gp_dscr_hps, use_same_bandwidth, add_gp_groupings, \
    esp_kernel_type)


def get_euclidean_integral_gp_kernel_with_scale(kernel_type, scale, kernel_hyperparams,
                                                gp_cts_hps, gp_dscr_hps,
                                                use_same_bandwidth, add_gp_groupings=None,
                                                esp_kernel_type=None):
  """ The kernel a hyper-parameter vector describes (interface of euclidean_gp.py:808-900 for the
      se / matern / poly / expdecay kernels, plain or additive; the reference's 'esp' kernels are not
      part of the device path).  Which values of the two vectors feed which argument of which kernel
      class is the table of dragonfly_amd/hp_layout.py.  An additive kernel carries the scale itself
      and its groups' kernels scale 1 (kernel.py:484-494).  Returns (kernel, left-over continuous,
      left-over discrete hyper-parameters). """
  # pylint: disable=unused-argument
  dim = kernel_hyperparams['dim']
  values, cls_name, gp_cts_hps, gp_dscr_hps = consume_kernel_hps(kernel_type, dim, kernel_hyperparams,
                                                                 gp_cts_hps, gp_dscr_hps, use_same_bandwidth)
  additive = add_gp_groupings is not None
  groups = add_gp_groupings if additive else [list(range(dim))]
  make = getattr(gp_kernel, cls_name)
  parts = [make(dim=len(grp), scale=1.0 if additive else scale, **group_kernel_args(kernel_type, values, grp))
           for grp in groups]
  kernel = gp_kernel.AdditiveKernel(scale=scale, kernel_list=parts, groupings=groups) if additive else parts[0]
  return kernel, gp_cts_hps, gp_dscr_hps


# Additive-model helpers (euclidean_gp.py:718-774) -------------------------------------------------
def optimise_cts_hps_for_given_dscr_hps_in_add_model(given_dscr_hps, num_groups_per_group_size, dim,
                                                     hp_tune_max_evals, cts_hp_optimise,
                                                     tuning_objective):
  """ Additive-model search (euclidean_gp.py:718-746): the last discrete hyper-parameter is the
      group size; several random partitions of the coordinates into groups of that size are tried,
      the continuous hyper-parameters are optimised for each (with the evaluation budget split
      between them, at least 500 each) and the best (value, hps, groupings) wins.  A negative
      number of partitions means the reference's default: one for singleton groups (all
      partitions are the same), else between 5 and 25 growing with the dimension.  The random
      permutations come from the global np.random state, one np.random.permutation(dim) each. """
  group_size = given_dscr_hps[-1]
  num_partitions = num_groups_per_group_size
  if num_partitions < 0:
    num_partitions = 1 if group_size == 1 else max(5, min(2 * dim, 25))
  evals_each = int(max(500, hp_tune_max_evals / num_partitions))
  best = (-np.inf, None, None)
  for _ in range(num_partitions):
    order = list(np.random.permutation(dim))
    partition = Namespace(add_gp_groupings=[order[lo:lo + group_size] for lo in range(0, dim, group_size)])
    objective = lambda cts_hps, _p=partition: tuning_objective(cts_hps, given_dscr_hps[:], other_gp_params=_p)
    value, cts_hps, _ = cts_hp_optimise(objective, evals_each)
    if value > best[0]:
      best = (value, cts_hps, partition)
  return best


class EuclideanGPFitter(object):
  """ Fits a GP by tuning the kernel hyper-parameters (euclidean_gp.py:205-340 on top of
      gp_core.py:306-821).  Hyper-parameter vector order (euclidean_gp.py:217-219): continuous =
      [mean value (if tuned), log noise variance (if tuned), log scale, log bandwidths...],
      discrete = [matern nu (if tuned), additive group size (if additive)]. """
  # pylint: disable=too-many-instance-attributes

  _option_specs = euclidean_gp_args

  def __init__(self, X, Y, options=None, reporter=None):
    assert len(X) == len(Y)
    self.dim = len(X[0])
    self.reporter = reporter
    self.options = load_options(self._option_specs, partial_options=options)
    self.X = X
    self.Y = np.asarray(Y, dtype=np.float64)
    self.num_data = len(X)
    self._X_dev = None
    self.batch_lml = True      # False: one dfh_gp_fit per candidate (the reference's loop shape)
    self.pdoo_frontier = 32    # open boxes prefetched per batch by the direct / pdoo tuners (0: one at a time)
    self._set_up()

  # -- set up (gp_core.py:323-356, 393-416; euclidean_gp.py:215-276) -------------------------------
  def _set_up(self):
    self.cts_hp_bounds = []
    self.dscr_hp_vals = []
    self.param_order = []
    epsilon = 0.0001
    self.Y_var = np.array(self.Y).std() ** 2 + epsilon if len(self.Y) > 0 else epsilon
    self._set_up_mean_and_noise_variance_bounds()
    self._child_set_up()
    self.methods_to_use = [elem.lower() for elem in self.options.hp_tune_criterion.split('-')]
    for method in self.methods_to_use:
      if method not in ['ml', 'post_sampling', 'post_mean']:
        raise ValueError('hp_tune_criterion should be ml or post_sampling.')
      if method == 'post_mean':
        raise NotImplementedError('Not implemented post_mean yet.')        # gp_core.py:497-499: neither has the reference
    # probabilities of the tuning methods (gp_core.py:358-378, 380-392)
    self.methods_to_use_counter = {key: 0 for key in self.methods_to_use}
    probs = self.options.hp_tune_probs
    n_methods = len(self.methods_to_use)
    if probs == 'uniform':
      self.hp_tune_probs = np.ones(n_methods) / float(n_methods)
    elif probs == 'adaptive':
      self.hp_tune_uniform_sampling_prob = 0.05
      self.hp_tune_sampling_weights = {key: 1.0 for key in self.methods_to_use}
      self.hp_tune_probs = self._get_adaptive_hp_tune_probs()
    else:
      self.hp_tune_probs = np.array([float(x) for x in probs.split('-')])
      if len(self.hp_tune_probs) != n_methods:
        self.hp_tune_probs = np.ones(n_methods) / float(n_methods)
    self.hp_tune_probs = self.hp_tune_probs / self.hp_tune_probs.sum()
    self.cts_hp_bounds = np.array(self.cts_hp_bounds)
    self.num_hps = len(self.cts_hp_bounds) + len(self.dscr_hp_vals)
    self._set_up_ml_hp_tune()
    if 'post_sampling' in self.methods_to_use and self.options.post_hp_tune_method not in ('slice', 'nuts'):
      raise ValueError('Unknown post_hp_tune_method %s.' % (self.options.post_hp_tune_method))

  def _get_adaptive_hp_tune_probs(self):
    """ gp_core.py:380-392: a uniform floor plus weights ~ successes / sqrt(1 + uses) """
    n_methods = len(self.methods_to_use)
    floor = self.hp_tune_uniform_sampling_prob * np.ones((n_methods,)) / n_methods
    successes = np.array([self.hp_tune_sampling_weights[key] for key in self.methods_to_use])
    uses = np.array([self.methods_to_use_counter[key] for key in self.methods_to_use])
    weights = successes / np.sqrt(1 + uses)
    ret = floor + (1 - self.hp_tune_uniform_sampling_prob) * weights / weights.sum()
    return ret / ret.sum()

  def update_hp_tune_method_weight(self, method, weight_to_add=1):
    """ gp_core.py:743-746 """
    if self.options.hp_tune_probs == 'adaptive':
      self.hp_tune_sampling_weights[method] += weight_to_add

  def _has_user_mean_func(self):
    return getattr(self.options, 'mean_func', None) is not None

  def _set_up_mean_and_noise_variance_bounds(self):
    """ gp_core.py:393-416.  A tuned constant mean lives within three widths of the median of the
        labels, the width being the average of half the label range and the label standard deviation
        (no labels: median 0, half range 1); a tuned noise variance within [0.005, 0.2] Var(Y), searched
        in logs.  Both come before the kernel's hyper-parameters in the continuous vector. """
    if self.options.mean_func_type == 'tune' and not self._has_user_mean_func():
      labels = np.asarray(self.Y, dtype=np.float64)
      centre, half_range = (np.median(labels), 0.5 * (labels.max() - labels.min())) if labels.size else (0.0, 1.0)
      width = 0.5 * (half_range + np.sqrt(self.Y_var))
      self.mean_func_bounds = [centre - 3 * width, centre + 3 * width]
      self._add_cts_hp("noise_mean", self.mean_func_bounds)      # (sic: the reference's label)
    if self.options.noise_var_type == 'tune':
      self.noise_var_log_bounds = [np.log(frac * self.Y_var) for frac in (0.005, 0.2)]
      self._add_cts_hp("noise_var", self.noise_var_log_bounds)

  def _add_cts_hp(self, name, box):
    self.cts_hp_bounds.append(box)
    self.param_order.append([name, "cts"])

  def _child_set_up(self):
    """ euclidean_gp.py:215-252 """
    if self.options.kernel_type not in ['se', 'matern', 'poly', 'esp', 'default']:
      raise ValueError('Unknown kernel_type. Should be either se, matern or poly.')
    if self.options.kernel_type == 'poly':
      raise NotImplementedError('Not implemented Poly kernel yet.')       # euclidean_gp.py:280-282: nor has the reference
    if self.options.kernel_type == 'esp':
      raise NotImplementedError('esp kernels are outside the device path (SURVEY.md section 2); '
                                'use the reference fitter for them.')
    if self.options.noise_var_type not in ['tune', 'label', 'value']:
      raise ValueError('Unknown noise_var_type. Should be either tune, label or value.')
    if self.options.mean_func_type not in ['mean', 'median', 'const', 'zero', 'tune']:
      raise ValueError('Unknown mean_func_type. Should be mean/median/const/zero/tune.')
    self.kernel_type = _DFLT_KERNEL_TYPE if self.options.kernel_type == 'default' else \
                       self.options.kernel_type
    # the scale, then the kernel's own hyper-parameters in the order of hp_layout.KERNEL_HP_LAYOUT
    # (boxes as in euclidean_gp.py:254-268: scale within [0.1, 10] Var(Y), bandwidths within
    # [0.01, 10] x the Frobenius norm of the inputs, nu over {0.5, 1.5, 2.5})
    self.scale_log_bounds = [np.log(0.1 * self.Y_var), np.log(10 * self.Y_var)]
    self.param_order.append(["scale", "cts"])
    X_std_norm = np.linalg.norm(_as_2d_array(self.X), 'fro') + 1e-4
    single_bandwidth_log_bounds = [np.log(0.01 * X_std_norm), np.log(10 * X_std_norm)]
    boxes = {'dim_bandwidths': single_bandwidth_log_bounds, 'same_dim_bandwidths': single_bandwidth_log_bounds}
    value_lists = {'nu': [0.5, 1.5, 2.5]}
    kernel_hyperparams = prep_euclidean_integral_kernel_hyperparams(self.kernel_type, self.options, self.dim)
    self.bandwidth_log_bounds = []
    for name, source, count in describe_kernel_hps(self.kernel_type, self.dim, kernel_hyperparams,
                                                   self.options.use_same_bandwidth):
      if source == 'cts':
        self.bandwidth_log_bounds += [boxes[name]] * count
        self.param_order += [[name, "cts"] for _ in range(count)]
      else:
        self.dscr_hp_vals.append(value_lists[name])
        self.param_order.append([name, "dscr"])
    self.cts_hp_bounds += [self.scale_log_bounds] + self.bandwidth_log_bounds
    if self.options.use_additive_gp:
      self.add_group_size_idx_in_dscr_hp_vals = len(self.dscr_hp_vals)
      self.add_max_group_size = min(self.options.add_max_group_size, self.dim)
      self.dscr_hp_vals.append([x+1 for x in range(self.add_max_group_size)])
      self.param_order.append(["additive_grp", "dscr"])

  def _set_up_ml_hp_tune(self):
    """ gp_core.py:423-474.  Every optimiser sees the tuning objective as a *batch* objective
        (candidates in, log marginal likelihoods out, one dfh_gp_lml_batch call):
          rand / rand_exp_sampling  the whole random sample is one batch (the reference evaluates
                 it one candidate at a time, vectorised=False, gp_core.py:437,441; the random draws
                 are the same calls in the same order, so a seeded run picks the same candidates);
          pdoo   the tree search of dragonfly_amd.doo, a frontier of boxes per batch;
          direct the reference's own fall-back when its Fortran DIRECT is not built -- PDOO
                 (oper_utils.py:130-133); stand-alone there is no Fortran DIRECT, so always.
        'default' resolves as in the reference (gp_core.py:77-82): direct up to 60
        hyper-parameters, pdoo beyond. """
    method = self.options.ml_hp_tune_opt
    if method == 'default':
      method = 'pdoo' if self.num_hps > 60 else 'direct'
    if method not in ['rand', 'rand_exp_sampling', 'direct', 'pdoo']:
      raise ValueError('Unknown ml_hp_tune_opt %s.' % (method))
    self.ml_hp_tune_opt_method = method
    if self.options.hp_tune_max_evals is not None and self.options.hp_tune_max_evals > 0:
      self.hp_tune_max_evals = self.options.hp_tune_max_evals
    elif method in ['direct', 'pdoo']:
      self.hp_tune_max_evals = min(1e4, max(500, self.num_hps * 50))
    elif method == 'rand':
      self.hp_tune_max_evals = min(1e4, max(500, self.num_hps * 200))
    else:
      self.hp_tune_max_evals = min(1e5, max(500, self.num_hps * 400))
    def _rand_wrap(obj, max_evals):
      opt_val, opt_pt, _ = random_maximise(obj, self.cts_hp_bounds, max_evals, vectorised=True)
      return opt_val, opt_pt, None
    def _tree_wrap(obj, max_evals):
      opt_val, opt_pt, _ = pdoo_maximise_batched(obj, self.cts_hp_bounds, max_evals,
                                                 frontier=self.pdoo_frontier,
                                                 depth=2 if self.pdoo_frontier > 0 else 0)
      return opt_val, opt_pt, None
    def _rand_exp_sampling_wrap(obj, max_evals):
      sample_cts_hps, sample_dscr_hps, lml_vals = \
        random_sample_cts_dscr(obj, self.cts_hp_bounds, self.dscr_hp_vals, max_evals,
                               vectorised=True)
      sample_probs = np.exp(lml_vals - max(lml_vals))
      sample_probs = sample_probs / sample_probs.sum()
      return sample_cts_hps, sample_dscr_hps, sample_probs
    self.cts_hp_optimise = _rand_wrap if method == 'rand' else _tree_wrap
    self.hp_sampler = _rand_exp_sampling_wrap

  def _uses_additive_model(self):
    return self.options.use_additive_gp

  # -- building GPs (gp_core.py:501-543; euclidean_gp.py:325-339) ----------------------------------
  def _device_X(self):
    """ The training inputs are uploaded once and reused by every candidate GP. """
    if self._X_dev is None:
      self._X_dev = get_engine().to_device(_as_2d_array(self.X))
    return self._X_dev

  # mean_func_type -> the constant a fitted GP takes as its prior mean (gp_core.py:510-521); 'tune'
  # reads it off the hyper-parameter vector instead, anything else (e.g. 'zero') is 0
  _CONSTANT_MEANS = {
    'mean': lambda fitter: np.mean(fitter.Y),
    'median': lambda fitter: np.median(fitter.Y),
    'upper_bound': lambda fitter: np.mean(fitter.Y) + 3 * np.std(fitter.Y),
    'const': lambda fitter: fitter.options.mean_func_const,
  }

  def _mean_and_noise_from_hps(self, gp_cts_hps):
    """ gp_core.py:506-538: peels the mean value and the log noise variance -- whichever of the two are
        tuned, in that order -- off the front of the continuous hyper-parameters.  Returns (mean_func,
        its constant value or None for a user mean function, noise_var, remaining hyper-parameters). """
    rest = gp_cts_hps
    if self._has_user_mean_func():
      mean_func, const = self.options.mean_func, None
    else:
      if self.options.mean_func_type == 'tune':
        const, rest = np.asarray(rest[0]).item(), rest[1:]
      else:
        const = self._CONSTANT_MEANS.get(self.options.mean_func_type, lambda fitter: 0)(self)
      mean_func = ConstantMean(const)
    noise_var_type = self.options.noise_var_type
    if noise_var_type == 'tune':
      noise_var, rest = np.exp(rest[0]), rest[1:]
    elif noise_var_type == 'label':
      noise_var = self.options.noise_var_label * (self.Y.std() ** 2)
    else:
      noise_var = self.options.noise_var_value
    return mean_func, const, noise_var, rest

  def build_gp(self, gp_cts_hps, gp_dscr_hps, other_gp_params=None, *args, **kwargs):
    """ gp_core.py:501-543 """
    if self.num_hps != len(gp_cts_hps) + len(gp_dscr_hps):
      raise ValueError('gp_hyperparams should be of length %d. Given length: %d.'%(
          self.num_hps, len(gp_cts_hps) + len(gp_dscr_hps)))
    mean_func, _, noise_var, gp_cts_hps = self._mean_and_noise_from_hps(gp_cts_hps)
    built = self._child_build_gp(mean_func, noise_var, gp_cts_hps, gp_dscr_hps,
                                 other_gp_params=other_gp_params, *args, **kwargs)
    assert all(len(left_over) == 0 for left_over in built[1:])     # every hyper-parameter found its place
    return built[0]

  def _kernel_from_hps(self, gp_cts_hps, gp_dscr_hps, other_gp_params=None):
    """ euclidean_gp.py:325-336: the kernel of a candidate, and the left-over hyper-parameters. """
    kernel_hyperparams = prep_euclidean_integral_kernel_hyperparams(self.kernel_type,
                                                                    self.options, self.dim)
    add_gp_groupings = None
    if self.options.use_additive_gp:
      gp_dscr_hps = gp_dscr_hps[:-1]
      add_gp_groupings = other_gp_params.add_gp_groupings
    return get_euclidean_integral_gp_kernel(self.kernel_type, kernel_hyperparams, gp_cts_hps,
                                            gp_dscr_hps, self.options.use_same_bandwidth,
                                            add_gp_groupings)

  def _child_build_gp(self, mean_func, noise_var, gp_cts_hps, gp_dscr_hps,
                      other_gp_params=None, *args, **kwargs):
    """ euclidean_gp.py:325-339 """
    kernel, gp_cts_hps, gp_dscr_hps = self._kernel_from_hps(gp_cts_hps, gp_dscr_hps, other_gp_params)
    build_posterior = kwargs.pop('build_posterior', True)
    ret_gp = EuclideanGP(self.X, self.Y, kernel, mean_func, noise_var, *args,
                         build_posterior=False, **kwargs)
    ret_gp._X_dev_hint = self._device_X()     # pylint: disable=protected-access
    if build_posterior:
      ret_gp.build_posterior()
    return ret_gp, gp_cts_hps, gp_dscr_hps

  def _tuning_objective(self, gp_cts_hps, gp_dscr_hps, other_gp_params=None, *args, **kwargs):
    """ gp_core.py:551-564: log marginal likelihood of the GP with these hyper-parameters. The
        candidate is fitted on the device and released at once (only its lml is kept). """
    built_gp = self.build_gp(gp_cts_hps, gp_dscr_hps, other_gp_params=other_gp_params,
                             *args, **kwargs)
    ret = built_gp.compute_log_marginal_likelihood()
    built_gp._invalidate()     # pylint: disable=protected-access
    return ret

  def _tuning_objective_batch(self, cts_hps_list, dscr_hps, other_gp_params=None):
    """ The tuning objective (gp_core.py:551-564) for a list of candidates: one dfh_gp_lml_batch
        call (include/dfhip.h).  dscr_hps is one list shared by all candidates or one list per
        candidate.  Returns an ndarray of log marginal likelihoods, in candidate order. """
    num = len(cts_hps_list)
    per_cand_dscr = len(dscr_hps) > 0 and isinstance(dscr_hps[0], (list, tuple, np.ndarray))
    if num == 0:
      return np.zeros((0,))
    if not self.batch_lml or (hasattr(self.options, 'mean_func') and self.options.mean_func is not None):
      # a user mean function is an arbitrary host callable: one fit per candidate
      return np.array([self._tuning_objective(cts, list(dscr_hps[i]) if per_cand_dscr else list(dscr_hps),
                                              other_gp_params=other_gp_params)
                       for i, cts in enumerate(cts_hps_list)])
    specs, mean_consts, noise_vars = [], [], []
    for i, cts in enumerate(cts_hps_list):
      dscr = list(dscr_hps[i]) if per_cand_dscr else list(dscr_hps)
      if self.num_hps != len(cts) + len(dscr):
        raise ValueError('gp_hyperparams should be of length %d. Given length: %d.'%(
            self.num_hps, len(cts) + len(dscr)))
      _, mean_const, noise_var, rest = self._mean_and_noise_from_hps(cts)
      kernel, left_cts, left_dscr = self._kernel_from_hps(rest, dscr, other_gp_params)
      assert len(left_cts) == 0 and len(left_dscr) == 0
      if not kernel.has_device_spec():
        # a composition the device does not evaluate itself (host-kernel mode): one fit per candidate
        return np.array([self._tuning_objective(c, list(dscr_hps[j]) if per_cand_dscr else list(dscr_hps),
                                                other_gp_params=other_gp_params)
                         for j, c in enumerate(cts_hps_list)])
      specs.append(kernel.to_spec(self.dim))
      mean_consts.append(float(mean_const))
      noise_vars.append(float(noise_var))
    return get_engine().gp_lml_batch(specs, self._device_X(), self.Y, mean_consts, noise_vars)

  def _optimise_cts_hps_for_given_dscr_hps(self, given_dscr_hps):
    """ gp_core.py:576-583 / euclidean_gp.py:303-313 """
    if self.options.use_additive_gp:
      return optimise_cts_hps_for_given_dscr_hps_in_add_model(list(given_dscr_hps), \
        self.options.num_groups_per_group_size, self.dim, self.hp_tune_max_evals, \
        self.cts_hp_optimise, self._tuning_objective_batch)
    cts_tuning_obj = lambda arg: self._tuning_objective_batch(arg, list(given_dscr_hps))
    opt_cts_val, opt_cts_hps, _ = self.cts_hp_optimise(cts_tuning_obj, self.hp_tune_max_evals)
    return opt_cts_val, opt_cts_hps, None

  def lml_batch(self, cts_hps_list, dscr_hps_list, other_gp_params=None):
    """ what dragonfly_amd.hp_sampling asks of a fitter: the log marginal likelihoods of a list of
        (continuous, discrete) hyper-parameter candidates, one device call """
    return self._tuning_objective_batch(cts_hps_list, dscr_hps_list, other_gp_params)

  def _sample_cts_dscr_hps_for_post_sampling(self, num_samples):
    """ gp_core.py:592-726 """
    additive = self.options.use_additive_gp
    sampler = PosteriorHPSampler(self, add_dim=self.dim if additive else None,
                                 add_max_group_size=self.add_max_group_size if additive else None)
    return sampler.sample(num_samples)

  def _sample_hps_for_rand_exp_sampling_in_add_model(self):
    """ euclidean_gp.py:748-775: per sample a group size, a random partition of the coordinates, the
        other discrete hyper-parameters and a uniform point of the continuous box -- drawn in that
        order from the global np.random state --; weights exp(lml) WITHOUT the shift by the maximum
        that the non-additive sampler applies (the reference's arithmetic).  One device call per
        sample: every sample has its own grouping. """
    gidx = self.add_group_size_idx_in_dscr_hp_vals
    cts_all, dscr_all, other_all, vals = [], [], [], []
    for _ in range(int(self.hp_tune_max_evals)):
      group_size = np.random.choice(self.dscr_hp_vals[gidx])
      order = list(np.random.permutation(self.dim))
      other = Namespace(add_gp_groupings=[order[i:i + group_size] for i in range(0, self.dim, group_size)])
      dscr = [np.random.choice(categ) for categ in self.dscr_hp_vals]
      dscr[gidx] = group_size
      cts = map_to_bounds(np.random.random((len(self.cts_hp_bounds),)), self.cts_hp_bounds)
      vals.append(self._tuning_objective_batch([cts], [dscr], other)[0])
      cts_all.append(cts); dscr_all.append(dscr); other_all.append(other)
    probs = np.exp(vals)
    return cts_all, dscr_all, other_all, probs / probs.sum()

  def fit_gp(self, num_samples=1, hp_tune_criterion=None):
    """ gp_core.py:783-821. Returns ('fitted_gp', gp, (cts_hps, dscr_hps)); for rand_exp_sampling
        ('sample_hps_with_probs', cts, dscr, other_params, probs); for post_sampling
        ('post_fitted_gp', gp, hps) with one sample, else ('post_sample_hps_with_probs', cts, dscr,
        other_params). """
    if hp_tune_criterion is None:
      hp_tune_criterion = self.options.hp_tune_criterion
    if hp_tune_criterion == 'post_sampling':
      sample_cts_hps, sample_dscr_hps, sample_other_gp_params = \
        self._sample_cts_dscr_hps_for_post_sampling(num_samples)
      if num_samples == 1:
        opt_gp = self.build_gp(sample_cts_hps[0], sample_dscr_hps[0],
                               other_gp_params=sample_other_gp_params[0])
        return 'post_fitted_gp', opt_gp, (sample_cts_hps, sample_dscr_hps)
      return ('post_sample_hps_with_probs', sample_cts_hps, sample_dscr_hps, sample_other_gp_params)
    if hp_tune_criterion != 'ml':
      raise ValueError('hp_tune_criterion should be ml or post_sampling.')
    if self.ml_hp_tune_opt_method in ['direct', 'rand', 'pdoo']:
      # every combination of the discrete hyper-parameters gets its own continuous search; the first
      # combination reaching the highest likelihood wins (gp_core.py:789-803)
      winner = (-np.inf, None, None, None)
      for dscr_hps in itertools_product(*self.dscr_hp_vals):
        value, cts_hps, other_params = self._optimise_cts_hps_for_given_dscr_hps(dscr_hps)
        if value > winner[0]:
          winner = (value, list(cts_hps), list(dscr_hps), other_params)
      _, best_cts_hps, best_dscr_hps, best_other_params = winner
      opt_gp = self.build_gp(best_cts_hps, best_dscr_hps, other_gp_params=best_other_params)
      return 'fitted_gp', opt_gp, (best_cts_hps, best_dscr_hps)
    if self._uses_additive_model():
      sample_cts_hps, sample_dscr_hps, sample_other_gp_params, sample_probs = \
        self._sample_hps_for_rand_exp_sampling_in_add_model()
      return ('sample_hps_with_probs', sample_cts_hps, sample_dscr_hps,
              sample_other_gp_params, sample_probs)
    sample_cts_hps, sample_dscr_hps, sample_probs = \
      self.hp_sampler(self._tuning_objective_batch, self.hp_tune_max_evals)
    sample_other_gp_params = [None] * len(sample_cts_hps)
    return ('sample_hps_with_probs', sample_cts_hps, sample_dscr_hps,
            sample_other_gp_params, sample_probs)

  def fit_gp_for_gp_bandit(self, num_samples=1):
    """ gp_core.py:748-781 """
    self.hp_tune_results = {}
    for method in self.methods_to_use:
      ret = self.fit_gp(num_samples, method)
      kind = ret[0]
      if kind in ('fitted_gp', 'post_fitted_gp'):
        kept = ret[1]
      elif kind == 'sample_hps_with_probs':
        # num_samples of the weighted candidates; without replacement only if the option says so AND enough
        # of them carry weight (gp_core.py:762-770)
        candidates, probs = list(zip(*ret[1:4])), ret[-1]
        replace = True if sum(probs > 0) < num_samples else self.options.rand_exp_sampling_replace
        picks = np.random.choice(len(candidates), size=(num_samples,), replace=replace, p=probs)
        kept = [candidates[idx] for idx in picks]
      elif kind == 'post_sample_hps_with_probs':
        kept = list(zip(*ret[1:4]))
      else:
        raise ValueError('Unknown option %s for results of fit_gp.' % (kind))
      self.hp_tune_results[method] = (kind, kept)

  def get_next_gp(self):
    """ gp_core.py:728-741 """
    if self.options.hp_tune_probs == 'adaptive':
      self.hp_tune_probs = self._get_adaptive_hp_tune_probs()
    # p= as in the reference: it draws one uniform even for a single method (the no-p form does not)
    method = np.random.choice(self.methods_to_use, p=self.hp_tune_probs)
    fit_type = self.hp_tune_results[method][0]
    if fit_type in ('fitted_gp', 'post_fitted_gp'):
      gp = self.hp_tune_results[method][1]
    else:
      next_gp_hps = self.hp_tune_results[method][1].pop(0)
      self.hp_tune_results[method][1].append(next_gp_hps)
      gp = self.build_gp(next_gp_hps[0], next_gp_hps[1], other_gp_params=next_gp_hps[2],
                         build_posterior=False)
    return (fit_type, method, gp)
