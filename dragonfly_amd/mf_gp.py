"""Multi-fidelity GP on the MI355X -- host-side mirror of dragonfly/gp/mf_gp.py:18-107 (class MFGP),
dragonfly/gp/euclidean_gp.py:347-412 (class EuclideanMFGP) and :418-710 (class EuclideanMFGPFitter).  A multi-fidelity GP is a GP on the
joint (fidelity, domain) points with a product kernel; everything numerical is the same device
path as dragonfly_amd.gp_core.GP (kernel matrix, Cholesky, posterior)."""
import numpy as np

from . import kernel as gp_kernel
from . import euclidean_gp as egp
from .gp_core import GP
from .option_handler import get_option_specs


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
    both_kernels = fidel_kernel is not None and domain_kernel is not None
    if both_kernels:
      # the two factor kernels decide the split (and are what get_fidel_kernel / get_domain_kernel hand out)
      self.fidel_kernel, self.domain_kernel = fidel_kernel, domain_kernel
      self.fidel_dim, self.domain_dim = fidel_kernel.dim, domain_kernel.dim
    else:
      # a ready-made joint kernel: the split has to be named, the data alone do not decide it
      # (euclidean_gp.py:364-368)
      try:
        self.fidel_dim, self.domain_dim = kwargs.pop('fidel_dim'), kwargs.pop('domain_dim')
      except KeyError:
        raise Exception('Specify fidel_dim and domain_dim.')
    joint_dim = self.fidel_dim + self.domain_dim
    # joint point = [z, x]: fidelity coordinates first
    self.fidel_coords = list(range(self.fidel_dim))
    self.domain_coords = list(range(self.fidel_dim, joint_dim))
    if mf_kernel is None:
      mf_kernel = gp_kernel.CoordinateProductKernel(joint_dim, kernel_scale, [fidel_kernel, domain_kernel],
                                                    [self.fidel_coords, self.domain_coords])
    super(EuclideanMFGP, self).__init__(ZZ, XX, YY, mf_kernel, mean_func, noise_var,
                                        *args, **kwargs)

  def _test_fidel_domain_dims(self, test_fidel_dim, test_domain_dim):
    """ euclidean_gp.py:379-385: the reference's message for points of the wrong shape """
    if (test_fidel_dim, test_domain_dim) != (self.fidel_dim, self.domain_dim):
      raise ValueError('ZZ, XX dimensions should be (%d, %d). Given (%d, %d)'%( \
                       self.fidel_dim, self.domain_dim, test_fidel_dim, test_domain_dim))

  def get_ZX_from_ZZ_XX(self, ZZ, XX):
    """ euclidean_gp.py:387-403.  With the fidelity coordinates first the reference's re-ordering of
        the concatenated coordinates is the identity, so the joint point is the concatenation: a list
        of rows for a list of points, one vector for a single point, [] for no points. """
    if hasattr(ZZ, '__iter__') and len(ZZ) == 0:
      return []
    single = not hasattr(ZZ[0], '__iter__')
    if single:
      self._test_fidel_domain_dims(len(ZZ), len(XX))
      return np.concatenate((ZZ, XX))
    self._test_fidel_domain_dims(len(ZZ[0]), len(XX[0]))
    return list(np.hstack((np.asarray(ZZ), np.asarray(XX))))

  def _training_rows(self, rows, data_idxs):
    return list(rows[:self.num_tr_data]) if data_idxs is None else [rows[i] for i in data_idxs]

  def get_domain_pts(self, data_idxs=None):
    """ euclidean_gp.py:405-408 """
    return self._training_rows(self.XX, data_idxs)

  def get_fidel_pts(self, data_idxs=None):
    """ euclidean_gp.py:410-413 """
    return self._training_rows(self.ZZ, data_idxs)


# Options of the multi-fidelity fitter (euclidean_gp.py:75-130); the 'esp' kernels are not part of
# the device path and their options are left out.
basic_mf_euc_gp_args = [
  get_option_specs('fidel_kernel_type', False, 'se', 'se | matern | poly | expdecay'),
  get_option_specs('fidel_matern_nu', False, 2.5, 'negative: tuned'),
  get_option_specs('fidel_use_same_bandwidth', False, False, ''),
  get_option_specs('fidel_use_same_scalings', False, False, ''),
  get_option_specs('fidel_poly_order', False, -1, ''),
  get_option_specs('domain_kernel_type', False, 'se', 'se | matern | poly'),
  get_option_specs('domain_matern_nu', False, 2.5, 'negative: tuned'),
  get_option_specs('domain_use_same_bandwidth', False, False, ''),
  get_option_specs('domain_use_same_scalings', False, False, ''),
  get_option_specs('domain_poly_order', False, -1, ''),
  get_option_specs('domain_use_additive_gp', False, False, ''),
  get_option_specs('domain_add_max_group_size', False, 6, ''),
  get_option_specs('domain_add_grouping_criterion', False, 'randomised_ml', ''),
  get_option_specs('domain_num_groups_per_group_size', False, -1, ''),
  get_option_specs('domain_add_group_size_criterion', False, 'sampled', ''),
]


class EuclideanMFGPFitter(egp.EuclideanGPFitter):
  """ Fits a multi-fidelity GP by maximum likelihood (euclidean_gp.py:418-710 on top of
      gp_core.py:306-821): the tuners, the batched tuning objective (one dfh_gp_lml_batch call per
      batch of candidates) and the bandit bookkeeping are EuclideanGPFitter's; this class adds the
      hyper-parameter layout of the product kernel scale * k_fidel(z, z') * k_domain(x, x').
      Continuous hyper-parameters, in order: [mean value, log noise variance (if tuned)], log scale,
      the fidelity kernel's (log bandwidths | log offset, log powers), the domain kernel's log
      bandwidths; discrete: [fidelity nu], [domain nu], [additive group size] (if tuned). """

  _option_specs = egp.mandatory_gp_args + basic_mf_euc_gp_args

  def __init__(self, ZZ, XX, YY, options=None, reporter=None):
    assert len(ZZ) == len(XX) == len(YY)
    self.ZZ, self.XX, self.YY = ZZ, XX, YY
    self.fidel_dim = len(ZZ[0])
    self.domain_dim = len(XX[0])
    self.input_dim = self.fidel_dim + self.domain_dim
    self.num_tr_data = len(YY)
    joint = np.concatenate((egp._as_2d_array(ZZ), egp._as_2d_array(XX)), axis=1)   # [z, x]: EuclideanMFGP's layout
    super(EuclideanMFGPFitter, self).__init__(list(joint), YY, options, reporter)

  def _uses_additive_model(self):
    return self.options.domain_use_additive_gp

  # -- set up (euclidean_gp.py:433-619) -----------------------------------------------------------
  def _child_set_up(self):
    opts = self.options
    if opts.fidel_kernel_type not in ['se', 'matern', 'poly', 'expdecay']:
      raise ValueError('Unknown fidel_kernel_type. Should be in {se, matern, poly, ' +
                       'expdecay.')
    if opts.domain_kernel_type not in ['se', 'matern', 'poly']:
      raise ValueError('Unknown domain_kernel_type. Should be either se or poly.')
    if opts.noise_var_type not in ['tune', 'label', 'value']:
      raise ValueError('Unknown noise_var_type. Should be either tune, label or value.')
    if opts.mean_func_type not in ['mean', 'median', 'const', 'zero', 'upper_bound', 'tune']:
      raise ValueError(('Unknown mean_func_type. Should be one of ', 'mean/median/const/zero.'))
    self.ZZ_std_norm = np.linalg.norm(egp._as_2d_array(self.ZZ), 'fro') + 5e-5
    self.XX_std_norm = np.linalg.norm(egp._as_2d_array(self.XX), 'fro') + 5e-5
    self.ZX_std_norm = np.sqrt(self.ZZ_std_norm**2 + self.XX_std_norm**2)
    self.scale_log_bounds = [np.log(0.1 * self.Y_var), np.log(10 * self.Y_var)]
    self.cts_hp_bounds.append(self.scale_log_bounds)
    self.param_order.append(["scale", "cts"])
    # fidelity kernel, then domain kernel: the order of the hyper-parameter vector
    if opts.fidel_kernel_type in ('se', 'matern'):
      self.fidel_bandwidth_log_bounds = self._bandwidth_bounds_and_order(
          'fidel_bandwidth_log_bounds', self.fidel_dim, opts.fidel_use_same_bandwidth,
          opts.fidel_use_same_bandwidth)
      if opts.fidel_kernel_type == 'matern' and opts.fidel_matern_nu < 0:
        self.dscr_hp_vals.append([0.5, 1.5, 2.5])
        self.param_order.append(["nu", "dscr"])
    elif opts.fidel_kernel_type == 'poly':
      self._get_poly_kernel_bounds(self.ZZ, self.XX, opts.fidel_use_same_scalings)
    else:
      self._fidel_expdecay_kernel_setup()
    if opts.domain_kernel_type in ('se', 'matern'):
      # the reference sizes these bounds per dimension whatever domain_use_same_bandwidth says
      # (euclidean_gp.py:574-575 passes False) and only records the order by the option
      self.domain_bandwidth_log_bounds = self._bandwidth_bounds_and_order(
          'domain_bandwidth_log_bounds', self.domain_dim, False, opts.domain_use_same_bandwidth)
      if opts.domain_kernel_type == 'matern' and opts.domain_matern_nu < 0:
        self.dscr_hp_vals.append([0.5, 1.5, 2.5])
        self.param_order.append(["nu", "dscr"])
    else:
      self._get_poly_kernel_bounds(self.ZZ, self.XX, opts.domain_use_same_scalings)
    if opts.domain_use_additive_gp:      # after the kernels: the group size is the LAST discrete value
      self.domain_add_group_size_idx_in_dscr_hp_vals = len(self.dscr_hp_vals)
      self.domain_add_max_group_size = min(opts.domain_add_max_group_size, self.domain_dim)
      self.dscr_hp_vals.append([x+1 for x in range(self.domain_add_max_group_size)])
      self.param_order.append(["additive_grp", "dscr"])

  def _bandwidth_bounds_and_order(self, option_name, dim, same_for_bounds, same_for_order):
    """ euclidean_gp.py:496-511, 567-582: user bounds if given, else [0.01, 10] x the joint data norm """
    given = getattr(self.options, option_name, None)
    bounds = given if given is not None else \
             self._get_bandwidth_log_bounds(dim, self.ZX_std_norm, same_for_bounds)
    self.cts_hp_bounds.extend(bounds)
    if same_for_order:
      self.param_order.append(["same_dim_bandwidths", "cts"])
    else:
      for _ in range(dim):
        self.param_order.append(["dim_bandwidths", "cts"])
    return bounds

  @classmethod
  def _get_bandwidth_log_bounds(cls, dim, single_bw_bounds, use_same_bandwidth):
    """ euclidean_gp.py:606-614 """
    if isinstance(single_bw_bounds, (float, int)):
      single_bw_bounds = [0.01*single_bw_bounds, 10*single_bw_bounds]
    single = [np.log(x) for x in single_bw_bounds]
    return [single] if use_same_bandwidth else [single] * dim

  def _fidel_expdecay_kernel_setup(self):
    """ euclidean_gp.py:520-540: log offset in [0.1, 10] x Var(Y)/sqrt(n), log powers in [0.1, 50] """
    given = getattr(self.options, 'fidel_expdecay_offset_log_bounds', None)
    if given is not None:
      self.fidel_expdecay_offset_log_bounds = given
    else:
      scale_range = self.Y_var / np.sqrt(self.num_tr_data)
      self.fidel_expdecay_offset_log_bounds = [np.log(0.1 * scale_range), np.log(10 * scale_range)]
    given = getattr(self.options, 'fidel_expdecay_power_log_bounds', None)
    self.fidel_expdecay_power_log_bounds = given if given is not None else \
                                           [[np.log(1e-1), np.log(50)]] * self.fidel_dim
    self.cts_hp_bounds.append(self.fidel_expdecay_offset_log_bounds)
    self.cts_hp_bounds.extend(self.fidel_expdecay_power_log_bounds)

  def _get_poly_kernel_bounds(self, ZZ, XX, use_same_scalings):
    """ euclidean_gp.py:616-618: the reference has no bounds for polynomial kernels either """
    raise NotImplementedError('Yet to implement polynomial kernel.')

  # -- kernels and GPs from hyper-parameters (euclidean_gp.py:640-710) -----------------------------
  @classmethod
  def _factor_hyperparams(cls, kernel_type, dim, matern_nu, poly_order):
    hyperparams = {'dim': dim}
    if kernel_type == 'matern' and matern_nu > 0:
      hyperparams['nu'] = matern_nu
    elif kernel_type == 'poly':
      hyperparams['order'] = poly_order
    return hyperparams

  def _mf_kernels_from_hps(self, gp_cts_hps, gp_dscr_hps, other_gp_params=None):
    """ (scale, fidelity kernel, domain kernel, left-over cts, left-over dscr); the fidelity kernel
        consumes its hyper-parameters first (euclidean_gp.py:682-706) """
    opts = self.options
    ke_scale = np.exp(gp_cts_hps[0])
    gp_cts_hps = gp_cts_hps[1:]
    fidel_kernel, gp_cts_hps, gp_dscr_hps = egp.get_euclidean_integral_gp_kernel_with_scale(
        opts.fidel_kernel_type, 1.0,
        self._factor_hyperparams(opts.fidel_kernel_type, self.fidel_dim, opts.fidel_matern_nu, opts.fidel_poly_order),
        gp_cts_hps, gp_dscr_hps, opts.fidel_use_same_bandwidth, None)
    add_gp_groupings = None
    if opts.domain_use_additive_gp:
      gp_dscr_hps = gp_dscr_hps[:-1]
      add_gp_groupings = other_gp_params.add_gp_groupings
    domain_kernel, gp_cts_hps, gp_dscr_hps = egp.get_euclidean_integral_gp_kernel_with_scale(
        opts.domain_kernel_type, 1.0,
        self._factor_hyperparams(opts.domain_kernel_type, self.domain_dim, opts.domain_matern_nu, opts.domain_poly_order),
        gp_cts_hps, gp_dscr_hps, opts.domain_use_same_bandwidth, add_gp_groupings)
    return ke_scale, fidel_kernel, domain_kernel, gp_cts_hps, gp_dscr_hps

  def _kernel_from_hps(self, gp_cts_hps, gp_dscr_hps, other_gp_params=None):
    """ The joint kernel of a candidate, as EuclideanMFGP composes it (for the batched objective). """
    ke_scale, fidel_kernel, domain_kernel, gp_cts_hps, gp_dscr_hps = \
      self._mf_kernels_from_hps(gp_cts_hps, gp_dscr_hps, other_gp_params)
    coords = [list(range(self.fidel_dim)), list(range(self.fidel_dim, self.input_dim))]
    kernel = gp_kernel.CoordinateProductKernel(self.input_dim, ke_scale, [fidel_kernel, domain_kernel], coords)
    return kernel, gp_cts_hps, gp_dscr_hps

  def _child_build_gp(self, mean_func, noise_var, gp_cts_hps, gp_dscr_hps,
                      other_gp_params=None, *args, **kwargs):
    """ euclidean_gp.py:678-710 """
    ke_scale, fidel_kernel, domain_kernel, gp_cts_hps, gp_dscr_hps = \
      self._mf_kernels_from_hps(gp_cts_hps, gp_dscr_hps, other_gp_params)
    build_posterior = kwargs.pop('build_posterior', True)
    ret_gp = EuclideanMFGP(self.ZZ, self.XX, self.YY, None, ke_scale, fidel_kernel, domain_kernel,
                           mean_func, noise_var, *args, build_posterior=False, reporter=self.reporter, **kwargs)
    ret_gp._X_dev_hint = self._device_X()     # pylint: disable=protected-access
    if build_posterior:
      ret_gp.build_posterior()
    return ret_gp, gp_cts_hps, gp_dscr_hps

  def _optimise_cts_hps_for_given_dscr_hps(self, given_dscr_hps):
    """ euclidean_gp.py:622-633 """
    if not self.options.domain_use_additive_gp:
      cts_tuning_obj = lambda arg: self._tuning_objective_batch(arg, list(given_dscr_hps))
      opt_cts_val, opt_cts_hps, _ = self.cts_hp_optimise(cts_tuning_obj, self.hp_tune_max_evals)
      return opt_cts_val, opt_cts_hps, None
    return egp.optimise_cts_hps_for_given_dscr_hps_in_add_model(list(given_dscr_hps), \
      self.options.domain_num_groups_per_group_size, self.domain_dim, self.hp_tune_max_evals, \
      self.cts_hp_optimise, self._tuning_objective_batch)
