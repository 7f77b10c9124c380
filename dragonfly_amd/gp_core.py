"""GP posterior on the MI355X -- host-side mirror of dragonfly/gp/gp_core.py:86-303 (class GP).

Same constructor, attributes and methods as the reference's GP, so it is a drop-in for the code
that consumes a GP (gp_bandit, the acquisitions, user code).  The kernel matrix, its Cholesky
factor, alpha, the log marginal likelihood, the posterior mean / std / covariance and the
Gaussian draws are all computed by libdfhip.so and stay resident in HBM; `L`, `alpha` and
`K_trtr_wo_noise` are materialised as NumPy arrays only when something reads them
(gpb_acquisitions.py:169,171 and gp_core.py:203 do).  There is no NumPy compute path.
"""
import os
import sys

import numpy as np

from .engine import get_engine
from .general_utils import draw_gaussian_samples
from .kernel import _as_2d_array


class ConstantMean(object):
  """ The constant prior mean the fitter gives a GP (gp_core.py:527-530: one value per point).  The
      value is exposed so that device-resident candidates need not come back to the host just to
      have a constant evaluated on them. """

  def __init__(self, value):
    self.constant_value = value

  def __call__(self, x):
    return np.array([self.constant_value] * len(x))


def _check_feature_label_lengths_and_format(X, Y):
  """ gp_core.py:72-76 """
  if len(X) != len(Y):
    raise ValueError('Length of X (' + str(len(X)) + ') and Y (' + \
      str(len(Y)) + ') do not match.')


# DFH_HOST_CANDIDATES=1 keeps every random draw of the acquisitions on the host (A/B switch)
DEVICE_NORMALS = os.environ.get('DFH_HOST_CANDIDATES', '0') != '1'


class GP(object):
  """ Base class for Gaussian processes (gp_core.py:86). """
  # pylint: disable=attribute-defined-outside-init
  # pylint: disable=too-many-instance-attributes

  def __init__(self, X, Y, kernel, mean_func, noise_var, build_posterior=True,
               reporter=None, handle_non_psd_kernels='guaranteed_psd'):
    super(GP, self).__init__()
    _check_feature_label_lengths_and_format(X, Y)
    self._fitted = None
    self._cache = {}
    self.incremental_updates = True   # add_data_*: append to the cached factor when possible
    self.set_data(X, Y, build_posterior=False)
    self.kernel = kernel
    self.mean_func = mean_func
    self.noise_var = noise_var
    self.reporter = reporter
    self.handle_non_psd_kernels = handle_non_psd_kernels
    self.num_tr_data = len(self.Y)
    self._set_up()
    if build_posterior:
      self.build_posterior()

  def _set_up(self):
    """ gp_core.py:112-118: a kernel that is not guaranteed PSD needs a way to handle it.  The
        device builds 'project_first' / 'try_before_project' posteriors from a caller-evaluated
        Gram matrix (dfh_gp_fit_gram with the projection of csrc/psdproj.hip), so such GPs run in
        host-kernel mode whatever the kernel object. """
    if self.handle_non_psd_kernels not in ('guaranteed_psd', 'project_first', 'try_before_project'):
      raise ValueError('Unknown option for handle_non_psd_kernels: %s' % (self.handle_non_psd_kernels))
    if not self.kernel.is_guaranteed_psd():
      assert self.handle_non_psd_kernels in ['project_first', 'try_before_project']

  @property
  def _generic(self):
    """ True when the kernel is evaluated by the caller: a kernel object without a device
        description (any Kernel with is_guaranteed_psd(), e.g. the reference's PolyKernel), or a
        subclass that overrides the documented hook _get_training_kernel_matrix (gp_core.py:149).
        The Gram / cross matrices then come from the host; Cholesky, solves and the posterior still
        run on the device (dfh_gp_fit_gram, dfh_gp_predict_gram). """
    has_spec = hasattr(self.kernel, 'to_spec') and \
               getattr(self.kernel, 'has_device_spec', lambda: True)()
    return (not has_spec) or self.handle_non_psd_kernels != 'guaranteed_psd' or \
           type(self)._get_training_kernel_matrix is not GP._get_training_kernel_matrix

  def _write_message(self, msg):
    if self.reporter:
      self.reporter.write(msg)
    else:
      sys.stdout.write(msg)

  # -- data ----------------------------------------------------------------------------------
  def set_data(self, X, Y, build_posterior=True):
    """ gp_core.py:127-133 """
    self.X = list(X)
    self.Y = list(Y)
    self.num_tr_data = len(self.Y)
    self._X_dev_hint = None
    self._invalidate()
    if build_posterior:
      self.build_posterior()

  def add_data_single(self, x_new, y_new, *args, **kwargs):
    """ gp_core.py:135-137 """
    self.add_data_multiple([x_new], [y_new], *args, **kwargs)

  def add_data_multiple(self, X_new, Y_new, build_posterior=True):
    """ gp_core.py:139-146 """
    _check_feature_label_lengths_and_format(X_new, Y_new)
    fitted, fit_sig, n_old = self._fitted, getattr(self, '_fit_sig', None), self.num_tr_data
    self.X.extend(X_new)
    self.Y.extend(Y_new)
    self.num_tr_data = len(self.Y)
    self._X_dev_hint = None
    self._invalidate()
    if build_posterior:
      # The reference rebuilds from scratch; with an unchanged kernel and noise variance the
      # extended posterior is a block-row append of the cached factor (dfh_gp_append, O(n^2 q)).
      if fitted is not None and len(X_new) > 0 and fitted.n == n_old and self.incremental_updates \
         and not self._generic and fit_sig == self._posterior_signature(fitted.d):
        Y_centred = np.asarray(self.Y, dtype=np.float64) - self.mean_func(self.X)
        self._fitted = fitted.append(_as_2d_array(X_new), Y_centred)
        self._fit_sig = fit_sig
      else:
        self.build_posterior()

  def _invalidate(self):
    # drop (not free): shallow copies made by the synchronous acquisitions share the handle;
    # the HBM buffers are released when the last reference to the FittedGP goes away.
    self._fitted = None
    self._cache = {}

  # -- posterior -------------------------------------------------------------------------------
  def _get_training_kernel_matrix(self):
    """ gp_core.py:149-153 """
    return self.kernel(self.X, self.X)

  def _X_array(self):
    return _as_2d_array(self.X)

  def build_posterior(self):
    """ gp_core.py:155-163: K, L = chol(K + noise I), alpha -- one device call. """
    self._invalidate()
    if self.num_tr_data == 0:
      return
    if self._generic:
      Y_centred = np.asarray(self.Y, dtype=np.float64) - self.mean_func(self.X)
      K = np.ascontiguousarray(self._get_training_kernel_matrix(), dtype=np.float64)
      self._cache['K'] = K
      self._fitted = get_engine().gp_fit_gram(K, Y_centred, self.noise_var,
                                              handle_non_psd_kernels=self.handle_non_psd_kernels)
      self._fit_sig = None
      return
    hint = getattr(self, '_X_dev_hint', None)      # training inputs already resident in HBM
    X = hint if (hint is not None and hint.shape[0] == self.num_tr_data) else self._X_array()
    Y_centred = np.asarray(self.Y, dtype=np.float64) - self.mean_func(self.X)
    spec = self.kernel.to_spec(in_dim=X.shape[1])
    self._fitted = get_engine().gp_fit(spec, X, Y_centred, self.noise_var)
    self._fit_sig = self._posterior_signature(X.shape[1])

  def _posterior_signature(self, in_dim):
    """ What the cached factor depends on besides the data: kernel and noise variance. """
    return (self.kernel.to_spec(in_dim=in_dim).signature(), float(self.noise_var))

  def _need_fit(self):
    if self._fitted is None and self.num_tr_data > 0:
      raise RuntimeError('The posterior has not been built (call build_posterior()).')
    return self._fitted

  @property
  def L(self):
    """ Lower Cholesky factor of K + noise_var I (n x n ndarray; gp_core.py:158). """
    if self.num_tr_data == 0 or self._fitted is None:
      return None
    if 'L' not in self._cache:
      self._cache['L'] = self._fitted.get_L()
    return self._cache['L']

  @property
  def alpha(self):
    """ (K + noise_var I)^-1 (Y - m(X))  (gp_core.py:162-163). """
    if self.num_tr_data == 0 or self._fitted is None:
      return None
    if 'alpha' not in self._cache:
      self._cache['alpha'] = self._fitted.get_alpha()
    return self._cache['alpha']

  @property
  def K_trtr_wo_noise(self):
    """ kernel(X, X)  (gp_core.py:157). """
    if self.num_tr_data == 0 or self._fitted is None:
      return None
    if 'K' not in self._cache:
      self._cache['K'] = self._get_training_kernel_matrix() if self._generic else self._fitted.get_K()
    return self._cache['K']

  @property
  def device_gp(self):
    """ The FittedGP handle (HBM-resident posterior) for fused device calls. """
    return self._need_fit()

  def eval(self, X_test, uncert_form='none'):
    """ gp_core.py:165-190. uncert_form: 'none' | 'std' | 'covar'. """
    if uncert_form not in ('none', 'std', 'covar'):
      raise ValueError('uncert_form should be none, covar or std.')
    test_mean = self.mean_func(X_test)
    Xt = X_test if self._generic else _as_2d_array(X_test)   # a caller-evaluated kernel takes any objects
    if self.num_tr_data == 0:
      # no data: the posterior is the prior (K_tetr is n_test x 0 in the reference)
      pred_mean = test_mean + np.zeros(len(Xt))
      if uncert_form == 'none':
        return pred_mean, None
      K_tete = self.kernel(Xt, Xt)
      return pred_mean, (K_tete if uncert_form == 'covar' else np.sqrt(np.diag(K_tete)))
    fitted = self._need_fit()
    if self._generic:
      # gp_core.py:172-188 with the caller's kernel; the solves and products run on the device
      K_tetr = np.ascontiguousarray(self.kernel(X_test, self.X), dtype=np.float64)
      if uncert_form == 'none':
        mu_raw, _ = fitted.predict_gram(K_tetr)
        return test_mean + mu_raw, None
      K_tete = np.ascontiguousarray(self.kernel(X_test, X_test), dtype=np.float64)
      if uncert_form == 'covar' or not self.kernel.is_guaranteed_psd():
        mu_raw, covar = fitted.predict_covar_gram(K_tetr, K_tete)
        covar = self._post_covar_from_raw(covar)                     # gp_core.py:182-183
        return test_mean + mu_raw, (covar if uncert_form == 'covar' else np.sqrt(np.diag(covar)))
      mu_raw, sd = fitted.predict_gram(K_tetr, np.diag(K_tete).copy())
      return test_mean + mu_raw, sd
    if uncert_form == 'covar':
      mu_raw, covar = fitted.predict_covar(Xt)
      return test_mean + mu_raw, covar
    mu_raw, sd = fitted.predict(Xt, want_std=(uncert_form == 'std'))
    return test_mean + mu_raw, sd

  def eval_with_hallucinated_observations(self, X_test, X_halluc, uncert_form='none'):
    """ gp_core.py:192-220: mean from the real data, uncertainty from the GP augmented with the
        in-progress points. """
    pred_mean, _ = self.eval(X_test, uncert_form='none')
    if uncert_form == 'none':
      return (pred_mean, None)
    if uncert_form not in ('std', 'covar'):
      raise ValueError('uncert_form should be none, covar or std.')
    if self.num_tr_data == 0:
      raise NotImplementedError('Hallucinated observations need at least one real observation.')
    fitted = self._need_fit()
    if self._generic:
      return (pred_mean, self._generic_hallucinated_uncert(X_test, X_halluc, uncert_form))
    Xt = _as_2d_array(X_test)
    Xh = _as_2d_array(X_halluc)
    if uncert_form == 'covar':
      _, covar = fitted.predict_covar(Xt, X_halluc=Xh)
      return (pred_mean, covar)
    _, sd = fitted.predict(Xt, want_std=True, X_halluc=Xh)
    return (pred_mean, sd)

  def _post_covar_from_raw(self, raw_post_covar):
    """ get_post_covar_from_raw_covar (gp_core.py:849-857): the posterior covariance of a kernel that
        is not guaranteed PSD is projected onto {eigenvalues >= 0.05 noise_var} -- on the device. """
    if self.kernel.is_guaranteed_psd():
      return raw_post_covar
    return get_engine().project_psd(raw_post_covar, epsilon=0.05 * self.noise_var)

  def _generic_hallucinated_uncert(self, X_test, X_halluc, uncert_form):
    """ gp_core.py:199-220 for a caller-evaluated kernel: the augmented factor is assembled block-
        wise exactly as the reference does, every solve / factorisation on the device. """
    from .general_utils import solve_lower_triangular, stable_cholesky
    if self.handle_non_psd_kernels != 'guaranteed_psd':
      # the whole augmented matrix takes the projection branch (gp_core.py:199-206), then :207-213
      X_aug = list(self.X) + list(X_halluc)
      K_haha = self.kernel(X_halluc, X_halluc)
      K_trha = self.kernel(self.X, X_halluc)
      aug_K = np.vstack((np.hstack((self.K_trtr_wo_noise, K_trha)), np.hstack((K_trha.T, K_haha))))
      aug = get_engine().gp_fit_gram(aug_K, np.zeros(len(X_aug)), self.noise_var,
                                     handle_non_psd_kernels=self.handle_non_psd_kernels)
      _, covar = aug.predict_covar_gram(np.ascontiguousarray(self.kernel(X_test, X_aug), dtype=np.float64),
                                        np.ascontiguousarray(self.kernel(X_test, X_test), dtype=np.float64))
      aug.free()
      covar = self._post_covar_from_raw(covar)
      return covar if uncert_form == 'covar' else np.sqrt(np.diag(covar))
    K_haltr = self.kernel(X_halluc, self.X)                       # gp_core.py:201
    L = self.L
    B = solve_lower_triangular(L, K_haltr.T).T                    # q x n
    K_halhal = self.kernel(X_halluc, X_halluc) + self.noise_var * np.eye(len(X_halluc))
    Lh = stable_cholesky(K_halhal - B.dot(B.T))
    n, q = self.num_tr_data, len(X_halluc)
    L_aug = np.zeros((n + q, n + q))
    L_aug[:n, :n], L_aug[n:, :n], L_aug[n:, n:] = L, B, Lh
    K_aug_te = np.vstack([self.kernel(self.X, X_test), self.kernel(X_halluc, X_test)])
    V = solve_lower_triangular(L_aug, K_aug_te)
    covar = self.kernel(X_test, X_test) - V.T.dot(V)
    return covar if uncert_form == 'covar' else np.sqrt(np.diag(covar))

  def compute_log_marginal_likelihood(self):
    """ gp_core.py:222-227 (evaluated on the device together with the fit). """
    if self.num_tr_data == 0:
      return -0.0
    return self._need_fit().lml

  def __str__(self):
    return '%s, noise-var=%0.3f (n=%d)'%(self._child_str(), self.noise_var, len(self.Y))

  def _child_str(self):
    raise NotImplementedError('Implement in child class. !')

  # -- sampling --------------------------------------------------------------------------------
  def draw_samples(self, num_samples, X_test=None, mean_vals=None, covar=None):
    """ gp_core.py:250-254.  A single joint draw at X_test runs fused on the device
        (covariance, stable_cholesky and L u never leave HBM); the standard normals continue
        the global np.random state exactly as draw_gaussian_samples' np.random.normal call does
        (Engine.random_normals: generated on the device, bit for bit). """
    if X_test is not None and num_samples == 1 and self.num_tr_data > 0 and not self._generic:
      Xt = _as_2d_array(X_test)
      test_mean = self.mean_func(X_test)
      fit = self._need_fit()
      draw = getattr(fit.engine, 'random_normals', None)      # the stand-in engine of the CPU tests has none
      if draw is not None and DEVICE_NORMALS:
        U = draw(len(Xt))                                       # np.random.normal(size=(m, 1)), in HBM
      else:
        U = np.random.normal(size=(len(Xt), 1)).ravel()
      _, _, samples, _ = fit.thompson(Xt, U, block=len(Xt), mean_vals=test_mean, return_samples=True)
      return samples.reshape((1, -1))
    if X_test is not None:
      mean_vals, covar = self.eval(X_test, 'covar')
    return draw_gaussian_samples(num_samples, mean_vals, covar)

  def draw_samples_with_hallucinated_observations(self, num_samples, X_test, X_halluc):
    """ gp_core.py:256-261 """
    mean_vals, aug_covar = self.eval_with_hallucinated_observations(X_test, X_halluc,
                                                                    uncert_form='covar')
    return draw_gaussian_samples(num_samples, mean_vals, aug_covar)
