"""TEST INFRASTRUCTURE ONLY -- never imported by the product.

A stand-in for dragonfly_amd.engine.Engine / FittedGP whose arithmetic is oracle/ref_numpy.py.  It
exists for ONE purpose: the CPU-side plumbing test of the drop-in seams (SURVEY.md section 8d,
config C1: "null/NumPy backend must reproduce x_next bit-for-bit under the same seed") -- the real
reference optimiser runs with dragonfly_amd.install() in the build container, where there is no
GPU, and must take exactly the decisions it takes without it.  The product has no CPU path: the
real Engine raises without libdfhip.so / a gfx950 device (tests/test_abi.py)."""
import numpy as np

from oracle import ref_numpy as O


def to_oracle_spec(spec):
  """ dragonfly_amd.engine.KernelSpec -> oracle KernelSpec """
  if spec.kind in ('se', 'matern', 'poly', 'expdecay'):
    return O.KernelSpec(spec.kind, spec.dim, spec.scale, spec.bandwidths, nu=spec.nu)
  subs = [O.KernelSpec(kind, len(grp), sc, bw, nu=nu)
          for kind, grp, sc, nu, bw in zip(spec.sub_kinds, spec.groups, spec.sub_scales, spec.sub_nus,
                                           spec.sub_bandwidths)]
  groups = [list(g) for g in spec.groups]
  if getattr(spec, 'group_factors', None) is None:
    return O.KernelSpec(spec.kind, spec.dim, spec.scale, groups=groups, subs=subs)
  # a product with additive factors (struct dfh_kernel_desc: group_factor / factor_is_sum / factor_scale):
  # the additive factor becomes an additive kernel over the union of its groups' columns
  f_groups, f_subs = [], []
  for f, is_sum in enumerate(spec.factor_sums):
    members = [g for g, ff in enumerate(spec.group_factors) if ff == f]
    if not is_sum:
      f_groups.append(groups[members[0]]); f_subs.append(subs[members[0]])
      continue
    cols = [c for g in members for c in groups[g]]
    local = [[cols.index(c) for c in groups[g]] for g in members]
    f_groups.append(cols)
    f_subs.append(O.KernelSpec('additive', len(cols), spec.factor_scales[f], groups=local, subs=[subs[g] for g in members]))
  return O.KernelSpec('product', spec.dim, spec.scale, groups=f_groups, subs=f_subs)


class OracleFittedGP(object):
  """ FittedGP's interface over GPOracle (zero mean: the mirrors centre Y and add the mean). """

  def __init__(self, engine, spec, X, y_centred, noise_var, gram=None, handle_non_psd_kernels='guaranteed_psd'):
    self.engine = engine
    X = np.asarray(X, dtype=np.float64)
    kernel = (lambda A, B=None: gram) if gram is not None else to_oracle_spec(spec)
    self.oracle = O.GPOracle(X, np.asarray(y_centred, dtype=np.float64), kernel, 0.0, noise_var)
    if handle_non_psd_kernels != 'guaranteed_psd':      # _get_cholesky_decomp's other branches
      og = self.oracle
      og.L = O.get_cholesky_decomp(og.K_trtr_wo_noise, noise_var, handle_non_psd_kernels)
      og.jitter_power = None
      og.alpha = O.solve_upper_triangular(og.L.T, O.solve_lower_triangular(og.L, og.Y))
    self.spec, self.n, self.d = spec, len(X), (X.shape[1] if X.ndim == 2 else 0)
    self.lml = float(self.oracle.lml())
    self.jitter_power = self.oracle.jitter_power

  def free(self):
    pass

  def get_L(self):
    return self.oracle.L

  def get_alpha(self):
    return self.oracle.alpha

  def get_K(self):
    return self.oracle.K_trtr_wo_noise

  def predict(self, Xs, want_std=True, X_halluc=None):
    Xs = np.asarray(Xs, dtype=np.float64)
    if not want_std:
      return self.oracle.eval(Xs, 'none')
    if X_halluc is not None and len(X_halluc) > 0:
      return self.oracle.eval_with_hallucinated_observations(Xs, X_halluc, 'std')
    return self.oracle.eval(Xs, 'std')

  def predict_covar(self, Xs, X_halluc=None):
    if X_halluc is not None and len(X_halluc) > 0:
      return self.oracle.eval_with_hallucinated_observations(Xs, X_halluc, 'covar')
    return self.oracle.eval(np.asarray(Xs, dtype=np.float64), 'covar')

  def predict_gram(self, K_cross, k_ss=None, mean_const=0.0, mean_vals=None):
    K_cross = np.asarray(K_cross, dtype=np.float64)
    mu = K_cross.dot(self.oracle.alpha) + (mean_const if mean_vals is None else np.asarray(mean_vals))
    if k_ss is None:
      return mu, None
    V = O.solve_lower_triangular(self.oracle.L, K_cross.T)
    return mu, np.sqrt(np.asarray(k_ss) - np.diag(V.T.dot(V)))       # gp_core.py:180-187

  def predict_covar_gram(self, K_cross, K_tete):
    K_cross = np.asarray(K_cross, dtype=np.float64)
    V = O.solve_lower_triangular(self.oracle.L, K_cross.T)
    return K_cross.dot(self.oracle.alpha), np.asarray(K_tete) - V.T.dot(V)

  def acq_argmax(self, acq, Xs, params=(0.0, 0.0), mean_const=0.0, mean_vals=None, X_halluc=None,
                 return_vals=False):
    mu, sd = self.predict(Xs, True, X_halluc)
    mu = mu + (mean_const if mean_vals is None else np.asarray(mean_vals))
    vals = O.acq_values(acq, mu, sd, params[0], params[1] if len(params) > 1 else 0.0)
    best_val, best_idx = O.argmax_first(vals)
    return (best_val, best_idx, vals) if return_vals else (best_val, best_idx)

  def thompson(self, Xs, U, block=4096, mean_const=0.0, mean_vals=None, return_samples=False):
    shift = mean_const if mean_vals is None else np.asarray(mean_vals)
    samples = self.oracle.draw_samples_blocked(Xs, np.ravel(U), block) + shift
    best_val, best_idx = O.argmax_first(samples)
    return (best_val, best_idx, samples, None) if return_samples else (best_val, best_idx)

  def add_ucb_group(self, group, beta, Xg, return_vals=False):
    # O.add_ucb_group_values computes beta from the time step; here beta is given
    add = self.oracle.kernel
    kern_j, grp = add.subs[group], add.groups[group]
    K_tetr = add.scale * kern_j(Xg, self.oracle.X[:, grp])
    mu = K_tetr.dot(self.oracle.alpha) + np.array([0] * len(Xg))
    V = O.solve_lower_triangular(self.oracle.L, K_tetr.T)
    covar = add.scale * kern_j(Xg, Xg) - V.T.dot(V)
    vals = mu + beta * np.sqrt(np.diag(covar))
    best_val, best_idx = O.argmax_first(vals)
    return (best_val, best_idx, vals) if return_vals else (best_val, best_idx)

  def add_ucb_all(self, betas, cands_per_group, return_vals=False, sizes=None):
    res = [self.add_ucb_group(g, betas[g], np.asarray(c), True) for g, c in enumerate(cands_per_group)]
    out = (np.array([r[0] for r in res]), np.array([r[1] for r in res]))
    return out + ([r[2] for r in res],) if return_vals else out

  def append(self, X_new, y_centred_all, allow_jitter=True):
    X_all = np.vstack([self.oracle.X, np.asarray(X_new, dtype=np.float64)])
    return OracleFittedGP(self.engine, self.spec, X_all, y_centred_all, self.oracle.noise_var)


class OracleEngine(object):
  """ The part of Engine's interface the mirrors use with host-generated candidates. """

  def to_device(self, host):
    return np.array(host, dtype=np.float64)

  def gp_fit(self, spec, X, y_centred, noise_var, allow_jitter=True):
    return OracleFittedGP(self, spec, X, y_centred, noise_var)

  def gp_fit_gram(self, K, y_centred, noise_var, allow_jitter=True, handle_non_psd_kernels='guaranteed_psd'):
    return OracleFittedGP(self, None, np.zeros((len(K), 0)), y_centred, noise_var, gram=np.asarray(K),
                          handle_non_psd_kernels=handle_non_psd_kernels)

  def project_psd(self, M, epsilon=0.0):
    return O.project_symmetric_to_psd_cone(np.asarray(M, dtype=np.float64), epsilon=epsilon)

  def kernel_matrix(self, spec, X1, X2=None, diag_add=0.0, out=None):
    K = to_oracle_spec(spec)(np.asarray(X1, dtype=np.float64), None if X2 is None else np.asarray(X2, dtype=np.float64))
    return K + diag_add * np.eye(len(K)) if diag_add else K

  def dist_squared(self, X1, X2):
    return O.dist_squared(np.asarray(X1, dtype=np.float64), np.asarray(X2, dtype=np.float64))

  def gemm(self, A, B, C_in=None, alpha=1.0, beta=0.0, transb=False, lower_only=False):
    # the engine's convention: B is [N x K] (C = A B^T) unless transb, then B is [K x N]
    prod = alpha * np.asarray(A).dot(np.asarray(B) if transb else np.asarray(B).T)
    return prod if C_in is None else prod + beta * C_in

  def cholesky(self, M):
    return np.linalg.cholesky(np.asarray(M))

  def stable_cholesky(self, M, return_power=False):
    return O.stable_cholesky(np.asarray(M), return_power=return_power)

  def solve_triangular(self, L_lower, b, upper=False):
    return O.solve_upper_triangular(L_lower.T, b) if upper else O.solve_lower_triangular(L_lower, b)

  lml_batch_sizes = None     # set to a list to record the batch sizes

  def gp_lml_batch(self, specs, X, y, mean_consts, noise_vars, allow_jitter=True, return_powers=False):
    if self.lml_batch_sizes is not None:
      self.lml_batch_sizes.append(len(specs))
    y = np.asarray(y, dtype=np.float64)
    return np.array([OracleFittedGP(self, s, X, y - c, nv).lml for s, c, nv in zip(specs, mean_consts, noise_vars)])


def patch_engine(monkeypatch):
  """ Point every mirror module's get_engine at one OracleEngine and keep candidates on the host. """
  from dragonfly_amd import euclidean_gp, general_utils, gp_core, gpb_acquisitions, kernel
  from dragonfly_amd import engine as engine_mod
  eng = OracleEngine()
  for mod in (engine_mod, euclidean_gp, general_utils, gp_core, kernel):
    monkeypatch.setattr(mod, 'get_engine', lambda _e=eng: _e)
  monkeypatch.setattr(gpb_acquisitions, 'DEVICE_CANDIDATES', False)
  return eng
