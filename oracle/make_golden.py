"""Generates tests/golden/*.npz by running the REAL reference (dragonfly 0.1.7 imported from
/root/reference, unmodified) on small seeded inputs.

    python oracle/make_golden.py            # needs /root/reference; run in the build container

The reference cannot travel to the GPU box, so its outputs are committed as fixtures.  They pin
  - the NumPy oracle (oracle/ref_numpy.py) in tests/test_oracle_golden.py  (CPU), and
  - the HIP engine (dragonfly_amd) in tests/test_gpu_golden.py              (MI355X),
for the results the reference's own unit tests do not pin: posterior mean / std / covariance,
alpha, L, lml, UCB / EI / PI / TTEI / TS / add-UCB choices, hallucinated std, fitter choices.

The only change to the reference's environment is the NumPy-2 compatibility shim below
(SURVEY.md section 8c): attributes NumPy removed after the reference was written.
"""
import math
import os
import sys
import warnings
from argparse import Namespace

import numpy as np

REF = os.environ.get('DRAGONFLY_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
if os.path.dirname(HERE) not in sys.path:       # `python oracle/make_golden.py` from anywhere: oracle.* importable
  sys.path.insert(0, os.path.dirname(HERE))


def import_reference():
  """ NumPy-2 shim + import. Never edits the reference. """
  np.math = math
  np.asscalar = lambda a: np.asarray(a).item()
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    for nm, t in (('object', object), ('int', int), ('float', float), ('bool', bool)):
      if nm not in np.__dict__:
        setattr(np, nm, t)
  if REF not in sys.path:
    sys.path.insert(0, REF)
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    import dragonfly  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
  return dragonfly


def test_functions():
  """ Synthetic objectives (in the unit cube, like Dragonfly's normalised domains). """
  return {
    'sin': lambda X: np.sin(3 * X.sum(axis=1)),
    'quad': lambda X: (X ** 2).dot((np.arange(X.shape[1]) + 1.0) / X.shape[1]),
  }


CASES = [
  # name, kernel kind, n, d, m, function, extra
  dict(name='se_d2_n40', kind='se', n=40, d=2, m=37, fn='sin'),
  dict(name='se_ard_d5_n50', kind='se', n=50, d=5, m=41, fn='quad', ard=True),
  dict(name='matern25_d6_n60', kind='matern', nu=2.5, n=60, d=6, m=33, fn='sin'),
  dict(name='matern15_d3_n45', kind='matern', nu=1.5, n=45, d=3, m=29, fn='quad'),
  dict(name='matern05_d2_n30', kind='matern', nu=0.5, n=30, d=2, m=25, fn='sin'),
  dict(name='se_d32_n130', kind='se', n=130, d=32, m=70, fn='quad', ard=True),
  dict(name='additive_d10_n80', kind='additive', n=80, d=10, m=40, fn='quad', group_size=3),
]


def build_reference_kernel(case, Y, rs):
  from dragonfly.gp import kernel as K
  d = case['d']
  scale = float(Y.var())
  if case['kind'] == 'additive':
    perm = list(rs.permutation(d))
    gs = case['group_size']
    groups = [perm[i:i + gs] for i in range(0, d, gs)]
    bws = [0.3 + 0.5 * rs.rand(len(g)) for g in groups]
    kinds = ['se' if i % 2 == 0 else 'matern' for i in range(len(groups))]
    subs = []
    for g, bw, kd in zip(groups, bws, kinds):
      subs.append(K.SEKernel(len(g), 1.0, bw) if kd == 'se' else K.MaternKernel(len(g), 2.5, 1.0, bw))
    kern = K.AdditiveKernel(scale, subs, groups)
    meta = dict(groups=np.array([list(map(int, g)) + [-1] * (gs - len(g)) for g in groups]),
                sub_bws=np.array([list(b) + [0.0] * (gs - len(b)) for b in bws]),
                sub_kinds=np.array([0 if kd == 'se' else 1 for kd in kinds]), scale=scale)
    return kern, meta
  bw = (0.2 * np.sqrt(d) * (0.5 + np.arange(d) / float(d))) if case.get('ard') else np.full(d, 0.25 * np.sqrt(d))
  if case['kind'] == 'se':
    return K.SEKernel(d, scale, bw), dict(scale=scale, bw=bw)
  return K.MaternKernel(d, case['nu'], scale, bw), dict(scale=scale, bw=bw, nu=case['nu'])


def gen_gp_cases():
  from dragonfly.gp.gp_core import GP
  from dragonfly.opt import gpb_acquisitions as A
  from dragonfly.exd.domains import EuclideanDomain
  fns = test_functions()
  for ci, case in enumerate(CASES):
    rs = np.random.RandomState(1000 + ci)
    n, d, m = case['n'], case['d'], case['m']
    X = rs.random_sample((n, d))
    Y = fns[case['fn']](X) + 0.05 * rs.randn(n)
    kern, meta = build_reference_kernel(case, Y, rs)
    mean_c = float(np.median(Y))
    noise = float(Y.var() / 20)
    mean_func = lambda x, _c=mean_c: np.array([_c] * len(x))
    gp = GP(list(X), list(Y), kern, mean_func, noise)
    Xs = rs.random_sample((m, d))
    mu, sd = gp.eval(Xs, 'std')
    _, cov = gp.eval(Xs, 'covar')
    Xh = rs.random_sample((3, d))
    _, sd_h = gp.eval_with_hallucinated_observations(Xs, list(Xh), 'std')
    out = dict(X=X, Y=Y, Xs=Xs, Xh=Xh, mean_c=mean_c, noise=noise,
               K=gp.K_trtr_wo_noise, L=gp.L, alpha=gp.alpha,
               lml=gp.compute_log_marginal_likelihood(), mu=mu, sd=sd, cov=cov, sd_h=sd_h)
    for k, v in meta.items():
      out['kern_' + k] = v
    # acquisitions through the reference's own callables, 'rand' maximiser, seeded global RNG
    bounds = np.array([[0.0, 1.0]] * d)
    best = float(Y.max())
    def anc(max_evals, in_progress=()):
      return Namespace(max_evals=max_evals, t=n, domain=EuclideanDomain(bounds),
                       curr_max_val=best, eval_points_in_progress=list(in_progress),
                       acq_opt_method='rand', handle_parallel='halluc', is_mf=False,
                       domain_bounds=bounds)
    for ai, acq in enumerate(['ucb', 'ei', 'pi', 'ttei', 'ts']):
      np.random.seed(5000 + 10 * ci + ai)
      out['asy_' + acq] = getattr(A.asy, acq)(gp, anc(64))
    np.random.seed(6000 + ci)
    out['asy_ucb_halluc'] = A.asy.ucb(gp, anc(64, in_progress=[Xh[0], Xh[1]]))
    np.random.seed(7000 + ci)
    out['syn_ei_3'] = np.array(A.syn.ei(3, gp, anc(48)))
    if case['kind'] == 'additive':
      np.random.seed(8000 + ci)
      out['asy_add_ucb'] = A.asy.add_ucb(gp, anc(120))
    # acquisition values on Xs straight from the reference formulas
    beta = A._get_ucb_beta_th(A._get_gp_ucb_dim(gp), n)            # pylint: disable=protected-access
    out['beta_th'] = beta
    out['val_ucb'] = mu + beta * sd
    nd = (mu - best) / sd
    out['val_ei'] = sd * A._expected_improvement_for_norm_diff(nd)  # pylint: disable=protected-access
    out['val_pi'] = A.normal_distro.cdf(nd)
    comb = np.sqrt(0.3 ** 2 + sd ** 2)
    out['val_ttei'] = comb * A._expected_improvement_for_norm_diff((mu - best) / comb)  # pylint: disable=protected-access
    # joint TS draw with recorded normals
    np.random.seed(9000 + ci)
    state_U = np.random.RandomState(9000 + ci).normal(size=(m, 1))
    out['ts_U'] = state_U.ravel()
    out['ts_sample'] = gp.draw_samples(1, Xs).ravel()      # consumes the same normals from np.random
    np.savez_compressed(os.path.join(OUT, 'gp_' + case['name'] + '.npz'), **out)
    print('wrote gp_%s' % case['name'])


def gen_fitter_case():
  """ EuclideanGPFitter, ML by random search, seeded. """
  from dragonfly.gp.euclidean_gp import EuclideanGPFitter
  rs = np.random.RandomState(77)
  n, d = 45, 3
  X = rs.random_sample((n, d))
  Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
  res = {}
  for kt in ('se', 'matern'):
    opts = Namespace(kernel_type=kt, ml_hp_tune_opt='rand', hp_tune_max_evals=60, hp_tune_criterion='ml')
    np.random.seed(4242)
    fitter = EuclideanGPFitter(list(X), list(Y), options=opts)
    _, gp, hps = fitter.fit_gp()
    res[kt + '_cts_hps'] = np.array(hps[0], dtype=float)
    res[kt + '_lml'] = gp.compute_log_marginal_likelihood()
    res[kt + '_noise'] = gp.noise_var
    res[kt + '_scale'] = gp.kernel.hyperparams['scale']
    res[kt + '_bw'] = np.asarray(gp.kernel.hyperparams['dim_bandwidths'], dtype=float)
    Xs = rs.random_sample((20, d))
    mu, sd = gp.eval(Xs, 'std')
    res[kt + '_Xs'], res[kt + '_mu'], res[kt + '_sd'] = Xs, mu, sd
  np.savez_compressed(os.path.join(OUT, 'fitter_d3_n45.npz'), X=X, Y=Y, **res)
  print('wrote fitter_d3_n45')


POST_SAMPLING_CASES = {
  # name: (fitter options, num_samples)
  'se_one': (dict(kernel_type='se', hp_tune_criterion='post_sampling', post_hp_tune_burn=6), 1),
  'matern_nu_three': (dict(kernel_type='matern', matern_nu=-1.0, hp_tune_criterion='post_sampling',
                           post_hp_tune_burn=4, post_hp_tune_offset=3), 3),
  'additive_two': (dict(kernel_type='se', use_additive_gp=True, add_max_group_size=3, hp_tune_criterion='post_sampling',
                        post_hp_tune_burn=3, post_hp_tune_offset=2, mean_func_type='median'), 2),
}
ADD_REXP_OPTS = dict(kernel_type='se', use_additive_gp=True, add_max_group_size=3, ml_hp_tune_opt='rand_exp_sampling',
                     hp_tune_max_evals=25, hp_tune_criterion='ml')


def gen_post_sampling_cases():
  """ The reference's EuclideanGPFitter with hp_tune_criterion='post_sampling' (gp_core.py:592-726: slice
      sampling of the continuous hyper-parameters, Metropolis on the discrete ones and on the additive
      grouping's seed) and its additive rand_exp_sampling (euclidean_gp.py:748-775), seeded: the sampled
      hyper-parameters, groupings and the state of the random stream afterwards. """
  from dragonfly.gp.euclidean_gp import EuclideanGPFitter
  rs = np.random.RandomState(91)
  n, d = 30, 4
  X = rs.random_sample((n, d))
  Y = np.sin(3 * X[:, 0] + X[:, 1]) + np.cos(2 * X[:, 2]) + 0.05 * rs.randn(n)
  res = {}
  for name, (opts, num) in sorted(POST_SAMPLING_CASES.items()):
    np.random.seed(2718)
    fitter = EuclideanGPFitter(list(X), list(Y), options=Namespace(**opts))
    ret = fitter.fit_gp(num, 'post_sampling')
    if num == 1:
      kind, gp, hps = ret
      cts, dscr = hps
      res[name + '_lml'] = gp.compute_log_marginal_likelihood()
      others = [None]
    else:
      kind, cts, dscr, others = ret
    res[name + '_kind'] = np.array(kind)
    res[name + '_cts'] = np.array(cts, dtype=float)
    res[name + '_dscr'] = np.array(dscr, dtype=float)
    for t, o in enumerate(others):
      if o is not None and o.add_gp_groupings is not None:
        res[name + '_grouping_%d' % t] = np.array(sum([list(g) + [-1] for g in o.add_gp_groupings], []), dtype=float)
    res[name + '_rand_after'] = np.random.random()
    print('post_sampling %s: %s cts %s dscr %s' % (name, kind, np.round(res[name + '_cts'], 3).tolist(), res[name + '_dscr'].tolist()))
  np.random.seed(31415)
  fitter = EuclideanGPFitter(list(X), list(Y), options=Namespace(**ADD_REXP_OPTS))
  kind, cts, dscr, others, probs = fitter.fit_gp()
  res['add_rexp_kind'] = np.array(kind)
  res['add_rexp_cts'] = np.array(cts, dtype=float)
  res['add_rexp_dscr'] = np.array(dscr, dtype=float)
  res['add_rexp_probs'] = np.array(probs, dtype=float)
  res['add_rexp_first_grouping'] = np.array(sum([list(g) + [-1] for g in others[0].add_gp_groupings], []), dtype=float)
  res['add_rexp_rand_after'] = np.random.random()
  np.savez_compressed(os.path.join(OUT, 'post_sampling_d4_n30.npz'), X=X, Y=Y, **res)
  print('wrote post_sampling_d4_n30')


def gen_c1_case():
  """ BASELINE config 1: Branin 2-D, n = 200, SE kernel, UCB over 1000 random candidates, driven
      through the reference's own optimiser objects in ask/tell mode. """
  from dragonfly.opt import gp_bandit
  from dragonfly.exd.domains import EuclideanDomain
  from dragonfly.exd.experiment_caller import EuclideanFunctionCaller
  from dragonfly.utils.option_handler import load_options
  from dragonfly.utils.euclidean_synthetic_functions import get_mf_branin_function
  branin_function = get_mf_branin_function(1)[1]     # euclidean_synthetic_functions.py:108-149
  bounds = [[-5, 10], [0, 15]]
  opts = load_options(gp_bandit.get_all_euc_gp_bandit_args())
  opts.kernel_type = 'se'
  opts.acq = 'ucb'
  opts.acq_opt_method = 'rand'
  opts.acq_opt_max_evals = 1000
  opts.gpb_hp_tune_criterion = 'ml'
  opts.gpb_ml_hp_tune_opt = 'rand'
  opts.hp_tune_max_evals = 50
  np.random.seed(101)
  func_caller = EuclideanFunctionCaller(None, EuclideanDomain(bounds))
  opt = gp_bandit.EuclideanGPBandit(func_caller, ask_tell_mode=True, options=opts, reporter='silent')
  opt.initialise()
  Xraw = np.random.RandomState(101).random_sample((200, 2)) * np.array([15.0, 15.0]) + np.array([-5.0, 0.0])
  Yv = np.array([branin_function(x) for x in Xraw])
  opt.tell([(x, y) for x, y in zip(Xraw, Yv)])
  opt.first_qinfos = []
  # record exactly what the single acquisition call sees
  record = {}
  from dragonfly.opt import gpb_acquisitions as A
  orig = A.asy.ucb
  def spy(gp, anc_data):
    st = np.random.get_state()
    record['gp'] = gp
    record['anc'] = anc_data
    record['state'] = st
    return orig(gp, anc_data)
  A.asy.ucb = spy
  try:
    x_next = opt.ask()
  finally:
    A.asy.ucb = orig
  gp = record['gp']
  anc = record['anc']
  # replay to capture the candidates and values
  np.random.set_state(record['state'])
  cands = np.random.random((int(anc.max_evals), 2)) * (anc.domain.bounds[:, 1] - anc.domain.bounds[:, 0]) + anc.domain.bounds[:, 0]
  mu, sd = gp.eval(cands, 'std')
  beta = A._get_ucb_beta_th(A._get_gp_ucb_dim(gp), anc.t)   # pylint: disable=protected-access
  vals = mu + beta * sd
  np.savez_compressed(os.path.join(OUT, 'c1_branin.npz'), Xn=np.array(gp.X), Yn=np.array(gp.Y),
                      scale=gp.kernel.hyperparams['scale'],
                      bw=np.asarray(gp.kernel.hyperparams['dim_bandwidths'], dtype=float),
                      noise=gp.noise_var, mean_c=float(gp.mean_func([np.zeros(2)])[0]),
                      cands=cands, t=anc.t, beta=beta, vals=vals, argmax=int(vals.argmax()),
                      x_next_normalised=cands[int(vals.argmax())], x_next_raw=np.array(x_next),
                      L=gp.L, alpha=gp.alpha, max_evals=int(anc.max_evals))
  print('wrote c1_branin (x_next %s)' % (x_next,))


def gen_mfgp_case():
  """ EuclideanMFGP (euclidean_gp.py:347-412) with the coordinate-product kernel
      scale * SE(z) * Matern-2.5(x): posterior, hallucinated std, a joint sample. """
  from dragonfly.gp.euclidean_gp import EuclideanMFGP
  from dragonfly.gp import kernel as rk
  rs = np.random.RandomState(909)
  n, fd, dd, m = 70, 1, 3, 31
  ZZ, XX = rs.random_sample((n, fd)), rs.random_sample((n, dd))
  YY = np.sin(3 * XX.sum(axis=1)) * (0.5 + ZZ[:, 0]) + 0.05 * rs.randn(n)
  fbw, dbw = np.array([0.6]), np.array([0.35, 0.5, 0.8])
  scale, noise, mean_c = 1.7, float(YY.var() / 20), float(np.median(YY))
  fidel_kernel = rk.SEKernel(fd, 1.0, fbw)
  domain_kernel = rk.MaternKernel(dd, 2.5, 1.0, dbw)
  mean_func = lambda x: np.array([mean_c] * len(x))
  gp = EuclideanMFGP(list(ZZ), list(XX), list(YY), None, scale, fidel_kernel, domain_kernel,
                     mean_func, noise)
  Zs, Xs = rs.random_sample((m, fd)), rs.random_sample((m, dd))
  mu, sd = gp.eval_at_fidel(list(Zs), list(Xs), 'std')
  _, cov = gp.eval_at_fidel(list(Zs), list(Xs), 'covar')
  Zh, Xh = rs.random_sample((4, fd)), rs.random_sample((4, dd))
  _, sdh = gp.eval_at_fidel_with_hallucinated_observations(list(Zs), list(Xs), list(Zh), list(Xh), 'std')
  np.savez_compressed(os.path.join(OUT, 'mfgp_f1_d3_n70.npz'), ZZ=ZZ, XX=XX, YY=YY, fbw=fbw, dbw=dbw,
                      scale=scale, noise=noise, mean_c=mean_c, K=gp.K_trtr_wo_noise, L=gp.L,
                      alpha=gp.alpha, lml=gp.compute_log_marginal_likelihood(), Zs=Zs, Xs=Xs, mu=mu,
                      sd=sd, cov=cov, Zh=Zh, Xh=Xh, sdh=sdh)
  print('wrote mfgp_f1_d3_n70')


def gen_poly_expdecay_cases():
  """ The two kernels that are not stationary (kernel.py:331-437): their matrices, a plain GP
      with a polynomial kernel, and the multi-fidelity GP the reference's MF fitter builds for
      fidel_kernel_type='expdecay' (euclidean_gp.py:881-887, 707): scale * ExpDecay(z) * SE(x);
      posterior mean / std / covariance, hallucinated std, lml, a joint sample. """
  from dragonfly.gp.euclidean_gp import EuclideanMFGP
  from dragonfly.gp.gp_core import GP
  from dragonfly.gp import kernel as rk
  from dragonfly.utils.general_utils import draw_gaussian_samples
  rs = np.random.RandomState(1311)
  res = {}
  # kernel matrices
  X1, X2 = rs.random_sample((40, 3)) - 0.3, rs.random_sample((27, 3)) - 0.3
  scalings = np.array([0.8, 1.3, 0.45])
  for order in (1, 2, 3, 5):
    kern = rk.PolyKernel(3, order, 1.7, scalings)
    res['poly%d_K12' % order] = kern(X1, X2)
    res['poly%d_K11' % order] = kern(X1)
  Z1, Z2 = rs.random_sample((33, 2)), rs.random_sample((21, 2))
  powers = np.array([1.6, 0.7])
  ed = rk.ExpDecayKernel(2, 1.3, 0.21, powers)
  res.update(X1=X1, X2=X2, scalings=scalings, Z1=Z1, Z2=Z2, powers=powers, ed_scale=1.3, ed_offset=0.21,
             ed_K12=ed(Z1, Z2), ed_K11=ed(Z1))
  # a GP with the polynomial kernel
  n, m = 50, 29
  Xp = rs.random_sample((n, 3)) - 0.5
  Yp = (Xp.sum(axis=1)) ** 2 - Xp[:, 0] + 0.03 * rs.randn(n)
  pk = rk.PolyKernel(3, 3, 0.9, np.array([1.1, 0.7, 0.9]))
  p_noise, p_mean = float(Yp.var() / 15), float(np.median(Yp))
  pgp = GP(list(Xp), list(Yp), pk, lambda x: np.array([p_mean] * len(x)), p_noise)
  Xps = rs.random_sample((m, 3)) - 0.5
  p_mu, p_sd = pgp.eval(list(Xps), 'std')
  Xph = rs.random_sample((3, 3)) - 0.5
  _, p_sdh = pgp.eval_with_hallucinated_observations(list(Xps), list(Xph), 'std')
  res.update(p_X=Xp, p_Y=Yp, p_scalings=np.array([1.1, 0.7, 0.9]), p_order=3, p_scale=0.9, p_noise=p_noise,
             p_mean=p_mean, p_alpha=pgp.alpha, p_lml=pgp.compute_log_marginal_likelihood(), p_Xs=Xps, p_mu=p_mu,
             p_sd=p_sd, p_Xh=Xph, p_sdh=p_sdh)
  # the multi-fidelity GP with the exponential-decay fidelity kernel
  n, fd, dd, m = 64, 2, 3, 31
  ZZ, XX = rs.random_sample((n, fd)), rs.random_sample((n, dd))
  YY = np.sin(3 * XX.sum(axis=1)) * (1.0 - 0.4 / (1.0 + 3 * ZZ.sum(axis=1))) + 0.04 * rs.randn(n)
  f_powers, f_offset, dbw = np.array([1.2, 2.3]), 0.15, np.array([0.4, 0.55, 0.7])
  scale, noise, mean_c = 1.4, float(YY.var() / 20), float(np.median(YY))
  fidel_kernel = rk.ExpDecayKernel(fd, 1.0, f_offset, f_powers)
  domain_kernel = rk.SEKernel(dd, 1.0, dbw)
  gp = EuclideanMFGP(list(ZZ), list(XX), list(YY), None, scale, fidel_kernel, domain_kernel,
                     lambda x: np.array([mean_c] * len(x)), noise)
  Zs, Xs = rs.random_sample((m, fd)), rs.random_sample((m, dd))
  mu, sd = gp.eval_at_fidel(list(Zs), list(Xs), 'std')
  _, cov = gp.eval_at_fidel(list(Zs), list(Xs), 'covar')
  Zh, Xh = rs.random_sample((4, fd)), rs.random_sample((4, dd))
  _, sdh = gp.eval_at_fidel_with_hallucinated_observations(list(Zs), list(Xs), list(Zh), list(Xh), 'std')
  U = rs.randn(m)
  np.random.seed(77)
  sample = gp.draw_mf_samples(1, list(Zs), list(Xs)).ravel()
  np.random.seed(77)
  sample_normals = np.random.normal(size=(m, 1)).ravel()
  res.update(ZZ=ZZ, XX=XX, YY=YY, f_powers=f_powers, f_offset=f_offset, dbw=dbw, scale=scale, noise=noise,
             mean_c=mean_c, K=gp.K_trtr_wo_noise, L=gp.L, alpha=gp.alpha,
             lml=gp.compute_log_marginal_likelihood(), Zs=Zs, Xs=Xs, mu=mu, sd=sd, cov=cov, Zh=Zh, Xh=Xh,
             sdh=sdh, sample=sample, sample_normals=sample_normals, U=U)
  np.savez_compressed(os.path.join(OUT, 'poly_expdecay.npz'), **res)
  print('wrote poly_expdecay')


def gen_mf_fitter_case():
  """ EuclideanMFGPFitter (euclidean_gp.py:418-710), ML by random search, seeded: the chosen
      hyper-parameters and the fitted GP's predictions for three fidelity / domain kernel pairs,
      one of them with a tuned Matern nu (a discrete hyper-parameter). """
  from dragonfly.gp.euclidean_gp import EuclideanMFGPFitter
  rs = np.random.RandomState(2718)
  n, fd, dd = 48, 2, 3
  ZZ, XX = rs.random_sample((n, fd)), rs.random_sample((n, dd))
  YY = np.sin(3 * XX.sum(axis=1)) * (1.0 - 0.4 / (1.0 + 3 * ZZ.sum(axis=1))) + 0.04 * rs.randn(n)
  Zs, Xs = rs.random_sample((20, fd)), rs.random_sample((20, dd))
  res = dict(ZZ=ZZ, XX=XX, YY=YY, Zs=Zs, Xs=Xs)
  cases = [('se_se', dict(fidel_kernel_type='se', domain_kernel_type='se')),
           ('expdecay_se', dict(fidel_kernel_type='expdecay', domain_kernel_type='se')),
           ('matern_matern', dict(fidel_kernel_type='matern', domain_kernel_type='matern', fidel_matern_nu=-1.0,
                                  domain_matern_nu=1.5))]
  for name, kw in cases:
    opts = Namespace(ml_hp_tune_opt='rand', hp_tune_max_evals=60, hp_tune_criterion='ml', **kw)
    np.random.seed(1618)
    fitter = EuclideanMFGPFitter(list(ZZ), list(XX), list(YY), options=opts)
    _, gp, hps = fitter.fit_gp()
    res[name + '_cts_hps'] = np.array(hps[0], dtype=float)
    res[name + '_dscr_hps'] = np.array(hps[1], dtype=float)
    res[name + '_bounds'] = np.array(fitter.cts_hp_bounds, dtype=float)
    res[name + '_lml'] = gp.compute_log_marginal_likelihood()
    res[name + '_noise'] = gp.noise_var
    res[name + '_scale'] = gp.kernel.hyperparams['scale']
    mu, sd = gp.eval_at_fidel(list(Zs), list(Xs), 'std')
    res[name + '_mu'], res[name + '_sd'] = mu, sd
    res[name + '_rand_after'] = np.random.random()       # the global stream after the fit
  np.savez_compressed(os.path.join(OUT, 'mf_fitter_f2_d3_n48.npz'), **res)
  print('wrote mf_fitter_f2_d3_n48')


MF_ADDITIVE_OPTS = dict(ml_hp_tune_opt='rand', hp_tune_max_evals=40, hp_tune_criterion='ml', fidel_kernel_type='se',
                        domain_kernel_type='se', domain_use_additive_gp=True, domain_add_max_group_size=3,
                        domain_num_groups_per_group_size=2)


def gen_mf_fitter_additive_case():
  """ EuclideanMFGPFitter with an ADDITIVE domain model (euclidean_gp.py:480-486, 622-633, 696-707: the
      joint kernel is a CoordinateProductKernel whose second kernel is an AdditiveKernel), ML by
      random search over groupings and continuous hyper-parameters, seeded: the chosen
      hyper-parameters and grouping, the joint Gram matrix, the fitted GP's predictions. """
  from dragonfly.gp.euclidean_gp import EuclideanMFGPFitter
  rs = np.random.RandomState(314)
  n, fd, dd = 40, 1, 6
  ZZ, XX = rs.random_sample((n, fd)), rs.random_sample((n, dd))
  YY = (np.sin(3 * XX[:, :3].sum(axis=1)) + XX[:, 3:].sum(axis=1) ** 2) * (1.0 - 0.4 / (1.0 + 3 * ZZ.sum(axis=1))) \
       + 0.04 * rs.randn(n)
  Zs, Xs = rs.random_sample((25, fd)), rs.random_sample((25, dd))
  res = dict(ZZ=ZZ, XX=XX, YY=YY, Zs=Zs, Xs=Xs)
  np.random.seed(99)
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    fitter = EuclideanMFGPFitter(list(ZZ), list(XX), list(YY), options=Namespace(**MF_ADDITIVE_OPTS))
    _, gp, hps = fitter.fit_gp()
  dk = gp.kernel.kernel_list[1]
  assert type(dk).__name__ == 'AdditiveKernel'
  res['cts_hps'] = np.array(hps[0], dtype=float)
  res['dscr_hps'] = np.array(hps[1], dtype=float)
  res['bounds'] = np.array(fitter.cts_hp_bounds, dtype=float)
  res['group_sizes'] = np.array([len(grp) for grp in dk.groupings])
  res['groupings_flat'] = np.array([int(c) for grp in dk.groupings for c in grp])
  res['lml'] = gp.compute_log_marginal_likelihood()
  res['noise'], res['scale'] = gp.noise_var, gp.kernel.hyperparams['scale']
  joint = np.concatenate((ZZ, XX), axis=1)
  res['K'] = gp.kernel(joint, joint)
  res['K_cross'] = gp.kernel(np.concatenate((Zs, Xs), axis=1), joint)
  mu, sd = gp.eval_at_fidel(list(Zs), list(Xs), 'std')
  res['mu'], res['sd'] = mu, sd
  res['rand_after'] = np.random.random()
  np.savez_compressed(os.path.join(OUT, 'mf_fitter_additive_f1_d6_n40.npz'), **res)
  print('wrote mf_fitter_additive_f1_d6_n40: groups', [list(map(int, grp)) for grp in dk.groupings])


def gen_pdoo_cases():
  """ The reference's PDOO (utils/doo.py, oper_utils.py:257-271) on closed-form objectives -- value,
      point and the full query sequence -- and its acquisitions maximised with acq_opt_method
      'pdoo' / 'direct' (the latter falls back to PDOO: no Fortran DIRECT here, oper_utils.py:130)
      on two of the GP cases above. """
  from dragonfly.utils.doo import DOOFunction, pdoo_wrap
  from dragonfly.utils import oper_utils as ref_ou
  from dragonfly.gp.gp_core import GP
  from dragonfly.gp import kernel as K
  from dragonfly.opt import gpb_acquisitions as A
  from dragonfly.exd.domains import EuclideanDomain
  from oracle.test_objectives import PDOO_CASES
  out = {}
  for name, fn, bounds, budget in PDOO_CASES:
    val, pt, _ = ref_ou.pdoo_maximise(fn, bounds, budget)
    _, _, hist = pdoo_wrap(DOOFunction(fn, bounds), budget, 1.0, 0.9, 2, 0.8, 1e-3, 0.5, return_history=True)
    out[name + '_val'], out[name + '_pt'] = val, pt
    out[name + '_queries'] = np.array(hist.query_points)
    print('pdoo %s: %d queries, value %.6f' % (name, len(hist.query_points), val))
  assert ref_ou.direct_ft_wrap is None, 'the fixture records the PDOO fall-back of "direct"'
  for case_name in ('se_d2_n40', 'matern25_d6_n60'):
    g = np.load(os.path.join(OUT, 'gp_' + case_name + '.npz'))
    d = g['X'].shape[1]
    if case_name.startswith('se'):
      kern = K.SEKernel(d, float(g['kern_scale']), g['kern_bw'])
    else:
      kern = K.MaternKernel(d, float(g['kern_nu']), float(g['kern_scale']), g['kern_bw'])
    mean_c = float(g['mean_c'])
    gp = GP(list(g['X']), list(g['Y']), kern, lambda x, _c=mean_c: np.array([_c] * len(x)), float(g['noise']))
    bounds = np.array([[0.0, 1.0]] * d)
    for method, acq, in_progress in (('pdoo', 'ucb', ()), ('pdoo', 'ei', ()), ('direct', 'ucb', ()),
                                     ('pdoo', 'pi', (g['Xh'][0],))):
      anc = Namespace(max_evals=300, t=len(g['Y']), domain=EuclideanDomain(bounds),
                      curr_max_val=float(g['Y'].max()), eval_points_in_progress=list(in_progress),
                      acq_opt_method=method, handle_parallel='halluc', is_mf=False, domain_bounds=bounds)
      with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        out['%s_%s_%s' % (case_name, method, acq)] = np.asarray(getattr(A.asy, acq)(gp, anc))
      print('gp %s %s %s ->' % (case_name, method, acq), out['%s_%s_%s' % (case_name, method, acq)])
  # add-UCB with a tree-search maximiser: one PDOO run per additive group (gpb_acquisitions.py:159-183)
  g = np.load(os.path.join(OUT, 'gp_additive_d10_n80.npz'))
  groups = [[int(c) for c in row if c >= 0] for row in g['kern_groups']]
  subs = []
  for grp, bw, kd in zip(groups, g['kern_sub_bws'], g['kern_sub_kinds']):
    bw = np.array(bw[:len(grp)])
    subs.append(K.SEKernel(len(grp), 1.0, bw) if kd == 0 else K.MaternKernel(len(grp), 2.5, 1.0, bw))
  add_kern = K.AdditiveKernel(float(g['kern_scale']), subs, groups)
  mean_c = float(g['mean_c'])
  add_gp = GP(list(g['X']), list(g['Y']), add_kern, lambda x, _c=mean_c: np.array([_c] * len(x)), float(g['noise']))
  bounds = np.array([[0.0, 1.0]] * g['X'].shape[1])
  anc = Namespace(max_evals=800, t=len(g['Y']), domain=EuclideanDomain(bounds), curr_max_val=float(g['Y'].max()),
                  eval_points_in_progress=[], acq_opt_method='pdoo', handle_parallel='halluc', is_mf=False,
                  domain_bounds=bounds)
  out['additive_d10_n80_pdoo_add_ucb'] = np.asarray(A.asy.add_ucb(add_gp, anc))
  print('gp additive pdoo add_ucb ->', out['additive_d10_n80_pdoo_add_ucb'])
  # the fitter's maximum-likelihood tuning by tree search (gp_core.py:427-434, 463-468)
  from dragonfly.gp.euclidean_gp import EuclideanGPFitter
  f = np.load(os.path.join(OUT, 'fitter_d3_n45.npz'))
  for kt, method in (('se', 'pdoo'), ('matern', 'direct')):
    opts = Namespace(kernel_type=kt, ml_hp_tune_opt=method, hp_tune_max_evals=250, hp_tune_criterion='ml')
    np.random.seed(4343)
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      _, gp, hps = EuclideanGPFitter(list(f['X']), list(f['Y']), options=opts).fit_gp()
    out['fit_%s_%s_cts_hps' % (kt, method)] = np.array(hps[0], dtype=float)
    out['fit_%s_%s_lml' % (kt, method)] = gp.compute_log_marginal_likelihood()
    print('fitter %s %s ->' % (kt, method), hps[0], gp.compute_log_marginal_likelihood())
  np.savez_compressed(os.path.join(OUT, 'pdoo_cases.npz'), **out)
  print('wrote pdoo_cases')


def gen_slice_cases():
  """ The reference's slice sampler (sampling/slice.py through distributions/model.py:51-54) on
      closed-form log densities: the chain, the number of density calls, and the next number of the
      global random stream after it. """
  from dragonfly.distributions.model import Model
  from oracle.test_objectives import SLICE_CASES
  out = {}
  for name, logp, start, num, burn, seed in SLICE_CASES:
    calls = [0]
    def counted(x, _f=logp, _c=calls):
      _c[0] += 1
      return _f(x)
    np.random.seed(seed)
    chain = Model(None, counted, None).draw_samples('slice', num, start, burn)
    out[name + '_chain'] = np.asarray(chain)
    out[name + '_calls'] = calls[0]
    out[name + '_next_random'] = np.random.random()
    print('slice %s: %d samples, %d density calls, mean %.4f' % (name, len(chain), calls[0], np.mean(chain)))
  np.savez_compressed(os.path.join(OUT, 'slice_cases.npz'), **out)
  print('wrote slice_cases')


def gen_trajectory_case():
  """ A multi-step run of the reference's EuclideanGPBandit (opt/gp_bandit.py:551) on Branin in
      ask/tell mode, with the book-keeping of its main loop (exd/exd_core.py:706-725: set the next
      GP before every step, a new model every few steps) driven by hand: every call that crosses a
      seam is recorded IN ORDER with the global RNG state before and after it --
        fit      EuclideanGPFitter.fit_gp_for_gp_bandit      (S3: data, options)
        next_gp  EuclideanGPFitter.get_next_gp               (S3: the GP's hyper-parameters)
        add      GP.add_data_multiple                        (S2)
        acq      gpb_acquisitions.asy.<ucb|ei|ts|ttei>      (S4: anc_data, the chosen point)
      tests/test_gpu_trajectory.py replays the events through the mirrors on the MI355X. """
  import json
  from dragonfly.opt import gp_bandit
  from dragonfly.opt import gpb_acquisitions as A
  from dragonfly.gp import euclidean_gp as EG
  from dragonfly.gp import gp_core as GC
  from dragonfly.exd.domains import EuclideanDomain
  from dragonfly.exd.experiment_caller import EuclideanFunctionCaller
  from dragonfly.utils.option_handler import load_options
  from dragonfly.utils.euclidean_synthetic_functions import get_mf_branin_function
  branin = get_mf_branin_function(1)[1]
  events = []

  def st():
    s = np.random.get_state()
    return [[int(v) for v in s[1]], int(s[2]), int(s[3]), float(s[4])]

  def plain(v):
    if isinstance(v, (bool, int, float, str)) or v is None:
      return v
    if isinstance(v, (np.integer,)):
      return int(v)
    if isinstance(v, (np.floating,)):
      return float(v)
    return None

  def gp_desc(gp):
    k = gp.kernel
    return dict(kernel=type(k).__name__, nu=float(k.hyperparams.get('nu', 0.0) or 0.0),
                scale=float(k.hyperparams['scale']),
                bw=[float(b) for b in np.ravel(k.hyperparams['dim_bandwidths'])],
                noise=float(gp.noise_var), mean=float(gp.mean_func([np.zeros(k.dim)])[0]), n=int(gp.num_tr_data))

  orig_fit, orig_next = EG.EuclideanGPFitter.fit_gp_for_gp_bandit, EG.EuclideanGPFitter.get_next_gp
  orig_add = GC.GP.add_data_multiple
  orig_acq = {name: getattr(A.asy, name) for name in ('ucb', 'ei', 'ts', 'ttei')}

  def fit_spy(self, num_samples=1):
    ev = dict(type='fit', before=st(), X=[[float(v) for v in x] for x in self.X], Y=[float(y) for y in self.Y],
              num_samples=int(num_samples),
              options={k: plain(v) for k, v in vars(self.options).items() if plain(v) is not None or v is None})
    events.append(ev)
    ret = orig_fit(self, num_samples)
    ev['after'] = st()
    return ret

  def next_spy(self):
    ev = dict(type='next_gp', before=st())
    events.append(ev)
    ret = orig_next(self)
    ev.update(after=st(), fit_type=ret[0], method=ret[1], gp=gp_desc(ret[2]))
    return ret

  def add_spy(self, X_new, Y_new):
    events.append(dict(type='add', X=[[float(v) for v in x] for x in X_new], Y=[float(y) for y in Y_new]))
    return orig_add(self, X_new, Y_new)

  def acq_spy(name):
    def spy(gp, anc):
      ev = dict(type='acq', acq=name, before=st(), gp=gp_desc(gp),
                anc=dict(max_evals=int(anc.max_evals), t=int(anc.t), curr_max_val=float(anc.curr_max_val),
                         acq_opt_method=str(anc.acq_opt_method), handle_parallel=str(anc.handle_parallel),
                         bounds=[[float(a), float(b)] for a, b in anc.domain.bounds],
                         in_progress=[[float(v) for v in x] for x in anc.eval_points_in_progress]))
      events.append(ev)
      pt = orig_acq[name](gp, anc)
      ev.update(after=st(), point=[float(v) for v in pt])
      return pt
    return spy

  opts = load_options(gp_bandit.get_all_euc_gp_bandit_args())
  opts.kernel_type = 'matern'
  opts.acq = 'ucb-ei-ts-ttei'
  opts.acq_opt_method = 'rand'
  opts.acq_opt_max_evals = 400
  opts.gpb_hp_tune_criterion = 'ml'
  opts.gpb_ml_hp_tune_opt = 'rand'
  opts.hp_tune_max_evals = 40
  opts.build_new_model_every = 4
  EG.EuclideanGPFitter.fit_gp_for_gp_bandit = fit_spy
  EG.EuclideanGPFitter.get_next_gp = next_spy
  GC.GP.add_data_multiple = add_spy
  for name in orig_acq:
    setattr(A.asy, name, acq_spy(name))
  try:
    np.random.seed(31415)
    caller = EuclideanFunctionCaller(None, EuclideanDomain([[-5, 10], [0, 15]]))
    opt = gp_bandit.EuclideanGPBandit(caller, ask_tell_mode=True, options=opts, reporter='silent')
    opt.initialise()
    Xraw = np.random.RandomState(7).random_sample((12, 2)) * np.array([15.0, 15.0]) + np.array([-5.0, 0.0])
    opt.tell([(x, branin(x)) for x in Xraw])
    opt.first_qinfos = []
    asked = []
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      for it in range(14):
        opt.step_idx = 12 + it
        if it % opts.build_new_model_every == 0:
          opt._build_new_model()          # pylint: disable=protected-access
        opt._set_next_gp()                # pylint: disable=protected-access  (exd_core.py: _main_loop_pre)
        x = np.array(opt.ask())
        asked.append(x)
        opt.tell([(x, branin(x))])
  finally:
    EG.EuclideanGPFitter.fit_gp_for_gp_bandit, EG.EuclideanGPFitter.get_next_gp = orig_fit, orig_next
    GC.GP.add_data_multiple = orig_add
    for name, fn in orig_acq.items():
      setattr(A.asy, name, fn)
  blob = np.frombuffer(json.dumps(events).encode('utf-8'), dtype=np.uint8)
  np.savez_compressed(os.path.join(OUT, 'trajectory_branin.npz'), events_json=blob, asked=np.array(asked))
  kinds = [e['type'] for e in events]
  print('wrote trajectory_branin: %d events (%s), acquisitions used: %s'
        % (len(events), ', '.join('%s x%d' % (k, kinds.count(k)) for k in sorted(set(kinds))),
           [e['acq'] for e in events if e['type'] == 'acq']))


def gen_nonpsd_cases():
  """ Kernels that are NOT guaranteed PSD through the real reference:
      (1) gp_core.GP with a sigmoid kernel tanh(a x.y + b) and handle_non_psd_kernels =
          'project_first' / 'try_before_project' (gp_core.py:827-857: eigen-projection of the Gram
          matrix and of every posterior covariance);
      (2) cartesian_product_gp.CPGP (cartesian_product_gp.py:208-248) with the reference's
          CartesianProductKernel over [SE on R^2, sigmoid on R^1], one part also from a distance
          list... (parts evaluated directly here).
      Stored: inputs, the reference's K, L, alpha, lml, eval mean / std / covar, hallucinated std. """
  from dragonfly.gp import gp_core as GC
  from dragonfly.gp.kernel import Kernel, SEKernel, CartesianProductKernel
  from dragonfly.gp.cartesian_product_gp import CPGP

  class SigmoidKernel(Kernel):
    """ k(x, y) = tanh(a x.y + b): symmetric, indefinite. """
    def __init__(self, a, b):
      super(SigmoidKernel, self).__init__()
      self.add_hyperparams(a=a, b=b)
    def is_guaranteed_psd(self):
      return False
    def _child_evaluate(self, X1, X2):
      X1, X2 = np.array(X1, dtype=float), np.array(X2, dtype=float)
      return np.tanh(self.hyperparams['a'] * X1.dot(X2.T) + self.hyperparams['b'])

  rs = np.random.RandomState(909)
  res = {}
  # (1) plain GP, sigmoid kernel on R^3
  n, d, m, q = 60, 3, 15, 2
  X = rs.randn(n, d)
  Y = np.sin(X.sum(axis=1)) + 0.05 * rs.randn(n)
  Xs, Xh = rs.randn(m, d), rs.randn(q, d)
  a, b, noise, mean_c = 0.7, -0.4, 0.05, float(np.median(Y))
  mean_func = lambda x: np.array([mean_c] * len(x))
  res.update(X=X, Y=Y, Xs=Xs, Xh=Xh, a=a, b=b, noise=noise, mean_c=mean_c)
  for mode in ('project_first', 'try_before_project'):
    gp = GC.GP(list(X), list(Y), SigmoidKernel(a, b), mean_func, noise, handle_non_psd_kernels=mode)
    mu, sd = gp.eval(list(Xs), 'std')
    _, cov = gp.eval(list(Xs), 'covar')
    _, sdh = gp.eval_with_hallucinated_observations(list(Xs), list(Xh), 'std')
    res.update({mode + '_K': gp.K_trtr_wo_noise, mode + '_L': gp.L, mode + '_alpha': gp.alpha,
                mode + '_lml': gp.compute_log_marginal_likelihood(), mode + '_mu': mu, mode + '_sd': sd,
                mode + '_cov': cov, mode + '_sdh': sdh})
  res['min_eig_K'] = float(np.linalg.eigvalsh(res['project_first_K']).min())
  # (2) CPGP: domain = R^2 x R^1, kernel = scale * SE(part 0) * sigmoid(part 1)
  n2, m2 = 50, 12
  P0, P1 = rs.rand(n2, 2), rs.randn(n2, 1)
  Yc = np.cos(3 * P0.sum(axis=1)) + 0.3 * P1[:, 0] + 0.05 * rs.randn(n2)
  T0, T1 = rs.rand(m2, 2), rs.randn(m2, 1)
  H0, H1 = rs.rand(q, 2), rs.randn(q, 1)
  cp_scale, bw0, a1, b1, noise2, mean2 = 1.3, np.array([0.4, 0.6]), 0.9, 0.2, 0.03, float(np.median(Yc))
  kern = CartesianProductKernel(cp_scale, [SEKernel(2, 1.0, bw0), SigmoidKernel(a1, b1)])
  to_lists = lambda A, B: [[A[i], B[i]] for i in range(len(A))]
  gp = CPGP(to_lists(P0, P1), list(Yc), kern, lambda x: np.array([mean2] * len(x)), noise2)
  mu, sd = gp.eval(to_lists(T0, T1), 'std')
  _, cov = gp.eval(to_lists(T0, T1), 'covar')
  _, sdh = gp.eval_with_hallucinated_observations(to_lists(T0, T1), to_lists(H0, H1), 'std')
  res.update(cp_P0=P0, cp_P1=P1, cp_Y=Yc, cp_T0=T0, cp_T1=T1, cp_H0=H0, cp_H1=H1, cp_scale=cp_scale, cp_bw0=bw0,
             cp_a=a1, cp_b=b1, cp_noise=noise2, cp_mean=mean2, cp_K=gp.K_trtr_wo_noise, cp_L=gp.L,
             cp_alpha=gp.alpha, cp_lml=gp.compute_log_marginal_likelihood(), cp_mu=mu, cp_sd=sd, cp_cov=cov,
             cp_sdh=sdh, cp_min_eig_K=float(np.linalg.eigvalsh(gp.K_trtr_wo_noise).min()))
  np.savez_compressed(os.path.join(OUT, 'nonpsd_gp.npz'), **res)
  print('wrote nonpsd_gp (min eig K: plain %.3f, cp %.3f)' % (res['min_eig_K'], res['cp_min_eig_K']))


def engine_trace_scenarios():
  """ (name, run, install keyword arguments, meta) for each of the 25 configurations of tests/test_gpu_install_end_to_end.py """
  tests_dir = os.path.join(os.path.dirname(HERE), 'tests')
  if tests_dir not in sys.path:
    sys.path.insert(0, tests_dir)
  import_reference()
  import test_install_end_to_end as E

  def maximise_function_run():
    from dragonfly import maximise_function
    f = lambda x: -float((x[0] - 0.3) ** 2 + (x[1] + 0.2) ** 2) + 0.05 * float(np.cos(7 * x[0]))
    np.random.seed(77)
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      val, pt, history = maximise_function(f, [[-1, 1], [-1, 1]], 7)
    return [np.array([val]), np.array(pt), np.array(history.query_points)]

  scenarios = []
  for i, cfg in enumerate(E.CONFIGS):
    scenarios.append(('ask_%02d_%s_%s_%s' % (i, cfg['kernel_type'], cfg['acq'], cfg['acq_opt_method']),
                      (lambda c=cfg: [np.array(p) for p in E._ask(c)[0]]), {}, dict(kind='ask', options=cfg)))       # pylint: disable=protected-access
  for i, (mode, workers, extra) in enumerate(E.FULL_RUNS):
    scenarios.append(('full_%02d_%s%d_%s' % (i, mode, workers, extra['acq'].replace('-', '_')),
                      (lambda m=mode, w=workers, e=extra: [E._full_run(m, w, e)]), {},                              # pylint: disable=protected-access
                      dict(kind='full run', mode=mode, workers=workers, options=extra)))
  for acq, method in (('ts', 'rand'), ('ucb', 'rand'), ('ucb', 'pdoo')):
    scenarios.append(('moo_%s_%s' % (acq, method), (lambda a=acq, m=method: [E._moo_run(a, m)]), {},              # pylint: disable=protected-access
                      dict(kind='multi-objective', acq=acq, method=method)))
  for workers, acq in ((1, None), (3, 'ucb-ts')):
    scenarios.append(('mf_%d_%s' % (workers, (acq or 'default').replace('-', '_')),
                      (lambda w=workers, a=acq: [np.array(v) for v in E._mf_run(w, a)[:2]]), dict(multi_fidelity=True),   # pylint: disable=protected-access
                      dict(kind='multi-fidelity', workers=workers, acq=acq)))
  scenarios.append(('maximise_function_defaults', maximise_function_run, {}, dict(kind='dragonfly.maximise_function, default options')))
  return scenarios


def record_engine_trace(run, install_kwargs):
  """ One scenario: the reference as it is, then the same run with install() on the recording stand-in engine.
      Returns (the reference's results, the install() run's results, the log of engine calls). """
  tests_dir = os.path.join(os.path.dirname(HERE), 'tests')
  if tests_dir not in sys.path:
    sys.path.insert(0, tests_dir)
  import engine_trace as ET
  from dragonfly_amd import install
  from dragonfly_amd import euclidean_gp, general_utils, gp_core, gpb_acquisitions, kernel
  from dragonfly_amd import engine as engine_mod
  mods = (engine_mod, euclidean_gp, general_utils, gp_core, kernel)
  want = run()                                   # the reference as it is
  eng, log = ET.recording_engine()
  saved = [(m, m.get_engine) for m in mods]
  saved_dc = gpb_acquisitions.DEVICE_CANDIDATES
  for m in mods:
    m.get_engine = (lambda _e=eng: _e)
  gpb_acquisitions.DEVICE_CANDIDATES = False
  install.install(**install_kwargs)
  try:
    got = run()
  finally:
    install.uninstall()
    for m, fn in saved:
      m.get_engine = fn
    gpb_acquisitions.DEVICE_CANDIDATES = saved_dc
  return want, got, log


def gen_engine_traces():
  """ For each of the 25 configurations of tests/test_gpu_install_end_to_end.py: the UNMODIFIED reference optimiser with
      dragonfly_amd.install(), on the NumPy stand-in engine behind a recorder (tests/engine_trace.py) -- every call that
      reaches the engine object, in order, with arguments and results -> tests/golden/engine_trace_<name>.npz.  Before
      anything is written the run is checked against the same run WITHOUT install(): the same points, bit for bit
      (dragonfly/opt/gp_bandit.py:405-421, 490, 647-673; apis/opt.py:138).
      The traces record the call stream of the install() of THIS commit: any change to what dragonfly_amd sends through
      the engine object (batching in slice_sampler.py, install.py, the mirrors) needs them re-recorded --
      tests/test_engine_traces_cpu.py::test_committed_trace_is_the_call_stream_of_this_tree re-records one and fails
      when the committed trace is stale. """
  scenarios = engine_trace_scenarios()
  import engine_trace as ET
  total = 0
  for name, run, install_kwargs, meta in scenarios:
    want, got, log = record_engine_trace(run, install_kwargs)
    assert len(got) == len(want) and all(np.array_equal(g, w) for g, w in zip(got, want)), name
    meta = dict(meta, name=name, events=len(log.events), reference_points_equal=True,
                result_shapes=[list(np.shape(w)) for w in want])
    ET.save(os.path.join(OUT, 'engine_trace_%s.npz' % name), log, meta)
    kinds = {}
    for ev in log.events:
      kinds[ev['m']] = kinds.get(ev['m'], 0) + 1
    total += len(log.events)
    print('wrote engine_trace_%s: %d calls %s' % (name, len(log.events), kinds), flush=True)
  print('engine traces: %d scenarios, %d calls' % (len(scenarios), total))


if __name__ == '__main__':
  os.makedirs(OUT, exist_ok=True)
  import_reference()
  if len(sys.argv) > 1 and sys.argv[1] == 'mfgp':
    gen_mfgp_case()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'polyexp':
    gen_poly_expdecay_cases()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'mffitter':
    gen_mf_fitter_case()
    gen_mf_fitter_additive_case()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'mffitter_additive':
    gen_mf_fitter_additive_case()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'post_sampling':
    gen_post_sampling_cases()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'slice':
    gen_slice_cases()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'pdoo':
    gen_pdoo_cases()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'nonpsd':
    gen_nonpsd_cases()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'trajectory':
    gen_trajectory_case()
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == 'engine_traces':
    gen_engine_traces()
    sys.exit(0)
  gen_gp_cases()
  gen_fitter_case()
  gen_c1_case()
  gen_mfgp_case()
  gen_poly_expdecay_cases()
  gen_mf_fitter_case()
  gen_mf_fitter_additive_case()
  gen_pdoo_cases()
  gen_slice_cases()
  gen_post_sampling_cases()
  gen_trajectory_case()
  gen_nonpsd_cases()
  gen_engine_traces()
