"""CPU plumbing test of the drop-in seams: the REAL
reference optimiser (dragonfly.opt.gp_bandit.EuclideanGPBandit, ask/tell) runs once as it is and
once with dragonfly_amd.install(); with the same seed it must recommend the same point, through
hyper-parameter tuning by the reference's own fitter, the GP, and the acquisition.

There is no GPU in the build container and no reference on the GPU box, so here the mirrors talk
to tests/oracle_engine.py (NumPy arithmetic behind the Engine interface -- test infrastructure).
What this pins is everything ABOVE the C-ABI: constructor signatures, attributes the reference
reads (gp.X, gp.kernel.hyperparams, ...), anc_data fields, random-number call order, return types.
The C-ABI side of the same acquisitions is pinned on the MI355X by tests/test_gpu_golden.py."""
import os
import warnings

from argparse import Namespace

import numpy as np
import pytest

REF = os.environ.get('DRAGONFLY_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'dragonfly')),
                                reason='needs the reference tree (build container only)')


def _branin_data():
  from dragonfly.utils.euclidean_synthetic_functions import get_mf_branin_function
  branin = get_mf_branin_function(1)[1]
  Xraw = np.random.RandomState(101).random_sample((60, 2)) * np.array([15.0, 15.0]) + np.array([-5.0, 0.0])
  return Xraw, np.array([branin(x) for x in Xraw])


def _ask(options_update, num_asks=2):
  """ tell 60 Branin evaluations, then ask; returns the recommended points and the fitted GP's
      hyper-parameters. """
  from dragonfly.opt import gp_bandit
  from dragonfly.exd.domains import EuclideanDomain
  from dragonfly.exd.experiment_caller import EuclideanFunctionCaller
  from dragonfly.utils.option_handler import load_options
  opts = load_options(gp_bandit.get_all_euc_gp_bandit_args())
  opts.gpb_hp_tune_criterion = 'ml'
  opts.hp_tune_max_evals = 40
  for k, v in options_update.items():
    setattr(opts, k, v)
  np.random.seed(2024)
  caller = EuclideanFunctionCaller(None, EuclideanDomain([[-5, 10], [0, 15]]))
  opt = gp_bandit.EuclideanGPBandit(caller, ask_tell_mode=True, options=opts, reporter='silent')
  opt.initialise()
  Xraw, Y = _branin_data()
  opt.tell([(x, y) for x, y in zip(Xraw, Y)])
  opt.first_qinfos = []
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    points = [np.array(opt.ask()) for _ in range(num_asks)]
  gp = opt.gp
  hps = (gp.kernel.hyperparams['scale'], np.asarray(gp.kernel.hyperparams['dim_bandwidths'], dtype=float),
         gp.noise_var, type(gp).__module__)
  return points, hps


CONFIGS = [
  dict(kernel_type='se', acq='ucb', acq_opt_method='rand', acq_opt_max_evals=300, gpb_ml_hp_tune_opt='rand'),
  dict(kernel_type='matern', acq='ei', acq_opt_method='rand', acq_opt_max_evals=300, gpb_ml_hp_tune_opt='rand'),
  dict(kernel_type='se', acq='ts', acq_opt_method='rand', acq_opt_max_evals=200, gpb_ml_hp_tune_opt='rand'),
  dict(kernel_type='se', acq='ttei', acq_opt_method='rand', acq_opt_max_evals=200, gpb_ml_hp_tune_opt='rand'),
  dict(kernel_type='se', acq='ucb', acq_opt_method='pdoo', acq_opt_max_evals=150, gpb_ml_hp_tune_opt='rand'),
  dict(kernel_type='se', acq='pi', acq_opt_method='direct', acq_opt_max_evals=150, gpb_ml_hp_tune_opt='pdoo'),
  dict(kernel_type='se', acq='add_ucb', acq_opt_method='rand', acq_opt_max_evals=300,
       gpb_ml_hp_tune_opt='rand'),
  dict(kernel_type='matern', acq='add_ucb', acq_opt_method='pdoo', acq_opt_max_evals=150,
       gpb_ml_hp_tune_opt='rand'),
  dict(kernel_type='default', acq='default', acq_opt_method='default', gpb_ml_hp_tune_opt='rand'),
  dict(kernel_type='se', acq='ucb', acq_opt_method='rand', acq_opt_max_evals=200, gpb_hp_tune_criterion='post_sampling',
       gpb_post_hp_tune_burn=15),
  dict(kernel_type='matern', acq='ei', acq_opt_method='rand', acq_opt_max_evals=200,
       gpb_hp_tune_criterion='ml-post_sampling', gpb_ml_hp_tune_opt='rand', gpb_post_hp_tune_burn=10),
]


@pytest.mark.parametrize('cfg', CONFIGS, ids=['%s-%s-%s-%d' % (c['kernel_type'], c['acq'], c['acq_opt_method'], i) for i, c in enumerate(CONFIGS)])
def test_reference_bandit_recommends_the_same_points_with_the_engine_installed(cfg, monkeypatch):
  from oracle.make_golden import import_reference
  import_reference()
  from oracle_engine import patch_engine
  from dragonfly_amd import install
  want_points, want_hps = _ask(cfg)
  assert want_hps[3].startswith('dragonfly.')
  eng = patch_engine(monkeypatch)
  eng.lml_batch_sizes = []
  install.install()
  try:
    got_points, got_hps = _ask(cfg)
  finally:
    install.uninstall()
  assert got_hps[3].startswith('dragonfly_amd.')               # the bandit's GP is the mirror
  # ... and its fitter tuned in batches: the whole random sample in one call, the tree search a
  # frontier per call (the reference makes one fit per candidate: hp_tune_max_evals = 40 of them)
  slice_only = cfg.get('gpb_hp_tune_criterion') == 'post_sampling'    # batches of a few candidates per loop
  assert len(eng.lml_batch_sizes) > 0 and max(eng.lml_batch_sizes) >= (3 if slice_only else 10)
  assert got_hps[0] == want_hps[0] and np.array_equal(got_hps[1], want_hps[1]) and got_hps[2] == want_hps[2]
  for got, want in zip(got_points, want_points):
    assert np.array_equal(got, want)


def _moo_run(acq, method='rand'):
  """ A short multi-objective run of the reference (opt/multiobjective_gp_bandit.py): two
      objectives on [0,1]^3, synthetic worker, 14 evaluations. """
  from dragonfly.opt.multiobjective_gp_bandit import multiobjective_gpb_from_multi_func_caller, \
      get_all_euc_moo_gp_bandit_args
  from dragonfly.exd.experiment_caller import EuclideanMultiFunctionCaller
  from dragonfly.exd.domains import EuclideanDomain
  from dragonfly.exd.worker_manager import SyntheticWorkerManager
  from dragonfly.utils.option_handler import load_options
  f1 = lambda x: -float(np.sum((np.asarray(x) - 0.2) ** 2))
  f2 = lambda x: -float(np.sum((np.asarray(x) - 0.8) ** 2))
  caller = EuclideanMultiFunctionCaller([f1, f2], EuclideanDomain([[0, 1]] * 3), vectorised=False)
  opts = load_options(get_all_euc_moo_gp_bandit_args())
  opts.gpb_hp_tune_criterion = 'ml'
  opts.gpb_ml_hp_tune_opt = 'rand'
  opts.hp_tune_max_evals = 30
  opts.acq_opt_max_evals = 100
  opts.acq_opt_method = method
  opts.acq = acq
  np.random.seed(7)
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    _, _, history = multiobjective_gpb_from_multi_func_caller(
        caller, SyntheticWorkerManager(1, time_distro='const'), 14, is_mf=False, options=opts,
        reporter='silent')
  return np.array(history.query_points)


@pytest.mark.parametrize('acq,method', [('ts', 'rand'), ('ucb', 'rand'), ('ucb', 'pdoo'), ('ucb', 'direct')])
def test_reference_multiobjective_bandit_inherits_the_engine(acq, method, monkeypatch):
  """ the multi-objective acquisitions (opt/multiobjective_gpb_acquisitions.py:
      19-107) only use gp.eval / gp.draw_samples of the GPs their Euclidean fitters build through the
      rebound module global, so they run on the engine without a line of their own. """
  from oracle.make_golden import import_reference
  import_reference()
  from oracle_engine import patch_engine
  from dragonfly_amd import install
  import dragonfly_amd.gp_core as mirror_gp
  want = _moo_run(acq, method)
  patch_engine(monkeypatch)
  built = []
  orig = mirror_gp.GP.build_posterior
  monkeypatch.setattr(mirror_gp.GP, 'build_posterior', lambda self: (built.append(1), orig(self))[1])
  install.install()
  try:
    got = _moo_run(acq, method)
  finally:
    install.uninstall()
  assert len(built) > 0                       # the mirror GP did the fitting
  assert got.shape == want.shape and np.array_equal(got, want)


def test_install_without_batched_tuning_leaves_the_fitter_alone(monkeypatch):
  from oracle.make_golden import import_reference
  import_reference()
  from oracle_engine import patch_engine
  from dragonfly_amd import install
  import dragonfly.opt.gp_bandit as ref_gp_bandit
  import dragonfly.gp.euclidean_gp as ref_egp
  cfg = CONFIGS[0]
  want_points, _ = _ask(cfg)
  eng = patch_engine(monkeypatch)
  eng.lml_batch_sizes = []
  install.install(batched_tuning=False)
  try:
    assert ref_gp_bandit.EuclideanGPFitter is ref_egp.EuclideanGPFitter
    got_points, _ = _ask(cfg)
  finally:
    install.uninstall()
  assert eng.lml_batch_sizes == [] and all(np.array_equal(a, b) for a, b in zip(got_points, want_points))
  install.install()
  try:
    assert ref_gp_bandit.EuclideanGPFitter is not ref_egp.EuclideanGPFitter
    assert issubclass(ref_gp_bandit.EuclideanGPFitter, ref_egp.EuclideanGPFitter)
  finally:
    install.uninstall()
  assert ref_gp_bandit.EuclideanGPFitter is ref_egp.EuclideanGPFitter


def _mf_run(num_workers=1, acq=None):
  """ A short multi-fidelity (BOCA) run of the reference: 1-D fidelity space, 2-D domain. """
  from dragonfly.opt import gp_bandit
  from dragonfly.exd.domains import EuclideanDomain
  from dragonfly.exd.experiment_caller import EuclideanFunctionCaller
  from dragonfly.exd.worker_manager import SyntheticWorkerManager
  from dragonfly.utils.option_handler import load_options
  f = lambda z, x: -float(np.sum((np.asarray(x) - 0.3) ** 2)) - \
                   0.3 * (1 - float(np.ravel(z)[0])) * float(np.sin(5 * np.sum(x)))
  cost = lambda z: 0.2 + 0.8 * float(np.ravel(z)[0])
  caller = EuclideanFunctionCaller(f, EuclideanDomain([[0, 1]] * 2), vectorised=False,
                                   raw_fidel_space=EuclideanDomain([[0, 1]]), fidel_cost_func=cost,
                                   raw_fidel_to_opt=np.array([1.0]))
  opts = load_options(gp_bandit.get_all_mf_euc_gp_bandit_args())
  opts.gpb_hp_tune_criterion = 'ml'
  opts.gpb_ml_hp_tune_opt = 'rand'
  opts.hp_tune_max_evals = 30
  opts.acq_opt_max_evals = 100
  opts.acq_opt_method = 'rand'
  if acq is not None:
    opts.acq = acq
  np.random.seed(9)
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    opt = gp_bandit.EuclideanGPBandit(caller, SyntheticWorkerManager(num_workers, time_distro='const'), is_mf=True,
                                      options=opts, reporter='silent')
    _, _, history = opt.optimise(8)
  return np.array(history.query_points), np.array(history.query_fidels), type(opt.gp).__module__


@pytest.mark.parametrize('num_workers,acq,mf_gp', [(1, None, True), (3, 'ucb-ts', True), (2, 'ei', False)])
def test_reference_multifidelity_bandit_with_the_mf_gp_installed(num_workers, acq, mf_gp, monkeypatch):
  """ install(multi_fidelity=True): BOCA's GP (gp/euclidean_gp.py:347-412, built at :707) is the
      mirror EuclideanMFGP -- coordinate-product kernel on the engine -- and the run is unchanged. """
  from oracle.make_golden import import_reference
  import_reference()
  from oracle_engine import patch_engine
  from dragonfly_amd import install
  want_pts, want_fidels, want_mod = _mf_run(num_workers, acq)
  assert want_mod.startswith('dragonfly.')
  patch_engine(monkeypatch)
  install.install(multi_fidelity=mf_gp)
  try:
    got_pts, got_fidels, got_mod = _mf_run(num_workers, acq)
  finally:
    install.uninstall()
  assert got_mod.startswith('dragonfly_amd.' if mf_gp else 'dragonfly.')
  assert np.array_equal(got_pts, want_pts) and np.array_equal(got_fidels, want_fidels)


def _full_run(mode, num_workers, extra):
  """ A complete (short) optimisation of the reference with several synthetic workers: in-progress
      evaluations are hallucinated (asy) or the batch is built sequentially (syn). """
  from dragonfly.opt import gp_bandit
  from dragonfly.exd.domains import EuclideanDomain
  from dragonfly.exd.experiment_caller import EuclideanFunctionCaller
  from dragonfly.exd.worker_manager import SyntheticWorkerManager
  from dragonfly.utils.option_handler import load_options
  f = lambda x: -float(np.sum((np.asarray(x) - np.array([0.2, 0.7, 0.5])) ** 2)) + 0.1 * float(np.sin(9 * x[0]))
  caller = EuclideanFunctionCaller(f, EuclideanDomain([[0, 1]] * 3), vectorised=False)
  opts = load_options(gp_bandit.get_all_euc_gp_bandit_args())
  opts.gpb_hp_tune_criterion = 'ml'
  opts.gpb_ml_hp_tune_opt = 'rand'
  opts.hp_tune_max_evals = 30
  opts.acq_opt_max_evals = 120
  opts.acq_opt_method = 'rand'
  opts.mode = mode
  for k, v in extra.items():
    if k != 'capital':
      setattr(opts, k, v)
  np.random.seed(31)
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    opt = gp_bandit.EuclideanGPBandit(caller, SyntheticWorkerManager(num_workers, time_distro='const'),
                                      options=opts, reporter='silent')
    _, _, history = opt.optimise(extra.get('capital', 18))
  return np.array(history.query_points)


FULL_RUNS = [
  ('asy', 3, dict(acq='ucb')),
  ('asy', 3, dict(acq='ei-ttei-ts')),
  ('syn', 3, dict(acq='ei')),
  ('syn', 2, dict(acq='ucb-add_ucb')),
  ('asy', 2, dict(acq='ucb', use_additive_gp=True, kernel_type='se')),
  ('asy', 1, dict(acq='pi', acq_opt_method='pdoo', acq_opt_max_evals=60)),
  ('asy', 1, dict(acq='ei', use_additive_gp=True, kernel_type='matern', gpb_ml_hp_tune_opt='pdoo')),
  ('asy', 2, dict(acq='add_ucb-ucb', gpb_ml_hp_tune_opt='direct', gpb_hp_tune_criterion='ml-post_sampling',
                  gpb_post_hp_tune_burn=6, capital=8)),
]


@pytest.mark.parametrize('mode,workers,extra', FULL_RUNS, ids=['%s%d-%s-%d' % (m, w, e['acq'], i) for i, (m, w, e) in enumerate(FULL_RUNS)])
def test_reference_full_runs_with_parallel_workers(mode, workers, extra, monkeypatch):
  from oracle.make_golden import import_reference
  import_reference()
  from oracle_engine import patch_engine
  from dragonfly_amd import install
  want = _full_run(mode, workers, extra)
  patch_engine(monkeypatch)
  install.install()
  try:
    got = _full_run(mode, workers, extra)
  finally:
    install.uninstall()
  assert got.shape == want.shape and np.array_equal(got, want)


def test_top_level_maximise_function_with_default_options(monkeypatch):
  """ dragonfly.maximise_function with nothing but defaults: default kernel, acquisition mix,
      'direct' maximisers and the ml / post_sampling tuning mix. """
  from oracle.make_golden import import_reference
  import_reference()
  from oracle_engine import patch_engine
  from dragonfly_amd import install
  from dragonfly import maximise_function
  f = lambda x: -float((x[0] - 0.3) ** 2 + (x[1] + 0.2) ** 2) + 0.05 * float(np.cos(7 * x[0]))
  def run():
    np.random.seed(77)
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      val, pt, history = maximise_function(f, [[-1, 1], [-1, 1]], 7)
    return val, np.array(pt), np.array(history.query_points)
  want = run()
  patch_engine(monkeypatch)
  install.install()
  try:
    got = run()
  finally:
    install.uninstall()
  assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])


@pytest.mark.parametrize('install_kwargs', [{}, dict(cartesian_product=True)], ids=['reference-cpgp', 'mirror-cpgp'])
def test_cartesian_product_domain_runs_keep_working(install_kwargs, monkeypatch):
  """ install() must not break what it does not accelerate: on a Cartesian-product domain (float +
      int + discrete variables) the acquisition entries dispatch to the reference's own callables
      and the CP GP evaluates its Euclidean factor kernels through the mirror kernel classes.
      With install(cartesian_product=True) the CP GP itself is the mirror class (project_first
      posterior on the engine, kernel parts on the host): same run, same points. """
  from oracle.make_golden import import_reference
  import_reference()
  from oracle_engine import patch_engine
  from dragonfly_amd import install
  from dragonfly import maximise_function, load_config
  config = load_config({'domain': [{'name': 'x', 'type': 'float', 'min': 0, 'max': 1, 'dim': 2},
                                   {'name': 'k', 'type': 'int', 'min': 1, 'max': 5},
                                   {'name': 'c', 'type': 'discrete', 'items': ['a', 'b', 'c']}]})
  def f(p):
    x, k, c = p
    return -float(np.sum((np.asarray(x) - 0.4) ** 2)) - 0.1 * (k - 3) ** 2 + {'a': 0.0, 'b': 0.3, 'c': -0.2}[c]
  def run():
    np.random.seed(5)
    from dragonfly.utils.reporters import get_reporter
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      val, pt, hist = maximise_function(f, config.domain, 9, config=config, reporter=get_reporter('silent'))
    return val, str(pt), str(hist.query_points)
  want = run()
  patch_engine(monkeypatch)
  patched = install.install(**install_kwargs)
  assert ('dragonfly.gp.cartesian_product_gp.CPGP' in patched) == bool(install_kwargs)
  try:
    got = run()
  finally:
    install.uninstall()
  assert got == want


def _api_mf():
  from dragonfly import maximise_multifidelity_function
  from dragonfly.utils.reporters import get_reporter
  f = lambda z, x: -float(np.sum((np.asarray(x) - 0.3) ** 2)) - 0.3 * (1 - float(z[0])) * float(np.sin(5 * np.sum(x)))
  np.random.seed(3)
  val, pt, hist = maximise_multifidelity_function(f, [[0, 1]], [[0, 1]] * 2, [1.0], lambda z: 0.2 + 0.8 * float(z[0]),
                                                  6, reporter=get_reporter('silent'))
  return val, str(pt), str(hist.query_points), str(hist.query_fidels)


def _api_mf_expdecay():
  """ BOCA with the exponential-decay fidelity kernel (the reference's ExpDecayKernel objects, left in
      place by install(), reach the device as product factors), ML tuning: the reference's posterior
      sampling has no parameter order for this kernel (gp_core.py:690 fails on it). """
  from dragonfly import maximise_multifidelity_function
  from dragonfly.utils.reporters import get_reporter
  f = lambda z, x: -float(np.sum((np.asarray(x) - 0.3) ** 2)) - 0.3 * (1 - float(z[0])) * float(np.sin(5 * np.sum(x)))
  np.random.seed(3)
  opts = Namespace(fidel_kernel_type='expdecay', mf_gp_fidel_kernel_type='expdecay', gpb_hp_tune_criterion='ml')
  val, pt, hist = maximise_multifidelity_function(f, [[0, 1]], [[0, 1]] * 2, [1.0], lambda z: 0.2 + 0.8 * float(z[0]),
                                                  7, options=opts, reporter=get_reporter('silent'))
  return val, str(pt), str(hist.query_points), str(hist.query_fidels)


def _api_moo():
  from dragonfly import multiobjective_maximise_functions
  from dragonfly.utils.reporters import get_reporter
  f1 = lambda x: -float(np.sum((np.asarray(x) - 0.2) ** 2))
  f2 = lambda x: -float(np.sum((np.asarray(x) - 0.8) ** 2))
  np.random.seed(4)
  vals, pts, hist = multiobjective_maximise_functions([f1, f2], [[0, 1]] * 2, 6, reporter=get_reporter('silent'))
  return str(vals), str(pts), str(hist.query_points)


def _api_min():
  from dragonfly import minimise_function
  from dragonfly.utils.reporters import get_reporter
  np.random.seed(6)
  val, pt, hist = minimise_function(lambda x: float((x[0] - 0.3) ** 2 + np.abs(x[1])), [[-1, 1], [-1, 1]], 7,
                                    reporter=get_reporter('silent'))
  return val, str(pt), str(hist.query_points)


@pytest.mark.parametrize('name,install_kwargs', [('mf', {}), ('mf', dict(multi_fidelity=True)),
                                                 ('mf_expdecay', dict(multi_fidelity=True)), ('moo', {}), ('min', {})],
                         ids=['mf', 'mf-mfgp', 'mf-expdecay', 'moo', 'min'])
def test_top_level_apis_with_default_options(name, install_kwargs, monkeypatch):
  """ dragonfly.maximise_multifidelity_function (BOCA; with and without the mirror MF GP),
      multiobjective_maximise_functions and minimise_function, all options at their defaults. """
  from oracle.make_golden import import_reference
  import_reference()
  from oracle_engine import patch_engine
  from dragonfly_amd import install
  run = {'mf': _api_mf, 'mf_expdecay': _api_mf_expdecay, 'moo': _api_moo, 'min': _api_min}[name]
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    want = run()
    patch_engine(monkeypatch)
    install.install(**install_kwargs)
    built = []
    if name == 'mf_expdecay':
      from dragonfly_amd import mf_gp
      orig_init = mf_gp.EuclideanMFGP.__init__
      def spy(self, *a, **k):
        orig_init(self, *a, **k)
        built.append((type(self.fidel_kernel).__name__, type(self.domain_kernel).__name__, self._generic))
      monkeypatch.setattr(mf_gp.EuclideanMFGP, '__init__', spy)
    try:
      got = run()
    finally:
      install.uninstall()
  assert got == want
  if name == 'mf_expdecay':
    # every GP with a plain domain kernel had its exponential-decay factor evaluated by the engine; only
    # products with an ADDITIVE domain factor (BOCA tries those too) are composed on the host
    assert ('ExpDecayKernel', 'SEKernel', False) in built or ('ExpDecayKernel', 'MaternKernel', False) in built
    assert all(dom == 'AdditiveKernel' for _, dom, generic in built if generic), set(built)
