"""GPU + reference present: the REAL Dragonfly optimiser on the REAL engine.

tests/test_install_end_to_end.py pins the seams above the C-ABI against a NumPy stand-in engine (no
GPU in the build container); the GPU suite pins the C-ABI against reference fixtures.  This file
joins the two halves wherever both a gfx950 device and a Dragonfly checkout exist
(DRAGONFLY_REFERENCE=/path/to/dragonfly-checkout): the unmodified reference optimiser
(dragonfly/opt/gp_bandit.py:490,651,670-673 acquisition lookup; gp/euclidean_gp.py:325-339 fitter ->
GP) runs once as it is -- NumPy on the host -- and once with dragonfly_amd.install(), every GP fit,
tuning batch and acquisition going through ctypes into libdfhip.so.  Same seed, same recommended
points.

The driver's GPU box has no reference checkout, so the file skips there; the builder's run (reference
shipped as untracked scratch for one call) is kept in profiles/r04_install_on_gpu.log."""
import os
import warnings

import numpy as np
import pytest

import test_install_end_to_end as E

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.path.isdir(os.path.join(E.REF, 'dragonfly')),
                                 reason='needs a Dragonfly checkout next to the GPU (DRAGONFLY_REFERENCE)')]


@pytest.fixture
def real_engine(engine, monkeypatch):
  """ The process-wide Engine with spies on the calls that reach libdfhip.so (the counts prove that the
      run went through the library and not through the reference's NumPy). """
  from oracle.make_golden import import_reference
  from dragonfly_amd import engine as engine_mod
  import_reference()
  calls = {'gp_fit': 0, 'gp_fit_gram': 0, 'lml_batch_sizes': []}
  cls = type(engine)
  orig_fit, orig_gram, orig_batch = cls.gp_fit, cls.gp_fit_gram, cls.gp_lml_batch

  def gp_fit(self, *a, **k):
    calls['gp_fit'] += 1
    return orig_fit(self, *a, **k)

  def gp_fit_gram(self, *a, **k):
    calls['gp_fit_gram'] += 1
    return orig_gram(self, *a, **k)

  def gp_lml_batch(self, specs, *a, **k):
    calls['lml_batch_sizes'].append(len(specs))
    return orig_batch(self, specs, *a, **k)
  monkeypatch.setattr(cls, 'gp_fit', gp_fit)
  monkeypatch.setattr(cls, 'gp_fit_gram', gp_fit_gram)
  monkeypatch.setattr(cls, 'gp_lml_batch', gp_lml_batch)
  assert engine_mod.get_engine() is engine and 'oracle' not in type(engine).__module__
  engine.calls = calls
  return engine


def _with_install(run, **install_kwargs):
  from dragonfly_amd import install
  install.install(**install_kwargs)
  try:
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      return run()
  finally:
    install.uninstall()


@pytest.mark.parametrize('cfg', E.CONFIGS, ids=['%s-%s-%s-%d' % (c['kernel_type'], c['acq'], c['acq_opt_method'], i)
                                                for i, c in enumerate(E.CONFIGS)])
def test_reference_bandit_on_the_real_engine(cfg, real_engine):
  """ The eleven ask/tell configurations of the CPU plumbing test (random / tree-search / default
      maximisers, ML and posterior-sampling tuning, UCB / EI / PI / TTEI / TS / add-UCB). """
  want_points, want_hps = E._ask(cfg)                      # pylint: disable=protected-access
  assert want_hps[3].startswith('dragonfly.')
  got_points, got_hps = _with_install(lambda: E._ask(cfg))  # pylint: disable=protected-access
  assert got_hps[3].startswith('dragonfly_amd.')
  calls = real_engine.calls
  assert calls['gp_fit'] > 0 and len(calls['lml_batch_sizes']) > 0
  print('libdfhip calls:', calls['gp_fit'], 'fits,', len(calls['lml_batch_sizes']), 'tuning batches, largest',
        max(calls['lml_batch_sizes']))
  # the tuned hyper-parameters are candidates of the seeded search: the same candidate wins
  assert got_hps[0] == want_hps[0] and np.array_equal(got_hps[1], want_hps[1]) and got_hps[2] == want_hps[2]
  for got, want in zip(got_points, want_points):
    assert np.array_equal(got, want)


@pytest.mark.parametrize('mode,workers,extra', E.FULL_RUNS,
                         ids=['%s%d-%s-%d' % (m, w, e['acq'], i) for i, (m, w, e) in enumerate(E.FULL_RUNS)])
def test_reference_full_runs_on_the_real_engine(mode, workers, extra, real_engine):
  """ Whole optimisation runs with parallel workers: hallucinated in-progress points, sequential
      batches, additive GPs. """
  want = E._full_run(mode, workers, extra)                              # pylint: disable=protected-access
  got = _with_install(lambda: E._full_run(mode, workers, extra))        # pylint: disable=protected-access
  assert real_engine.calls['gp_fit'] > 0
  assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize('acq,method', [('ts', 'rand'), ('ucb', 'rand'), ('ucb', 'pdoo')])
def test_reference_multiobjective_bandit_on_the_real_engine(acq, method, real_engine):
  want = E._moo_run(acq, method)                                        # pylint: disable=protected-access
  got = _with_install(lambda: E._moo_run(acq, method))                  # pylint: disable=protected-access
  assert real_engine.calls['gp_fit'] > 0
  assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize('num_workers,acq', [(1, None), (3, 'ucb-ts')])
def test_reference_multifidelity_bandit_on_the_real_engine(num_workers, acq, real_engine):
  want_pts, want_fidels, _ = E._mf_run(num_workers, acq)                # pylint: disable=protected-access
  got_pts, got_fidels, got_mod = _with_install(lambda: E._mf_run(num_workers, acq), multi_fidelity=True)  # pylint: disable=protected-access
  assert got_mod.startswith('dragonfly_amd.') and real_engine.calls['gp_fit'] > 0
  assert np.array_equal(got_pts, want_pts) and np.array_equal(got_fidels, want_fidels)


def test_top_level_maximise_function_on_the_real_engine(real_engine):
  from dragonfly import maximise_function
  f = lambda x: -float((x[0] - 0.3) ** 2 + (x[1] + 0.2) ** 2) + 0.05 * float(np.cos(7 * x[0]))

  def run():
    np.random.seed(77)
    with warnings.catch_warnings():
      warnings.simplefilter('ignore')
      val, pt, history = maximise_function(f, [[-1, 1], [-1, 1]], 7)
    return val, np.array(pt), np.array(history.query_points)
  want = run()
  got = _with_install(run)
  assert real_engine.calls['gp_fit'] > 0
  assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
