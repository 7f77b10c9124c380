"""The batched-frontier tree search (dragonfly_amd/doo.py) against the REAL reference's PDOO
(tests/golden/pdoo_cases.npz from oracle/make_golden.py: dragonfly/utils/doo.py through
oper_utils.pdoo_maximise): same value, same point, same sequence of queried points -- whether the
values are fetched one per call or a frontier per call.  The GPU part drives the device GP through
the acquisition mirrors with acq_opt_method 'pdoo' / 'direct'."""
from argparse import Namespace

import numpy as np
import pytest

from conftest import load_golden
from oracle.test_objectives import PDOO_CASES


@pytest.mark.parametrize('name,fn,bounds,budget', PDOO_CASES, ids=[c[0] for c in PDOO_CASES])
def test_pdoo_visits_what_the_reference_visits(name, fn, bounds, budget):
  from dragonfly_amd.doo import pdoo_maximise, pdoo_maximise_batched
  g = load_golden('pdoo_cases')
  rows = lambda pts: np.array([fn(p) for p in pts])
  runs = [pdoo_maximise(fn, bounds, budget, return_history=True)]
  for frontier, depth in ((0, 0), (8, 1), (32, 2), (200, 3)):
    runs.append(pdoo_maximise_batched(rows, bounds, budget, frontier=frontier, depth=depth, return_history=True))
  for val, pt, hist in runs:
    assert val == float(g[name + '_val'])
    assert np.array_equal(pt, g[name + '_pt'])
    assert np.array_equal(np.array(hist.query_points), g[name + '_queries'])
  one_by_one, batched = runs[1][2], runs[3][2]
  # the reference makes one callback per query; the cache alone removes the repeats of the PDOO
  # restarts, the frontier most of the remaining calls
  assert one_by_one.points_requested >= budget
  assert one_by_one.device_calls == one_by_one.points_evaluated <= one_by_one.points_requested / 3
  assert batched.device_calls * 3 <= one_by_one.device_calls
  assert batched.points_evaluated <= 4 * one_by_one.points_evaluated + 400


def test_pdoo_minimise_and_objective_shape_errors():
  from dragonfly_amd.doo import pdoo_minimise, pdoo_maximise_batched
  val, pt, _ = pdoo_minimise(lambda x: float(np.sum((np.asarray(x) - 0.25) ** 2)), [[0, 1]] * 2, 200)
  assert 0 <= val < 0.05 and np.all(np.abs(pt - 0.25) < 0.15)
  with pytest.raises(ValueError):
    pdoo_maximise_batched(lambda pts: np.zeros(len(pts) + 1), [[0, 1]] * 2, 100)


# ---------------------------------------------------------------------------------------------
def _gp_and_anc(case, method, max_evals=300, in_progress=()):
  from dragonfly_amd.gp_core import GP, ConstantMean
  from dragonfly_amd.oper_utils import EuclideanDomain
  from golden_kernels import device_kernel
  g = load_golden('gp_' + case)
  gp = GP(list(g['X']), list(g['Y']), device_kernel(g), ConstantMean(float(g['mean_c'])), float(g['noise']))
  d = g['X'].shape[1]
  bounds = np.array([[0.0, 1.0]] * d)
  anc = Namespace(max_evals=max_evals, t=len(g['Y']), domain=EuclideanDomain(bounds),
                  curr_max_val=float(g['Y'].max()), eval_points_in_progress=list(in_progress),
                  acq_opt_method=method, handle_parallel='halluc', is_mf=False, domain_bounds=bounds)
  return g, gp, anc


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['se_d2_n40', 'matern25_d6_n60'])
def test_gp_acquisitions_maximised_by_pdoo_match_reference(engine, case, monkeypatch):
  from dragonfly_amd import gpb_acquisitions as A
  ref = load_golden('pdoo_cases')
  for method, acq, with_halluc in (('pdoo', 'ucb', False), ('pdoo', 'ei', False), ('direct', 'ucb', False),
                                   ('pdoo', 'pi', True)):
    for frontier in (32, 0):
      monkeypatch.setattr(A, 'PDOO_FRONTIER', frontier)
      g, gp, anc = _gp_and_anc(case, method)
      if with_halluc:
        anc.eval_points_in_progress = [g['Xh'][0]]
      pt = np.asarray(getattr(A.asy, acq)(gp, anc))
      assert np.array_equal(pt, ref['%s_%s_%s' % (case, method, acq)]), (method, acq, frontier)


@pytest.mark.gpu
def test_pdoo_frontier_cuts_device_calls_and_keeps_the_answer(engine):
  from dragonfly_amd.doo import pdoo_maximise_batched
  g, gp, anc = _gp_and_anc('se_ard_d5_n50', 'pdoo')
  beta = 2.0
  def ucb(pts):
    mu, sd = gp.eval(pts, 'std')
    return mu + beta * sd
  v0, p0, h0 = pdoo_maximise_batched(ucb, anc.domain.bounds, 1000, frontier=0, depth=0, return_history=True)
  v1, p1, h1 = pdoo_maximise_batched(ucb, anc.domain.bounds, 1000, frontier=32, depth=2, return_history=True)
  assert v0 == v1 and np.array_equal(p0, p1)
  assert np.array_equal(np.array(h0.query_points), np.array(h1.query_points))
  assert h1.device_calls * 4 <= h0.device_calls


@pytest.mark.gpu
@pytest.mark.parametrize('kt,method', [('se', 'pdoo'), ('matern', 'direct')])
def test_fitter_tree_search_tuning_picks_reference_hyperparameters(engine, kt, method):
  """ EuclideanGPFitter with ml_hp_tune_opt 'pdoo' / 'direct' (gp_core.py:427-434, 463-468): the
      tree search sees the log marginal likelihoods a frontier per dfh_gp_lml_batch call and ends
      at the hyper-parameters the reference's one-fit-per-callback search ends at. """
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  ref, data = load_golden('pdoo_cases'), load_golden('fitter_d3_n45')
  opts = Namespace(kernel_type=kt, ml_hp_tune_opt=method, hp_tune_max_evals=250, hp_tune_criterion='ml')
  results = []
  for frontier in (32, 0):
    fitter = EuclideanGPFitter(list(data['X']), list(data['Y']), options=opts)
    fitter.pdoo_frontier = frontier
    kind, gp, hps = fitter.fit_gp()
    assert kind == 'fitted_gp'
    results.append((np.array(hps[0], dtype=float), gp.compute_log_marginal_likelihood()))
  for cts_hps, lml in results:
    assert np.array_equal(cts_hps, ref['fit_%s_%s_cts_hps' % (kt, method)])
    want = float(ref['fit_%s_%s_lml' % (kt, method)])
    assert abs(lml - want) <= 1e-9 * abs(want)


@pytest.mark.gpu
def test_add_ucb_with_tree_search_matches_reference(engine):
  """ asy.add_ucb with acq_opt_method='pdoo': one tree search per additive group over the group's
      box (gpb_acquisitions.py:159-183), group posteriors from dfh_gp_add_ucb_group. """
  from dragonfly_amd import gpb_acquisitions as A
  ref = load_golden('pdoo_cases')
  g, gp, anc = _gp_and_anc('additive_d10_n80', 'pdoo', max_evals=800)
  assert np.array_equal(np.asarray(A.asy.add_ucb(gp, anc)), ref['additive_d10_n80_pdoo_add_ucb'])
  assert anc.max_evals == 800
