"""CPU: the measurement code's own arithmetic (bench.py) -- the error norms it reports, the fp64-pipe bound, and the
two CPU legs (the oracle's port and, where a Dragonfly checkout exists, the reference's own functions) agreeing
bit for bit stage by stage."""
import os

import numpy as np
import pytest

import bench
import bench_configs as BC


def test_error_norms():
  b = np.array([1.0, 1e-6, -2.0, 0.0])
  a = b + np.array([1e-12, 1e-12, 0.0, 1e-12])
  assert bench.rel(a, b) == pytest.approx(1e-12 / 2.0)
  # element-wise: only entries with |b| >= 1e-3 max|b| count (1e-6 and 0 are masked)
  assert bench.rel_elem(a, b) == pytest.approx(1e-12)
  assert bench.both(a, b) == [bench.rel(a, b), bench.rel_elem(a, b)]
  assert bench.rel_elem([], []) == 0.0 and bench.rel_elem([0.0], [0.0]) == 0.0


def test_fp64_pipe_bound():
  r = bench.fp64_pipe_frac('se', 32, 1e9, 1, 1.0)
  # 2 * 32 flop per element on the matrix cores, 28 vector instructions per element
  assert r['min_ms_mfma'] == pytest.approx(2 * 32 * 1e9 / 78.6e12 * 1e3, rel=1e-3)
  assert r['min_ms_valu'] == pytest.approx(28 * 1e9 / 39.3e12 * 1e3, rel=1e-3)
  assert r['frac_of_fp64_pipe_peak'] == pytest.approx((r['min_ms_mfma'] + r['min_ms_valu']) / 1.0, rel=1e-3)


# (n = 2048 with 1100 candidates: OpenBLAS / LAPACK take their blocked, multi-threaded code paths there -- dpotrf's
#  recursive panels, dtrsm / dgemm blocking -- which n = 384 does not reach: `kind: port` is pinned to the reference's
#  own functions at a size where the library's blocking matters)
@pytest.mark.parametrize('n,m', [(384, 128), (2048, 1100)])
def test_cpu_legs_agree_stage_by_stage(monkeypatch, n, m):
  ref = os.environ.get('DRAGONFLY_REFERENCE', '/root/reference')
  if not os.path.isdir(os.path.join(ref, 'dragonfly')):
    pytest.skip('needs the reference tree (build container only)')
  prob = BC.config3()
  X, Y = prob['X'][:n], prob['Y'][:n]
  cands = np.random.RandomState(0).rand(m, BC.DIM)
  U = np.random.RandomState(1).randn(m)
  monkeypatch.setenv('DRAGONFLY_REFERENCE', ref)
  fr, kr, kind_r, _ = bench.cpu_functions(prob)
  monkeypatch.setenv('DRAGONFLY_REFERENCE', '')
  fp, kp, kind_p, _ = bench.cpu_functions(prob)
  assert (kind_r, kind_p) == ('reference', 'port')
  _, a = bench.oracle_stages(fr, kr, X, Y, prob['mean_c'], prob['noise'], [cands], [U], posterior_chunks=[cands])
  _, b = bench.oracle_stages(fp, kp, X, Y, prob['mean_c'], prob['noise'], [cands], [U], posterior_chunks=[cands])
  assert np.array_equal(a['alpha'], b['alpha']) and a['lml'] == b['lml']
  for key in ('mu', 'sd', 'draw'):
    assert np.array_equal(a['blocks'][0][key], b['blocks'][0][key])
  assert a['blocks'][0]['argmax'] == b['blocks'][0]['argmax']
  assert np.array_equal(a['posterior'][0]['sd'], b['posterior'][0]['sd'])


def test_contract_line_carries_the_side_targets_inside_roofline():
  """ The last stdout line of bench.py is the contract's JSON line cut from the full record: short enough to survive a
      tail, every side target a scalar INSIDE `roofline` (a record that keeps roofline's scalars keeps them). """
  import json
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  full = None
  for ln in open(os.path.join(root, 'profiles', 'r05_bench.json')):
    if ln.startswith('{"metric'):
      full = json.loads(ln)
  line = bench.contract_line(full)
  text = json.dumps(line)
  assert len(text) < 4000
  for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
    assert key in line
  r = line['roofline']
  assert all(not isinstance(v, (dict, list)) for v in r.values())
  assert r['cholesky_frac_of_fp64_mfma_peak'] == full['roofline']['side_targets']['cholesky_frac_of_fp64_mfma_peak']
  assert r['kernel_matrix_frac_by_section8d_bytes'] == full['roofline']['side_targets']['kernel_matrix_frac_by_section8d_bytes']
  assert r['chol_ms'] == full['chol_ms'] and r['solve_ms'] == full['solve_ms'] and r['comm_ranks_formed'] == 1
  assert r['C2_chol_ms'] == full['configs']['C2']['sections_ms']['chol']
  assert set(line['cpu_baseline']) == {'value', 'unit', 'cores', 'kind', 'sample'}
  assert 'workload' in line['config'] and 'model' not in line['config']
