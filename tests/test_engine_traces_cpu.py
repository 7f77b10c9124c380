"""CPU: tests/golden/engine_trace_*.npz (the reference optimiser's calls on the engine object, recorded by
oracle/make_golden.py: gen_engine_traces) replayed against the NumPy stand-in engine they were recorded on: the
record / replay machinery of tests/engine_trace.py reproduces every call exactly.  The MI355X replay of the same
traces is tests/test_gpu_engine_traces.py."""
import glob
import os

import pytest

import engine_trace as ET

HERE = os.path.dirname(os.path.abspath(__file__))
TRACES = sorted(glob.glob(os.path.join(HERE, 'golden', 'engine_trace_*.npz')))


def test_all_25_configurations_are_there():
  assert len(TRACES) == 25


# (the three longest traces -- slice-sampled tuning, thousands of one-candidate batches -- take a minute each on
#  the NumPy stand-in: one of them stands for the three here)
@pytest.mark.parametrize('path', [p for p in TRACES if not any(s in p for s in ('ask_10', 'full_07', 'maximise_function'))],
                         ids=lambda p: os.path.basename(p)[13:-4])
def test_trace_replays_exactly_on_the_stand_in(path):
  from oracle_engine import OracleEngine
  calls, worst, meta = ET.replay(path, OracleEngine(), tol=0.0)
  assert calls == meta['events'] and worst == 0.0 and meta['reference_points_equal']


def _event_key(ev, arrays_of):
  """ What identifies a call: target handle, method, and the shapes of the arrays it carries (a stale trace differs in
      how many candidates a tuning batch holds long before it differs in values). """
  def shape(v):
    if isinstance(v, dict):
      if 'a' in v:
        return ('a',) + tuple(arrays_of[v['a']].shape)
      for k in ('l', 't'):
        if k in v:
          return (k,) + tuple(shape(x) for x in v[k])
      return tuple(sorted(v))
    return v
  return (ev['h'], ev['m'], tuple(shape(a) for a in ev['args']), tuple(sorted((k, shape(v)) for k, v in ev['kwargs'].items())))


def test_committed_trace_is_the_call_stream_of_this_tree():
  """ The traces are only evidence for the install() that recorded them: a change to what dragonfly_amd sends through
      the engine object (round 5 merged the slice sampler's two loops into one density call, and four committed traces
      went stale unnoticed) must come with re-recorded traces.  Where the reference tree is present, ONE post-sampling
      configuration (ask_09: slice-sampled hyper-parameters, ~10 s) is re-recorded and compared with the committed
      trace: the same number of engine calls, the same first 50 and last 10 calls (method, target, argument shapes),
      the same arrays bit for bit in those calls' arguments. """
  import sys
  ref = os.environ.get('DRAGONFLY_REFERENCE', '/root/reference')
  if not os.path.isdir(os.path.join(ref, 'dragonfly')):
    pytest.skip('needs the reference tree (build container only)')
  sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
  import make_golden as MG
  import numpy as np
  name, run, install_kwargs, _ = [s for s in MG.engine_trace_scenarios() if s[0].startswith('ask_09')][0]
  want, got, log = MG.record_engine_trace(run, install_kwargs)
  assert all(np.array_equal(g, w) for g, w in zip(got, want))
  rec, arrays = ET.load(os.path.join(HERE, 'golden', 'engine_trace_%s.npz' % name))
  assert len(log.events) == rec['meta']['events'] == len(rec['events']), \
      ('stale trace: re-record with `python oracle/make_golden.py engine_traces`', len(log.events), rec['meta']['events'])
  pick = list(range(50)) + list(range(len(log.events) - 10, len(log.events)))
  for k in pick:
    assert _event_key(log.events[k], log.arrays) == _event_key(rec['events'][k], arrays), k

  def arrs(v, table, out):
    if isinstance(v, dict):
      if 'a' in v:
        out.append(table[v['a']])
      for key in ('l', 't'):
        for x in v.get(key, []) if isinstance(v.get(key), list) else []:
          arrs(x, table, out)
    return out
  for k in pick:
    a, b = [], []
    for v in log.events[k]['args']:
      arrs(v, log.arrays, a)
    for v in rec['events'][k]['args']:
      arrs(v, arrays, b)
    assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b)), k
