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
