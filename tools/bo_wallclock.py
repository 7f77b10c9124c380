"""Wall-clock of a REAL Dragonfly run, reference vs dragonfly_amd.install()  (verdict r4 item 5b):
    DRAGONFLY_REFERENCE=/path/to/checkout python tools/bo_wallclock.py <evals> <ref|install> [seed]
dragonfly.maximise_function (apis/opt.py:138 -> opt/gp_bandit.py:957) on Hartmann6 (6-d, the function of BASELINE
config 2) with default options; the time is split by spies on the seams: model building (GPFitter.fit_gp_for_gp_bandit
-- hyper-parameter tuning + fits, gp_core.py:427-499), next-GP draws (get_next_gp), acquisition (gpb_acquisitions.asy/syn
callables, gp_bandit.py:490,651), everything else (the optimiser's own Python).  Prints one JSON line."""
import json
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle.make_golden import import_reference

evals, mode = int(sys.argv[1]), sys.argv[2]
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
import_reference()
from dragonfly import maximise_function
import dragonfly.opt.gp_bandit as GB
import dragonfly.opt.gpb_acquisitions as A

import dragonfly.utils.euclidean_synthetic_functions as S
_sf = S.get_syn_func_caller('hartmann6', noise_type='no_noise')       # euclidean_synthetic_functions.py:16-49 on [0, 1]^6
objective = lambda x: float(_sf.func(np.asarray(x, dtype=float)))
domain_bounds = [[float(a), float(b)] for a, b in _sf.domain.bounds]

split = {'fit': 0.0, 'next_gp': 0.0, 'acq': 0.0}
counts = {'fit': 0, 'next_gp': 0, 'acq': 0}


def timed(key, fn):
  def w(*a, **k):
    t0 = time.perf_counter()
    try:
      return fn(*a, **k)
    finally:
      split[key] += time.perf_counter() - t0
      counts[key] += 1
  return w


if mode == 'install':
  from dragonfly_amd import install
  from dragonfly_amd.engine import get_engine
  get_engine()                                  # context creation is not part of the run
  install.install()
fitter_cls = GB.EuclideanGPFitter               # (after install: the batched subclass)
fitter_cls.fit_gp_for_gp_bandit = timed('fit', fitter_cls.fit_gp_for_gp_bandit)
fitter_cls.get_next_gp = timed('next_gp', fitter_cls.get_next_gp)
for ns in (A.asy, A.syn, A.seq):
  for name in ('ucb', 'ei', 'pi', 'ttei', 'ts', 'add_ucb'):
    if hasattr(ns, name):
      setattr(ns, name, timed('acq', getattr(ns, name)))
np.random.seed(seed)
prof = None
if os.environ.get('BO_PROFILE'):
  import cProfile
  prof = cProfile.Profile()
  prof.enable()
t0 = time.perf_counter()
with warnings.catch_warnings():
  warnings.simplefilter('ignore')
  val, pt, history = maximise_function(objective, domain_bounds, evals)
wall = time.perf_counter() - t0
if prof is not None:
  import pstats
  prof.disable()
  with open(os.environ['BO_PROFILE'], 'w') as fh:
    pstats.Stats(prof, stream=fh).sort_stats('cumulative').print_stats(45)
    pstats.Stats(prof, stream=fh).sort_stats('tottime').print_stats(30)
if os.environ.get('BO_POINTS'):
  np.save(os.environ['BO_POINTS'], np.array(history.query_points))
out = {'mode': mode, 'evals': evals, 'seed': seed, 'wall_s': round(wall, 3), 'max_val': float(val),
       'split_s': {k: round(v, 3) for k, v in split.items()}, 'host_other_s': round(wall - sum(split.values()), 3),
       'calls': counts, 'points_head': [[float(v) for v in p] for p in np.array(history.query_points)[:3]],
       'points_checksum': float(np.sum(np.array(history.query_points) * np.arange(1, 7))),
       'cpu_count': os.cpu_count()}
if mode == 'install':
  from dragonfly_amd import gaplog
  if gaplog.ENABLED:          # DFH_GAP_LOG set: the margins of this run's decisions (dragonfly_amd/gaplog.py)
    out['argmax_gaps'] = gaplog.summary()
print(json.dumps(out))
