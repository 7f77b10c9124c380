"""How close to a tie are the choices of real runs?  (verdict r5, item 9; SURVEY.md section 8b "arg-max identical")

    DFH_GAP_LOG=/tmp/x.json python tools/argmax_gaps.py [--hartmann EVALS]      (build container: needs /root/reference)

Runs the 25 configurations of tests/test_install_end_to_end.py -- the UNMODIFIED reference optimiser with
dragonfly_amd.install() on the NumPy stand-in engine (tests/oracle_engine.py) -- with dragonfly_amd.gaplog recording the
relative margin of every decision a value makes: best against second-best acquisition value (oper_utils.py:73), the tree
search's choice of the next leaf (doo.py:127-187), a random-search batch's best log marginal likelihood
(gp_core.py:435-445), the slice sampler's `y < log p(x)` (sampling/slice.py:52-88).  Prints one JSON object: per
configuration and overall, how many decisions there were, the smallest margin, how many lie below 1e-12 / 1e-10 / 1e-8.
With --hartmann N also dragonfly.maximise_function on Hartmann6 for N evaluations (the run of profiles/r05_bo_wallclock.json).
The same instrument runs on the device through tools/bo_wallclock.py (DFH_GAP_LOG set)."""
import json
import os
import sys
import time
import warnings

os.environ.setdefault('DFH_GAP_LOG', '/tmp/argmax_gaps_last.json')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np                                   # noqa: E402
import make_golden as MG                             # noqa: E402
from dragonfly_amd import gaplog, install            # noqa: E402
from dragonfly_amd import euclidean_gp, general_utils, gp_core, gpb_acquisitions, kernel   # noqa: E402
from dragonfly_amd import engine as engine_mod       # noqa: E402
from oracle_engine import OracleEngine               # noqa: E402


def with_stand_in(run, install_kwargs):
  eng = OracleEngine()
  mods = (engine_mod, euclidean_gp, general_utils, gp_core, kernel)
  saved = [(m, m.get_engine) for m in mods]
  saved_dc = gpb_acquisitions.DEVICE_CANDIDATES
  for m in mods:
    m.get_engine = (lambda _e=eng: _e)
  gpb_acquisitions.DEVICE_CANDIDATES = False
  install.install(**install_kwargs)
  try:
    return run()
  finally:
    install.uninstall()
    for m, fn in saved:
      m.get_engine = fn
    gpb_acquisitions.DEVICE_CANDIDATES = saved_dc


def merge(total, part):
  for kind, rec in part.items():
    t = total.setdefault(kind, {'count': 0, 'min': float('inf'), 'below_1e-12': 0, 'below_1e-10': 0, 'below_1e-8': 0, 'exact_ties': 0})
    t['count'] += rec['count']
    t['min'] = min(t['min'], rec['min'])
    for k in ('below_1e-12', 'below_1e-10', 'below_1e-8', 'exact_ties'):
      t[k] += rec[k]


def main():
  out = {'what': __doc__.split('\n\n')[0], 'configurations': {}, 'all_25': {}}
  for name, run, install_kwargs, _ in MG.engine_trace_scenarios():
    gaplog.reset()
    t0 = time.time()
    with_stand_in(run, install_kwargs)
    rec = gaplog.summary()
    out['configurations'][name] = {k: {kk: vv for kk, vv in v.items() if kk != 'deciles'} for k, v in rec.items()}
    merge(out['all_25'], rec)
    print('%-40s %5.1f s  %s' % (name, time.time() - t0, {k: (v['count'], '%.1e' % v['min']) for k, v in rec.items()}), file=sys.stderr, flush=True)
  if '--hartmann' in sys.argv:
    evals = int(sys.argv[sys.argv.index('--hartmann') + 1])
    from dragonfly import maximise_function
    import dragonfly.utils.euclidean_synthetic_functions as S
    sf = S.get_syn_func_caller('hartmann6', noise_type='no_noise')
    obj = lambda x: float(sf.func(np.asarray(x, dtype=float)))
    bounds = [[float(a), float(b)] for a, b in sf.domain.bounds]

    def hart():
      np.random.seed(11)
      with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return maximise_function(obj, bounds, evals)
    gaplog.reset()
    t0 = time.time()
    with_stand_in(hart, {})
    out['hartmann6_%d_evaluations_stand_in' % evals] = gaplog.summary()
    print('hartmann6 %d evaluations: %.1f s' % (evals, time.time() - t0), file=sys.stderr, flush=True)
  gaplog.reset()
  print(json.dumps(out))


if __name__ == '__main__':
  main()
