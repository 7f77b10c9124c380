"""Optional instrument (off unless DFH_GAP_LOG names a file): how far from a tie the choices of a run are.

The drop-in promises the reference's values to 1e-10 and the reference's arg-max; a second implementation of the same
mathematics can only keep the second promise where the best and the second-best value of a choice lie further apart than
the two implementations do (measured: <= 3.9e-11 relative over the 34 990 engine calls of tests/golden/engine_trace_*,
typically 1e-13 .. 1e-15).  Every place where a value decides something records the relative margin of that decision:

  acq_argmax / thompson   best against second-best acquisition value / joint-draw value over the candidate set
                          (dragonfly/utils/oper_utils.py:73: obj_vals.argmax())
  doo_expand              the leaf that is expanded next against the runner-up in the heap (utils/doo.py:127-187)
  doo_best / pdoo_run     the leaf / the run that is returned (doo.py:222-233, 250-256)
  hp_batch                best against second-best log marginal likelihood of a random-search batch (gp_core.py:435-445)
  slice_compare           a slice sampler's `y < log p(x)` (sampling/slice.py:52-88)

At exit one JSON object per kind goes to the file: count, the smallest margin, how many margins lie below 1e-12 / 1e-10 /
1e-8, and the decile margins.  tools/argmax_gaps.py runs the 25 configurations of tests/test_install_end_to_end.py and a
long Hartmann6 run under it."""
import atexit
import json
import os

import numpy as np

_PATH = os.environ.get('DFH_GAP_LOG')
ENABLED = bool(_PATH)
_records = {}
_TINY = 1e-300


def _add(kind, gap):
  _records.setdefault(kind, []).append(float(gap))


def top2(kind, values):
  """ margin of an arg-max over `values`: (best - second best) / |best|; NaNs and infinities do not compete """
  if not ENABLED:
    return
  v = np.asarray(values, dtype=float).ravel()
  v = v[np.isfinite(v)]
  if v.size < 2:
    return
  part = np.partition(v, v.size - 2)
  best, second = part[-1], part[-2]
  _add(kind, (best - second) / max(abs(best), _TINY))


def pair(kind, a, b):
  """ margin of a comparison a < b: |a - b| / max(|a|, |b|) """
  if not ENABLED:
    return
  a, b = float(a), float(b)
  if not (np.isfinite(a) and np.isfinite(b)):
    return
  _add(kind, abs(a - b) / max(abs(a), abs(b), _TINY))


def summary():
  out = {}
  for kind, gaps in sorted(_records.items()):
    g = np.sort(np.asarray(gaps))
    out[kind] = {'count': int(g.size), 'min': float(g[0]), 'below_1e-12': int(np.sum(g < 1e-12)),
                 'below_1e-10': int(np.sum(g < 1e-10)), 'below_1e-8': int(np.sum(g < 1e-8)),
                 'exact_ties': int(np.sum(g == 0.0)),
                 'deciles': [float(g[min(g.size - 1, int(q * g.size / 10))]) for q in range(10)]}
  return out


def reset():
  _records.clear()


def _dump():
  if ENABLED and _records:
    with open(_PATH, 'w') as f:
      json.dump(summary(), f, indent=1)


atexit.register(_dump)
