"""Run by tests/test_gpu_lml_wg.py in a subprocess with the schedule switches of the tuning objective in the
environment (they are read once per process): batched log marginal likelihoods against the oracle at sizes around
the tile edges, for a group that gets a team of workgroups per candidate and one that does not; prints the
context's fall-back counter and OK."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from dragonfly_amd.engine import get_engine, KernelSpec   # noqa: E402
from oracle import ref_numpy as O                          # noqa: E402

eng = get_engine()
worst = 0.0
# (n <= 128 reaches the one-workgroup form only when the one-launch small-problem kernel is off or does not apply)
for n, nb in ((1, 2), (2, 1), (17, 1), (45, 2), (63, 5), (63, 20), (64, 3), (100, 17), (127, 4), (130, 3), (150, 2), (191, 3), (200, 1), (255, 4), (257, 5), (448, 2), (641, 9), (1000, 4), (1000, 70), (1600, 3), (300, 300)):
  rs = np.random.RandomState(7 * n + nb)
  d = 4
  X = rs.rand(n, d)
  Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  yv = float(Y.var()) if n > 1 else 1.0
  specs, ospecs = [], []
  for c in range(nb):
    sc, bw = yv * (0.5 + rs.rand()), 0.3 + 0.6 * rs.rand(d)
    if c % 3 == 2:
      specs.append(KernelSpec('matern', d, sc, bw, nu=2.5)); ospecs.append(O.KernelSpec('matern', d, sc, bw, nu=2.5))
    else:
      specs.append(KernelSpec('se', d, sc, bw)); ospecs.append(O.KernelSpec('se', d, sc, bw))
  means = list(0.2 * rs.randn(nb))
  noises = list(yv * np.exp(rs.uniform(np.log(0.002), np.log(0.2), nb)))
  got = eng.gp_lml_batch(specs, X, Y, means, noises)
  for c in sorted(set([0, nb // 2, nb - 1])):
    ref = O.GPOracle(X, Y, ospecs[c], means[c], noises[c]).lml()
    rel = abs(got[c] - ref) / abs(ref)
    worst = max(worst, rel)
    assert rel <= 1e-10, (n, nb, c, got[c], ref)
# a candidate that needs the ladder inside a team group
n, d = 300, 2
rs = np.random.RandomState(3)
X = rs.rand(n, d); X[150:] = X[:150]
Y = np.cos(3 * X[:, 0]) + X[:, 1]
specs = [KernelSpec('se', d, 1.0, np.full(d, b)) for b in (0.3, 2.0, 0.5)]
lml, powers = eng.gp_lml_batch(specs, X, Y, None, [1e-3, 1e-18, 1e-2], return_powers=True)
og = O.GPOracle(X, Y, O.KernelSpec('se', d, 1.0, np.full(d, 2.0)), 0.0, 1e-18)
assert powers[0] is None and powers[2] is None and powers[1] == og.jitter_power, powers
# ... and inside the small-problem forms (one 64 x 64 tile; the one-launch small group): the reference's jitter power
for n in (40, 100):
  rs = np.random.RandomState(5 + n)
  X = rs.rand(n, d); X[n // 2:] = X[:n // 2]
  Y = np.cos(3 * X[:, 0]) + X[:, 1]
  lml, powers = eng.gp_lml_batch(specs, X, Y, None, [1e-3, 1e-18, 1e-2], return_powers=True)
  og = O.GPOracle(X, Y, O.KernelSpec('se', d, 1.0, np.full(d, 2.0)), 0.0, 1e-18)
  assert powers[0] is None and powers[2] is None and powers[1] == og.jitter_power, (n, powers, og.jitter_power)
  assert np.isfinite(lml[1])       # (a jittered, numerically singular system: the power is the contract; its value is held to computed bounds in tests/test_gpu_lml_wg.py)
  for c, (b, s2) in enumerate(((0.3, 1e-3), (2.0, 1e-18), (0.5, 1e-2))):
    if c != 1:
      ref = O.GPOracle(X, Y, O.KernelSpec('se', d, 1.0, np.full(d, b)), 0.0, s2).lml()
      assert abs(lml[c] - ref) <= 1e-10 * abs(ref), (n, c, lml[c], ref)
print('worst %.2e fallbacks %d' % (worst, eng.counters()['chol_fallbacks']))
print('OK')
