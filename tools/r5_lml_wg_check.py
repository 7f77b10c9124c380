"""Round 5: the one-workgroup-per-candidate tuning objective (chol.hip: lml_wg_kernel) against the NumPy oracle at
sizes around its tile edges, a candidate that needs the jitter ladder, then timings of large batches.
    python tools/r5_lml_wg_check.py [quick]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine, KernelSpec
from oracle import ref_numpy as O
eng = get_engine()
worst = 0.0
for n in (129, 130, 191, 192, 193, 200, 255, 256, 257, 320, 511, 512, 513, 1000, 1023, 1024, 1025, 2047):
  rs = np.random.RandomState(n)
  d = 5
  X = rs.rand(n, d); Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  Xd = eng.to_device(X)
  nb = 6
  specs, ospecs = [], []
  for c in range(nb):
    sc, bw = float(Y.var()) * (0.5 + rs.rand()), 0.3 + 0.6 * rs.rand(d)
    if c % 3 == 2:
      specs.append(KernelSpec('matern', d, sc, bw, nu=2.5)); ospecs.append(O.KernelSpec('matern', d, sc, bw, nu=2.5))
    else:
      specs.append(KernelSpec('se', d, sc, bw)); ospecs.append(O.KernelSpec('se', d, sc, bw))
  means = list(0.2 * rs.randn(nb)); noises = list(float(Y.var()) * np.exp(rs.uniform(np.log(0.002), np.log(0.2), nb)))
  got = eng.gp_lml_batch(specs, Xd, Y, means, noises)
  kc = nb if n <= 1100 else 2
  ref = [O.GPOracle(X, Y, ospecs[c], means[c], noises[c]).lml() for c in range(kc)]
  rel = max(abs(got[c] - ref[c]) / abs(ref[c]) for c in range(kc))
  worst = max(worst, rel)
  print('n=%5d  lml rel err max %.2e  %s' % (n, rel, 'OK' if rel <= 1e-10 else 'FAIL'), flush=True)
  Xd.free()
print('worst', worst)
# candidates that are not positive definite as they stand (duplicated rows, noise ~ 0) among good ones
n, d = 300, 3
rs = np.random.RandomState(7)
X = rs.rand(n, d); X[150:] = X[:150]; Y = np.sin(3 * X.sum(axis=1))
Xd = eng.to_device(X)
specs = [KernelSpec('se', d, 1.0, np.full(d, 0.5)) for _ in range(4)]
noises = [1e-2, 1e-17, 1e-3, 1e-17]
got, powers = eng.gp_lml_batch(specs, Xd, Y, [0.0] * 4, noises, return_powers=True)
ref = []
for c in range(4):
  og = O.GPOracle(X, Y, O.KernelSpec('se', d, 1.0, np.full(d, 0.5)), 0.0, noises[c])
  ref.append(og.lml())
print('ladder case: got', list(got), 'ref', ref, 'powers', powers)
Xd.free()
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
  sys.exit(0)
for (n, d, nbs) in ((200, 6, (8, 64, 256, 1024, 4096)), (500, 6, (8, 32, 64, 256, 1024)), (1000, 6, (4, 8, 16, 32, 64, 128, 256, 512, 2048)), (2000, 6, (8, 32, 64, 256, 512))):
  rs = np.random.RandomState(n)
  X = rs.rand(n, d); Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  Xd = eng.to_device(X)
  for nb in nbs:
    specs = [KernelSpec('se', d, float(Y.var()) * (0.5 + rs.rand()), 0.2 + 0.6 * rs.rand(d)) for _ in range(nb)]
    means = [0.0] * nb; noises = [float(Y.var() * 0.05)] * nb
    eng.gp_lml_batch(specs, Xd, Y, means, noises)
    ts = []
    for _ in range(3):
      eng.timings(True)
      t0 = time.perf_counter(); eng.gp_lml_batch(specs, Xd, Y, means, noises); ts.append((time.perf_counter() - t0) * 1e3)
      tm = eng.timings(False)
    print('n=%5d nb=%5d: %8.3f ms per call (%7.2f us per candidate)  sections %s' % (n, nb, sorted(ts)[1], sorted(ts)[1] * 1e3 / nb, {k: round(v, 3) for k, v in tm.items() if v}), flush=True)
  Xd.free()
