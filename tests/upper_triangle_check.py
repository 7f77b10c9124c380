"""Run by tests/test_gpu_upper_triangle_unread.py in a subprocess (the switches are read per process): a fit at n >= 2048
and everything that reads its factor afterwards -- alpha, lml, GP.eval, the hallucinated posterior, a joint Thompson
block, an append -- printed as a checksum-free dump to a file for comparison across switch settings."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from dragonfly_amd.engine import get_engine, KernelSpec   # noqa: E402

n, out = int(sys.argv[1]), sys.argv[2]
eng = get_engine()
rs = np.random.RandomState(n)
d = 5
X = rs.rand(n + 1, d)
Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n + 1)
spec = KernelSpec('matern', d, float(Y.var()), 0.4 * np.ones(d), nu=2.5)
noise = float(Y.var() / 30)
gp = eng.gp_fit(spec, X[:n], Y[:n] - 0.1, noise)
Xs = rs.rand(700, d)
mu, sd = gp.predict(Xs)
_, sdh = gp.predict(Xs, X_halluc=rs.rand(3, d))
U = rs.randn(700)
val, idx, samp, _ = gp.thompson(Xs, U, block=350, mean_const=0.1, return_samples=True)
ext = gp.append(X[n:n + 1], Y[:n + 1] - 0.1)
np.savez(out, L=np.tril(gp.get_L()), alpha=gp.get_alpha(), lml=gp.lml, mu=mu, sd=sd, sdh=sdh, samp=samp, idx=idx,
         ext_alpha=ext.get_alpha(), ext_lml=ext.lml)
print('OK')
