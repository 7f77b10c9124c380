"""Symmetric Gram matrix of an additive kernel (BASELINE config 5's shape: n = 4096, d = 100, 20 groups
of 5, SE) and of a product kernel: time per build, and the result against the generic multi-part
kernel (DFH_KM_SYMMULTI=0 in a second process).   python tools/time_addgram.py [out.npy]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dragonfly_amd.engine import Engine, KernelSpec   # noqa: E402

eng = Engine()
rs = np.random.RandomState(0)
for n, d, gsz, kind in ((4096, 100, 5, 'se'), (4096, 100, 5, 'matern'), (4000, 12, 3, 'se'), (16384, 40, 4, 'se')):
  groups = [list(range(i, min(d, i + gsz))) for i in range(0, d, gsz)]
  for comb in ('additive', 'product'):
    if comb == 'product' and d > 12:
      continue
    spec = KernelSpec(comb, d, 1.7, groups=groups, sub_kinds=[kind] * len(groups), sub_scales=[1.0] * len(groups),
                      sub_nus=[2.5] * len(groups), sub_bandwidths=[0.5 + rs.random_sample(len(g)) for g in groups])
    X = eng.to_device(rs.random_sample((n, d)))
    K = eng.empty((n, n))
    for _ in range(2):
      eng.kernel_matrix(spec, X, None, diag_add=0.1, out=K)
    eng.sync()
    t = time.time()
    for _ in range(10):
      eng.kernel_matrix(spec, X, None, diag_add=0.1, out=K)
    eng.sync()
    ms = (time.time() - t) * 100
    Kh = K.download()
    print('%s %s n=%d d=%d groups of %d: %.3f ms  (%.2f TB/s of output)  sym err %.1e  checksum %.17g' %
          (comb, kind, n, d, gsz, ms, n * n * 8 / ms / 1e9, np.abs(Kh - Kh.T).max(), Kh.sum()))
    X.free(); K.free()
