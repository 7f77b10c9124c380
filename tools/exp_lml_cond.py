"""Experiment: where does the lml error come from at cond ~ 1e13 (SE d=2, noise 1e-10 Var)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_numpy as O, ref_longdouble as T
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
N, d = 2048, 2
for seed in range(5):
  rs = np.random.RandomState(seed)
  X = rs.random_sample((N, d)); Y = (X ** 2).dot([0.5, 1.0]) + np.sin(3 * X[:, 0]) + 0.01 * rs.randn(N)
  bw = 0.3 * np.ones(d); scale = float(Y.var()); mean_c = float(np.median(Y))
  for frac in (1e-8, 1e-10):
    noise = frac * scale
    og = O.GPOracle(X, Y, O.KernelSpec('se', d, scale, bw), mean_c, noise)
    if og.jitter_power is not None:
      print(seed, frac, 'jitter', og.jitter_power); continue
    gp = eng.gp_fit(KernelSpec('se', d, scale, bw), X, Y - mean_c, noise)
    tr = T.gp_truth('se', bw, scale, X, Y - mean_c, noise, want_L=True)
    yc = Y - mean_c
    ld_t = np.log(np.diag(tr['L']).astype(np.longdouble)).sum()
    dot_t = (yc.astype(np.longdouble) * tr['alpha'].astype(np.longdouble)).sum()
    Ld = gp.get_L(); ad = gp.get_alpha()
    ld_d = np.log(np.diag(Ld)).sum(); dot_d = yc.dot(ad)
    ld_o = np.log(np.diag(og.L)).sum(); dot_o = yc.dot(og.alpha)
    print('seed %d noise %.0e  lml: dev-tru %.2e orc-tru %.2e | logdet dev %.2e orc %.2e | dot dev %.2e orc %.2e | lml %.6g logdet %.6g dot %.6g | steps %s'
          % (seed, frac, abs(gp.lml - tr['lml']) / abs(tr['lml']), abs(og.lml() - tr['lml']) / abs(tr['lml']),
             abs(ld_d - ld_t) / abs(tr['lml']), abs(ld_o - ld_t) / abs(tr['lml']), abs(dot_d - dot_t) / 2 / abs(tr['lml']),
             abs(dot_o - dot_t) / 2 / abs(tr['lml']), tr['lml'], float(ld_t), float(dot_t), gp.refine_steps()))
    gp.free()
