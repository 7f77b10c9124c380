"""Config 2's candidate stage (n = 4096, 65536 candidates) with the GEMM launches timed one by one (dfh_ctx_gemm_profile):
run once with DFH_TRSM_FUSED=1 and once with 0 to see where the row solve's time goes in either form (the one-product
form is docs/experiments/r05_trsm_one_product.patch; without it both runs time the two-launch form)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench_configs as BC
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
c = BC.config2()
spec = KernelSpec('matern', c['d'], c['scale'], c['bw'], nu=c['nu'])
Xd, yd, cd = eng.to_device(c['X']), eng.to_device(c['Y'] - c['mean_c']), eng.to_device(c['cands'])
gp = eng.gp_fit(spec, Xd, yd, c['noise'])
for rep in range(3):
  eng.gemm_profile(True, fetch=False)
  eng.timings(True)
  r = gp.acq_argmax('ei', cd, params=(c['best'], 0.0), mean_const=c['mean_c'])
  eng.sync()
  s = eng.timings(False)
  prof = eng.gemm_profile(False)
  print('fused=%s rep %d: trsm section %.3f ms; gemm variants: %s' % (os.environ.get('DFH_TRSM_FUSED', '1'), rep, s['trsm'],
        json.dumps({i: {'launches': v['launches'], 'ms': round(v['ms'], 3), 'TF/s': round(v['flop'] / max(v['ms'], 1e-9) / 1e9, 1)} for i, v in enumerate(prof) if v['launches']})), flush=True)
