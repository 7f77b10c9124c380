"""GPU: hallucinated observations and synchronous batches at size (the golden fixtures pin them at
n <= 130, q <= 3).  Reference: GP.eval_with_hallucinated_observations (gp/gp_core.py:192-220: the
(n+q) x (n+q) matrix re-factored per call) and the sequential batch construction of the syn
acquisitions (opt/gpb_acquisitions.py:90-115: each earlier recommendation is an in-progress point of
the next).  Here n = 4096 (BASELINE config 2's GP), q grows to 7."""
import numpy as np
import pytest

import bench_configs as BC
from conftest import relerr, relerr_elem
from oracle import ref_numpy as O
from truth_bounds import gp_case_bounds

pytestmark = pytest.mark.gpu


def test_hallucinated_std_and_sequential_batch_at_n4096(engine):
  from dragonfly_amd.engine import KernelSpec
  c = BC.config2()
  X, Y, mean_c, noise = c['X'], c['Y'], c['mean_c'], c['noise']
  og = O.GPOracle(X, Y, O.KernelSpec('matern', c['d'], c['scale'], c['bw'], nu=c['nu']), mean_c, noise)
  gp = engine.gp_fit(KernelSpec('matern', c['d'], c['scale'], c['bw'], nu=c['nu']), X, Y - mean_c, noise)
  cands = np.ascontiguousarray(c['cands'][:4096])
  picks_d, picks_o = [], []
  for q in range(8):
    Xh_d = cands[picks_d] if picks_d else None
    bv, bi, ei_d = gp.acq_argmax('ei', cands, params=(c['best'], 0.0), mean_const=mean_c, X_halluc=Xh_d, return_vals=True)
    mu_d, sd_d = gp.predict(cands, X_halluc=Xh_d)
    if picks_o:
      mu_o, sd_o = og.eval_with_hallucinated_observations(cands, cands[picks_o], 'std')
    else:
      mu_o, sd_o = og.eval(cands, 'std')
    ei_o = O.acq_values('ei', mu_o, sd_o, c['best'])
    # the mean does not see the hallucinated points (gp_core.py:195); the std shrinks around them
    assert relerr(mu_d + mean_c, mu_o) < 1e-10
    tol = 1e-10
    if q in (1, 7):      # the bound from the extended-precision truth, at the first and the last step
      tol = gp_case_bounds('matern', c['nu'], c['bw'], c['scale'], X, Y, mean_c, noise, cands, dict(sd_h=sd_o),
                           Xh=cands[picks_o])['sd_h']
    assert relerr(sd_d, sd_o) < tol, (q, relerr(sd_d, sd_o), tol)
    assert relerr(ei_d, ei_o) < max(tol, 1e-10), (q, relerr(ei_d, ei_o))
    print('q=%d sd_h: %.1e norm-wise, %.1e element-wise; EI %.1e' % (q, relerr(sd_d, sd_o), relerr_elem(sd_d, sd_o), relerr(ei_d, ei_o)))
    picks_d.append(int(bi))
    picks_o.append(O.argmax_first(ei_o)[1])
    assert picks_d == picks_o and bv == ei_d[bi]
  # (with observation noise a hallucinated point keeps a positive std: the reference recommends the same
  #  candidate several times in this batch, and so does the device -- the choices above are equal step by step)
  gp.free()
