"""GPU: the headline configuration (BASELINE configs 3 / 4: n = 16384, d = 32, SE-ARD) at FULL SIZE
against the oracle -- not by residuals: one oracle fit on the host (about 15 s of OpenBLAS on the
GPU box), then lml, alpha, posterior mean / std on all 65536 seed-203 candidates (config 3's candidate stage) and one joint
Thompson block of 4096 seed-204 candidates with the seed-304 normals, value by value.
Reference functions: gp/gp_core.py:155-190 (build_posterior, eval), :222-227 (lml), :250-254
(draw_samples), utils/general_utils.py:166-232."""
import numpy as np
import pytest

import bench_configs as BC
from conftest import relerr, relerr_elem
from oracle import ref_numpy as O

pytestmark = pytest.mark.gpu


def test_config3_config4_full_size_against_oracle(engine):
  from dragonfly_amd.engine import KernelSpec
  c = BC.config3()
  X, Y, mean_c, noise = c['X'], c['Y'], c['mean_c'], c['noise']
  assert X.shape == (16384, 32)
  og = O.GPOracle(X, Y, O.KernelSpec('se', 32, c['scale'], c['bw']), mean_c, noise)
  gp = engine.gp_fit(KernelSpec('se', 32, c['scale'], c['bw']), X, Y - mean_c, noise)
  assert gp.jitter_power == og.jitter_power
  assert abs(gp.lml - og.lml()) <= 1e-10 * abs(og.lml())
  assert relerr(gp.get_alpha(), og.alpha) < 1e-10
  # config 3's candidate stage at its stated size: the posterior at ALL 65536 seed-203 candidates;
  # the oracle in chunks of 4096 (its covariance is O(m^2); the diagonal is chunk-invariant)
  Xs_all = BC.config3_candidates(65536)
  mu_all, sd_all = gp.predict(Xs_all)
  mu_o_all, sd_o_all = og.eval_chunked(Xs_all, chunk=4096)
  assert relerr(mu_all + mean_c, mu_o_all) < 1e-10 and relerr(sd_all, sd_o_all) < 1e-10
  print('config 3 posterior over 65536 candidates: mu %.1e / %.1e, sd %.1e / %.1e (norm-wise / element-wise)'
        % (relerr(mu_all + mean_c, mu_o_all), relerr_elem(mu_all + mean_c, mu_o_all), relerr(sd_all, sd_o_all), relerr_elem(sd_all, sd_o_all)))
  # element-wise, over the entries not near zero (|b_i| >= 1e-3 max|b|): implied by the norm-wise bound / 1e-3
  assert relerr_elem(mu_all + mean_c, mu_o_all) < 1e-7 and relerr_elem(sd_all, sd_o_all) < 1e-7
  Xs = Xs_all[:4096]
  mu_o, sd_o = mu_o_all[:4096], sd_o_all[:4096]
  mu_d, sd_d = mu_all[:4096], sd_all[:4096]
  bv, bi, vals = gp.acq_argmax('ucb', Xs, params=(2.0, 0.0), mean_const=mean_c, return_vals=True)
  ucb_o = O.acq_values('ucb', mu_o, sd_o, 2.0)
  assert relerr(vals, ucb_o) < 1e-10 and bi == O.argmax_first(ucb_o)[1]
  # config 4: the first Thompson block of rank 0's shard (rows 0..4095 of the seed-204 set)
  B = BC.TS_BLOCK
  cands = np.random.RandomState(204).random_sample((B, 32))
  U = np.random.RandomState(304).standard_normal(B)
  draw_o = og.draw_samples_blocked(cands, U, B)
  _, cov_o = og.eval(cands, 'covar')
  _, pw_o = O.stable_cholesky(cov_o, return_power=True)
  tv, ti, draw_d, pw_d = gp.thompson(cands, U, block=B, mean_const=mean_c, return_samples=True)
  assert pw_d[0] == pw_o
  assert relerr(draw_d, draw_o) < 1e-10
  assert ti == int(np.argmax(draw_o)) and tv == draw_d[ti]
  # ... and the same block inside a longer shard gives the same draw (block boundaries are fixed)
  c2, U2 = np.random.RandomState(204).random_sample((2 * B, 32)), np.random.RandomState(304).standard_normal(2 * B)
  _, _, draw2, _ = gp.thompson(c2, U2, block=B, mean_const=mean_c, return_samples=True)
  assert np.array_equal(draw2[:B], draw_d)
  gp.free()
