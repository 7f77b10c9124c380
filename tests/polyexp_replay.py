"""Checks the polynomial and exponential-decay kernels (the two kernels of the reference that are
not stationary, dragonfly/gp/kernel.py:331-437) and the GPs built on them -- a plain GP with a
polynomial kernel, the multi-fidelity GP with the exponential-decay fidelity kernel
(euclidean_gp.py:881-887) -- against the outputs of the real reference
(tests/golden/poly_expdecay.npz, oracle/make_golden.py: gen_poly_expdecay_cases).  Shared by the
CPU plumbing test (stand-in engine) and the MI355X test."""
import numpy as np

from conftest import load_golden, relerr


def check(tol=1e-10, kernel_tol=1e-13):
  from dragonfly_amd import kernel as K
  from dragonfly_amd.gp_core import GP
  from dragonfly_amd.mf_gp import EuclideanMFGP
  g = load_golden('poly_expdecay')
  # kernel matrices
  for order in (1, 2, 3, 5):
    kern = K.PolyKernel(3, order, 1.7, g['scalings'])
    assert relerr(kern(g['X1'], g['X2']), g['poly%d_K12' % order]) < kernel_tol, order
    assert relerr(kern(g['X1']), g['poly%d_K11' % order]) < kernel_tol, order
  ed = K.ExpDecayKernel(2, float(g['ed_scale']), float(g['ed_offset']), g['powers'])
  assert relerr(ed(g['Z1'], g['Z2']), g['ed_K12']) < kernel_tol
  assert relerr(ed(g['Z1']), g['ed_K11']) < kernel_tol
  assert str(ed).startswith('ExpDec: sc=1.300, offset=0.210')
  assert str(K.PolyKernel(3, 3, 1.7, g['scalings'])) == 'Poly: d=3, scale=1.70, 0.80,1.30,0.45'
  # a GP with the polynomial kernel: its prior variance varies from point to point
  p_mean = float(g['p_mean'])
  pgp = GP(list(g['p_X']), list(g['p_Y']), K.PolyKernel(3, int(g['p_order']), float(g['p_scale']), g['p_scalings']),
           lambda x: np.array([p_mean] * len(x)), float(g['p_noise']))
  assert relerr(pgp.alpha, g['p_alpha']) < tol
  assert abs(pgp.compute_log_marginal_likelihood() - float(g['p_lml'])) <= tol * abs(float(g['p_lml']))
  mu, sd = pgp.eval(list(g['p_Xs']), 'std')
  assert relerr(mu, g['p_mu']) < tol and relerr(sd, g['p_sd']) < tol
  _, sdh = pgp.eval_with_hallucinated_observations(list(g['p_Xs']), list(g['p_Xh']), 'std')
  assert relerr(sdh, g['p_sdh']) < tol
  # the multi-fidelity GP: scale * ExpDecay(z) * SE(x)
  fd, dd = g['ZZ'].shape[1], g['XX'].shape[1]
  mean_c = float(g['mean_c'])
  mk = lambda n: EuclideanMFGP(list(g['ZZ'][:n]), list(g['XX'][:n]), list(g['YY'][:n]), None, float(g['scale']),
                               K.ExpDecayKernel(fd, 1.0, float(g['f_offset']), g['f_powers']),
                               K.SEKernel(dd, 1.0, g['dbw']), lambda x: np.array([mean_c] * len(x)),
                               float(g['noise']))
  gp = mk(len(g['YY']))
  assert not gp._generic                                  # the device evaluates the kernel itself
  assert relerr(gp.K_trtr_wo_noise, g['K']) < kernel_tol
  assert relerr(gp.L, g['L']) < tol and relerr(gp.alpha, g['alpha']) < tol
  assert abs(gp.compute_log_marginal_likelihood() - float(g['lml'])) <= tol * abs(float(g['lml']))
  mu, sd = gp.eval_at_fidel(list(g['Zs']), list(g['Xs']), 'std')
  assert relerr(mu, g['mu']) < tol and relerr(sd, g['sd']) < tol
  _, cov = gp.eval_at_fidel(list(g['Zs']), list(g['Xs']), 'covar')
  assert relerr(cov, g['cov']) < tol
  _, sdh = gp.eval_at_fidel_with_hallucinated_observations(list(g['Zs']), list(g['Xs']), list(g['Zh']),
                                                           list(g['Xh']), 'std')
  assert relerr(sdh, g['sdh']) < tol
  np.random.seed(77)
  sample = gp.draw_mf_samples(1, list(g['Zs']), list(g['Xs'])).ravel()
  # through the Cholesky factor of a posterior covariance (~sqrt(eps) conditioning): the bound is twice the
  # distance of the reference's own draw from the extended-precision draw on its mean and covariance
  from truth_bounds import draw_bound
  np.random.seed(77)
  U77 = np.random.normal(size=(len(g['mu']), 1)).ravel()
  sample_tol = draw_bound(g['mu'], g['cov'], U77, g['sample'])
  assert relerr(sample, g['sample']) <= sample_tol, (relerr(sample, g['sample']), sample_tol)
  # grown from 40 points by add_mf_data_multiple (block-row append of the factor)
  gp2 = mk(40)
  gp2.add_mf_data_multiple(list(g['ZZ'][40:]), list(g['XX'][40:]), list(g['YY'][40:]))
  assert relerr(gp2.alpha, g['alpha']) < tol
  return gp
