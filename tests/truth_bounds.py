"""Per-case parity bounds from the extended-precision truth (oracle/ld_truth.c).

north_star asks for 1e-10 relative agreement with the reference.  Where the reference's own double
arithmetic is further than that from the true value (Matern-0.5 diagonals, Thompson draws through
a covariance of condition ~1e9, tuned hyper-parameters with tiny noise) a literal 1e-10 cannot be
met by ANY correct implementation; the bound is then what tests/test_gpu_conditioning.py uses:

    bound = max(1e-10, 2 x err(reference value, truth))

computed per case and per quantity, never a blanket literal.  Only single SE / Matern kernels have
a truth; everything else is held to 1e-10 outright."""
import os

import numpy as np

from conftest import relerr

FLOOR = 1e-10
# How far from 1e-10 a bound may ever move.  The committed run (profiles/r04_truth_bounds_applied.json) relaxed
# bounds in two tests: 2.0e-3 for the log marginal likelihoods of Gram matrices that only factor with the
# stable_cholesky ladder's jitter (test_gpu_hp_tuning.py::test_lml_batch_jitter_ladder_per_candidate: the
# reference's own lml is 1.0e-3 from the extended-precision value there) and 4.2e-7 for Matern nu = 0.5 on a
# symmetric Gram matrix (test_gpu_golden.py, matern05_d2_n30).  A new case that needs more than this ceiling
# is a finding to look at, not a number to wave through.
CEILING = 5e-3

APPLIED = []      # (bound, test id) for every bound above the floor that a test of this run used


def bound(ref, truth, floor=FLOOR, factor=2.0):
  value = max(floor, factor * relerr(ref, truth))
  if value > floor:
    APPLIED.append((float(value), os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]))
  assert value <= CEILING, 'the reference is %.1e from the truth here: beyond the committed ceiling %.0e' % (value / factor, CEILING)
  return value


def applied_summary():
  """ What the run relaxed: the largest bound per test, largest first. """
  worst = {}
  for value, test in APPLIED:
    worst[test] = max(value, worst.get(test, 0.0))
  rows = sorted(worst.items(), key=lambda kv: -kv[1])
  return {'floor': FLOOR, 'ceiling': CEILING, 'bounds_above_floor': len(APPLIED), 'tests_with_relaxed_bounds': len(rows),
          'largest_bound_applied': rows[0][1] if rows else FLOOR, 'per_test_largest': [{'test': t, 'bound': v} for t, v in rows]}


def gp_case_bounds(kind, nu, bw, scale, X, Y, mean_c, noise, Xs, ref, best=None, ts_U=None, Xh=None):
  """ ref: dict of the reference's values, any of K, L, alpha, lml, mu, sd, cov, ei / ucb / pi / ttei
      (with their parameters as (acq, p0, p1) in ref['acq_params']), ts (the joint draw for ts_U), sd_h.
      Returns {name: bound}; names without a truth get FLOOR. """
  from oracle import ref_longdouble as T
  from oracle import ref_numpy as O
  out = {k: FLOOR for k in ref if k != 'acq_params'}
  if kind not in ('se', 'matern'):
    return out
  X, Xs = np.asarray(X, dtype=float), np.asarray(Xs, dtype=float)
  yc = np.asarray(Y, dtype=float) - mean_c
  best = float(np.max(Y)) if best is None else best
  ts_jit = 0.0
  if ts_U is not None and 'cov' in ref:
    _, pw = O.stable_cholesky(np.asarray(ref['cov']), return_power=True)
    ts_jit = 0.0 if pw is None else (10.0 ** pw) * float(np.diag(ref['cov']).max())
  tr = T.gp_truth(kind, bw, scale, X, yc, noise, Xs, mean_c, best, nu=nu or 0.0, want_L='L' in ref, want_K='K' in ref,
                  ts_normals=ts_U, ts_jitter=ts_jit, want_cov='cov' in ref)
  for name in ('alpha', 'mu', 'sd', 'L', 'K', 'cov'):
    if name in ref:
      out[name] = bound(ref[name], tr[name])
  if 'lml' in ref:
    out['lml'] = max(bound([ref['lml']], [tr['lml']]), out.get('alpha', FLOOR))      # (one number: see test_gpu_conditioning)
  if 'ts' in ref and ts_U is not None:
    out['ts'] = bound(ref['ts'], tr['draw'])
  for acq, p0, p1 in ref.get('acq_params', []):
    if acq in ref:
      out[acq] = bound(ref[acq], O.acq_values(acq, tr['mu'], tr['sd'], p0, p1))
  if 'sd_h' in ref and Xh is not None:
    Xa = np.concatenate([X, np.asarray(Xh, dtype=float)], axis=0)
    tra = T.gp_truth(kind, bw, scale, Xa, np.zeros(len(Xa)), noise, Xs, 0.0, 0.0, nu=nu or 0.0)
    out['sd_h'] = bound(ref['sd_h'], tra['sd'])
  return out


def gram_bounds(K, diag_add, y_centred, ref, K_cross=None, k_ss=None, K_tete=None, mean_const=0.0):
  """ The same for ANY kernel through the linear algebra on its (double) Gram matrices: bounds for the
      names in ref among alpha, lml, mu, sd, cov. """
  from oracle import ref_longdouble as T
  tr = T.gram_truth(K, diag_add, y_centred, K_cross, k_ss, K_tete, mean_const)
  out = {}
  for name in ref:
    if name == 'lml':
      out[name] = max(bound([ref['lml']], [tr['lml']]), bound(ref['alpha'], tr['alpha']) if 'alpha' in ref else FLOOR)
    elif name in tr:
      out[name] = bound(ref[name], tr[name])
    else:
      out[name] = FLOOR
  return out


def draw_bound(mu, cov, U, ref_draw):
  """ Bound for a joint Thompson draw mu + chol(cov + jitter I) U (general_utils.py:224-232) given the
      reference's own mean and covariance: twice the distance of its double-precision draw from the draw
      computed in extended precision from the same inputs (the jitter the ladder settles on included). """
  from oracle import ref_longdouble as T
  from oracle import ref_numpy as O
  cov = np.asarray(cov, dtype=float)
  _, pw = O.stable_cholesky(cov, return_power=True)
  jit = 0.0 if pw is None else (10.0 ** pw) * float(np.diag(cov).max())
  return bound(ref_draw, T.gaussian_draw(mu, cov, U, jit))


def kernel_draw_bound(kind, nu, bw, scale, X, y_centred, noise, Xs, mean_const, U, ref_draw, ref_cov):
  """ Bound for a joint Thompson draw when the posterior covariance is numerically singular (more
      candidates than training points): there the draw is sensitive to the rounding of the covariance
      ITSELF, so the truth starts from the kernel (oracle/ld_truth.c: covariance and draw in extended
      precision, with the jitter the reference's ladder settles on for ref_cov). """
  from oracle import ref_longdouble as T
  from oracle import ref_numpy as O
  _, pw = O.stable_cholesky(np.asarray(ref_cov, dtype=float), return_power=True)
  jit = 0.0 if pw is None else (10.0 ** pw) * float(np.diag(ref_cov).max())
  tr = T.gp_truth(kind, bw, scale, X, y_centred, noise, Xs, mean_const, 0.0, nu=nu or 0.0, ts_normals=U, ts_jitter=jit)
  return bound(ref_draw, tr['draw'])
