"""Extended-precision truth for the GP posterior (oracle/ld_truth.c through ctypes).

TEST INFRASTRUCTURE ONLY.  Used by the conditioning tests to adjudicate between the device and
the NumPy oracle: for each quantity the test reports device-vs-oracle, device-vs-truth and
oracle-vs-truth.  The truth follows the mathematics of
dragonfly/gp/gp_core.py:155-190,222-227 and dragonfly/gp/kernel.py:171-181,242-299 in x87 long
double (eps 1.1e-19) with direct-difference distances; see the header of ld_truth.c.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
  sys.path.insert(0, _HERE)
import build_truth      # noqa: E402  pylint: disable=wrong-import-position

_lib = None


def _load():
  global _lib
  if _lib is None:
    lib = C.CDLL(build_truth.build())
    lib.ld_gp_truth.restype = C.c_int64
    lib.ld_gp_truth.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                C.c_int, C.c_double, C.c_void_p, C.c_int64, C.c_double, C.c_double] + [C.c_void_p] * 7 + \
                               [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    lib.ld_kernel_matrix.restype = C.c_int
    lib.ld_kernel_matrix.argtypes = [C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    lib.ld_gaussian_draw.restype = C.c_int64
    lib.ld_gaussian_draw.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ld_set_threads.argtypes = [C.c_int]
    lib.ld_set_threads(min(32, os.cpu_count() or 1))
    _lib = lib
  return _lib


def _f64(a):
  return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
  return None if a is None else a.ctypes.data_as(C.c_void_p)


def gp_truth(kind, dim_bandwidths, scale, X, y_centred, diag_add, X_test=None, mean_const=0.0, best=0.0, nu=0.0,
             want_L=False, want_K=False, ts_normals=None, ts_jitter=0.0, want_cov=False):
  """ kind 'se' | 'matern'.  diag_add = noise_var (+ the jitter stable_cholesky added, if any).
      Returns a dict: alpha, lml, and (with X_test) mu, sd, ei; optionally L, K.  With ts_normals [m]:
      'draw' = mu + chol(Sigma + ts_jitter I) u, the joint Thompson draw over X_test ('cov' with
      want_cov: Sigma rounded to double). """
  lib = _load()
  X, yc, bw = _f64(X), _f64(y_centred).reshape(-1), _f64(dim_bandwidths).reshape(-1)
  n, d = X.shape
  Xs = None if X_test is None else _f64(X_test)
  m = 0 if Xs is None else Xs.shape[0]
  out = dict(alpha=np.empty(n), lml=np.empty(1))
  if m:
    out.update(mu=np.empty(m), sd=np.empty(m), ei=np.empty(m))
  L = np.empty((n, n)) if want_L else None
  K = np.empty((n, n)) if want_K else None
  u = None if ts_normals is None else _f64(ts_normals).reshape(-1)
  draw = np.empty(m) if u is not None else None
  cov = np.empty((m, m)) if want_cov else None
  rc = lib.ld_gp_truth(0 if kind == 'se' else 1, float(nu), float(scale), _p(bw), _p(X), _p(yc), n, d,
                       float(diag_add), _p(Xs), m, float(mean_const), float(best), _p(out['alpha']),
                       _p(out['lml']), _p(out.get('mu')), _p(out.get('sd')), _p(out.get('ei')), _p(L), _p(K),
                       _p(u), float(ts_jitter), _p(draw), _p(cov))
  if rc < -1000000:
    raise np.linalg.LinAlgError('truth: posterior covariance not positive definite at pivot %d' % (-rc - 1000000))
  if draw is not None:
    out['draw'] = draw
  if cov is not None:
    out['cov'] = cov
  if rc > 0:
    raise np.linalg.LinAlgError('truth: matrix not positive definite at pivot %d' % rc)
  if rc < 0:
    raise MemoryError('truth: allocation failed')
  out['lml'] = float(out['lml'][0])
  if want_L:
    out['L'] = L
  if want_K:
    out['K'] = K
  return out


def kernel_matrix(kind, dim_bandwidths, scale, X1, X2, nu=0.0):
  lib = _load()
  X1, X2, bw = _f64(X1), _f64(X2), _f64(dim_bandwidths).reshape(-1)
  K = np.empty((X1.shape[0], X2.shape[0]))
  lib.ld_kernel_matrix(0 if kind == 'se' else 1, float(nu), float(scale), _p(bw), _p(X1), X1.shape[0], _p(X2),
                       X2.shape[0], X1.shape[1], _p(K))
  return K


def gaussian_draw(mu, C_mat, u, jitter=0.0):
  """ mu + chol(C + jitter I) u in extended precision (one joint Thompson block, given its inputs). """
  lib = _load()
  Cm, mu, u = _f64(C_mat), _f64(mu).reshape(-1), _f64(u).reshape(-1)
  s = np.empty(len(mu))
  rc = lib.ld_gaussian_draw(_p(Cm), len(mu), float(jitter), _p(mu), _p(u), _p(s))
  if rc > 0:
    raise np.linalg.LinAlgError('truth: covariance not positive definite at pivot %d' % rc)
  return s


def gram_truth(K, diag_add, y_centred, K_cross=None, k_ss=None, K_tete=None, mean_const=0.0):
  """ The posterior's linear algebra in extended precision for ANY kernel, given its Gram matrices in
      double: K [n x n] (no noise), K_cross [m x n], k_ss [m] = k(x*, x*), K_tete [m x m] (for 'cov').
      Returns a dict: alpha, lml, and with K_cross mu (+ sd with k_ss, cov with K_tete). """
  lib = _load()
  if not hasattr(lib, '_gram_ready'):
    lib.ld_gram_truth.restype = C.c_int64
    lib.ld_gram_truth.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int64, C.c_double] + [C.c_void_p] * 5
    lib._gram_ready = True
  K, yc = _f64(K), _f64(y_centred).reshape(-1)
  n = K.shape[0]
  Kx = None if K_cross is None else _f64(K_cross)
  m = 0 if Kx is None else Kx.shape[0]
  kss = None if k_ss is None else _f64(k_ss).reshape(-1)
  Kss = None if K_tete is None else _f64(K_tete)
  out = dict(alpha=np.empty(n), lml=np.empty(1))
  if m:
    out['mu'] = np.empty(m)
    if kss is not None:
      out['sd'] = np.empty(m)
    if Kss is not None:
      out['cov'] = np.empty((m, m))
  rc = lib.ld_gram_truth(_p(K), n, float(diag_add), _p(yc), _p(Kx), _p(kss), _p(Kss), m, float(mean_const),
                         _p(out['alpha']), _p(out['lml']), _p(out.get('mu')), _p(out.get('sd')), _p(out.get('cov')))
  if rc > 0:
    raise np.linalg.LinAlgError('truth: matrix not positive definite at pivot %d' % rc)
  if rc < 0:
    raise MemoryError('truth: allocation failed')
  out['lml'] = float(out['lml'][0])
  return out
