"""Device-backed versions of the linear-algebra helpers of dragonfly/utils/general_utils.py that
sit on the GP hot path: dist_squared (:58-70), stable_cholesky (:166-204),
solve_lower/upper_triangular (:208-221), draw_gaussian_samples (:224-232), plus the two
one-line domain maps (:20-27).  Same names, arguments and error behaviour as the reference.
"""
import numpy as np

from .engine import get_engine


def map_to_cube(pts, bounds):
  """ general_utils.py:20-22 """
  return (pts - bounds[:, 0])/(bounds[:, 1] - bounds[:, 0])


def map_to_bounds(pts, bounds):
  """ general_utils.py:25-27 """
  return pts * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0]


def dist_squared(X1, X2):
  """ n1 x n2 matrix of squared distances, clipped at zero (general_utils.py:58-70). """
  A = np.asarray(X1, dtype=np.float64)
  B = np.asarray(X2, dtype=np.float64)
  if A.shape[1] != B.shape[1]:
    raise ValueError('Second dimension of X1 and X2 should be equal.')
  return get_engine().dist_squared(A, B)


def stable_cholesky(M, add_to_diag_till_psd=True):
  """ L with L L' = M; if M is numerically not psd, 10^p * max(diag(M)) is added to the diagonal
      for p = -11..4 (general_utils.py:166-204).  Raises np.linalg.LinAlgError when
      add_to_diag_till_psd is False and the plain factorisation fails, ValueError when the
      ladder is exhausted. """
  M = np.asarray(M, dtype=np.float64)
  if M.size == 0:
    return M
  eng = get_engine()
  if not add_to_diag_till_psd:
    return eng.cholesky(M)
  L, power = eng.stable_cholesky(M, return_power=True)
  if power is not None and power >= -7:
    # the reference warns once, at the first failed attempt with power > -9 (i.e. -8)
    from warnings import warn
    warn(('Could not compute Cholesky decomposition despite adding %0.4f to the '
          'diagonal. This is likely because the M is not positive semi-definite.')%(
              (10**-8) * np.diag(M).max()))
  return L


def _solve_triangular_common(A, b, lower):
  """ general_utils.py:208-213 """
  A = np.asarray(A, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  if b.shape[0] == 0 and A.size == 0:
    return np.zeros(b.shape)          # nothing to solve
  eng = get_engine()
  if lower:
    return eng.solve_triangular(A, b, upper=False)
  # A is upper triangular: A = L^T with L = A^T lower triangular
  return eng.solve_triangular(np.ascontiguousarray(A.T), b, upper=True)


def solve_lower_triangular(A, b):
  """ Solves Ax=b when A is lower triangular (general_utils.py:215-217). """
  return _solve_triangular_common(A, b, lower=True)


def solve_upper_triangular(A, b):
  """ Solves Ax=b when A is upper triangular (general_utils.py:219-221). """
  return _solve_triangular_common(A, b, lower=False)


def project_symmetric_to_psd_cone(M, is_symmetric=True, epsilon=0):
  """ V max(Lambda, epsilon) V^T of the symmetric M (general_utils.py:150-163), on the device:
      no eigen-decomposition, the matrix sign function by Newton-Schulz steps on the fp64 MFMA GEMM
      (csrc/psdproj.hip).  The reference's non-symmetric branch (np.linalg.eig) has no caller on
      the GP path and is not provided. """
  if not is_symmetric:
    raise NotImplementedError('project_symmetric_to_psd_cone: symmetric matrices only.')
  M = np.asarray(M, dtype=np.float64)
  if M.size == 0:
    return M
  return get_engine().project_psd(M, epsilon=float(epsilon))


def draw_gaussian_samples(num_samples, mu, K):
  """ num_samples draws from N(mu, K) (general_utils.py:224-232); the normals come from the
      global np.random state with the reference's call, the factor and the product from the
      device. """
  factor = stable_cholesky(K)
  normals = np.random.normal(size=(len(mu), num_samples))
  return get_engine().gemm(factor, normals, transb=True).T + mu
