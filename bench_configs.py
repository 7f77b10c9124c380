"""Synthetic inputs of BASELINE.json's configs: shared by bench.py and the
full-size parity tests.  Deterministic (seeded), unit-cube inputs, O(1) targets, float64."""
import numpy as np

N_TRAIN, DIM, TS_BLOCK = 16384, 32, 4096      # configs 3 / 4
CANDS_PER_GPU = 262144                        # config 4: 2 097 152 candidates over 8 GPUs
CANDS_TOTAL_8 = 8 * CANDS_PER_GPU


def hartmann6(X):
  """ Hartmann-6 on the unit cube (the reference defines the same function in
      exd/../euclidean_synthetic_functions.py:16-49); produces the targets of config 2. """
  A = np.array([[10, 3, 17, 3.5, 1.7, 8], [0.05, 10, 17, 0.1, 8, 14],
                [3, 3.5, 1.7, 10, 17, 8], [17, 8, 0.05, 10, 0.1, 14]], dtype=float)
  P = 1e-4 * np.array([[1312, 1696, 5569, 124, 8283, 5886], [2329, 4135, 8307, 3736, 1004, 9991],
                       [2348, 1451, 3522, 2883, 3047, 6650], [4047, 8828, 8732, 5743, 1091, 381]], dtype=float)
  alpha = np.array([1.0, 1.2, 3.0, 3.2])
  inner = (A[None, :, :] * (X[:, None, :] - P[None, :, :]) ** 2).sum(axis=2)
  return (alpha[None, :] * np.exp(-inner)).sum(axis=1)


def park1(X4):
  """ Park function 1 on [0,1]^4 (targets of the additive config 5). """
  x1, x2, x3, x4 = [np.maximum(X4[:, i], 1e-6) for i in range(4)]
  return -((x1 / 2) * (np.sqrt(1 + (x2 + x3 ** 2) * x4 / x1 ** 2) - 1) + (x1 + 3 * x4) * np.exp(1 + np.sin(x3)))


def config2():
  """ Hartmann6, n=4096, d=6, Matern-2.5, EI over 65536 candidates. """
  n, d, m = 4096, 6, 65536
  X = np.random.RandomState(102).random_sample((n, d))
  Y = hartmann6(X)
  return dict(n=n, d=d, m=m, X=X, Y=Y, kind='matern', nu=2.5, scale=float(Y.var()), bw=0.5 * np.ones(d),
              mean_c=float(np.median(Y)), noise=float(Y.var() / 20),
              cands=np.random.RandomState(202).random_sample((m, d)), best=float(Y.max()))


def config3(n=N_TRAIN):
  """ Synthetic d=32, n=16384, SE-ARD: X ~ U[0,1)^32 (seed 103), Y = sum_j j/d x_j^2 + 0.01 N(0,1),
      bandwidths 0.2 sqrt(32) (0.5 + j/32), mean = median(Y), noise = Var(Y)/20. """
  rs = np.random.RandomState(103)
  X = rs.random_sample((N_TRAIN, DIM))
  w = (np.arange(DIM) + 1.0) / DIM
  Y = (X ** 2).dot(w) + 0.01 * rs.randn(N_TRAIN)
  bw = 0.2 * np.sqrt(DIM) * (0.5 + np.arange(DIM) / 32.0)
  return dict(n=n, d=DIM, X=X[:n], Y=Y[:n], kind='se', scale=float(Y.var()), bw=bw,
              mean_c=float(np.median(Y)), noise=float(Y.var() / 20))


def config3_candidates(m=65536):
  return np.random.RandomState(203).random_sample((m, DIM))


def config4_rows(lo, hi):
  """ Rows [lo, hi) of the ONE seed-204 candidate set (2 097 152 x 32) and of the seed-304 standard
      normals -- what a single process drawing the whole set would hold in those rows. """
  cands = np.random.RandomState(204).random_sample((hi, DIM))[lo:]
  U = np.random.RandomState(304).standard_normal(hi)[lo:]
  return np.ascontiguousarray(cands), np.ascontiguousarray(U)


def config4_shard(rank, world=None):
  """ Rank's rows [rank*262144, (rank+1)*262144) of the ONE seed-204 candidate set
      (2 097 152 x 32) and of the seed-304 standard normals: the N-GPU run evaluates the first
      N*262144 rows of the same stream a single process would draw. """
  del world
  lo, hi = rank * CANDS_PER_GPU, (rank + 1) * CANDS_PER_GPU
  cands = np.random.RandomState(204).random_sample((hi, DIM))[lo:]
  U = np.random.RandomState(304).standard_normal(hi)[lo:]
  return np.ascontiguousarray(cands), np.ascontiguousarray(U)


def config5():
  """ Additive GP, d=100, 20 groups of 5 ('park1-100' tiling), n=4096, add-UCB over 20 x 3276. """
  n, d, G = 4096, 100, 20
  X = np.random.RandomState(105).random_sample((n, d))
  Y = sum(park1(X[:, 4 * i:4 * i + 4]) for i in range(25))
  perm = list(np.random.RandomState(405).permutation(d))
  groups = [perm[i:i + 5] for i in range(0, d, 5)]                   # euclidean_gp.py:733-735
  bws = [0.2 * np.sqrt(5) * np.ones(5) for _ in groups]
  m_j = 65536 // G
  cands = [np.random.RandomState(205 + j).random_sample((m_j, 5)) for j in range(G)]
  return dict(n=n, d=d, G=G, X=X, Y=Y, groups=groups, bws=bws, scale=float(Y.var()), noise=float(Y.var() / 20),
              mean_c=float(np.median(Y)), m_j=m_j, cands=cands)
