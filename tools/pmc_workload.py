"""Small fixed workload for rocprofv3 --pmc passes: the kernel-matrix build at n=16384 (HBM-bound
pass) and the n=15872, k=512 trailing SYRK + an 8192^3 GEMM (MFMA_F64 passes)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dragonfly_amd.engine import Engine, KernelSpec   # noqa: E402

eng = Engine()
rs = np.random.RandomState(0)
n, d = 16384, 32
X = eng.to_device(rs.rand(n, d))
Kd = eng.empty((n, n))
spec = KernelSpec('se', d, 1.3, 0.2 * np.sqrt(d) * (0.5 + np.arange(d) / 32.0))
for _ in range(3):
  eng.kernel_matrix(spec, X, None, out=Kd)
# Matern-2.5 Gram matrix, and the cross matrices K(X*, X) of a posterior chunk (strip kernel):
# SE 32768 x 16384 at d = 32 and Matern-2.5 65536 x 4096 at d = 6 (BASELINE config 2's shape)
spec_m = KernelSpec('matern', d, 1.3, 0.5 * np.ones(d), nu=2.5)
for _ in range(2):
  eng.kernel_matrix(spec_m, X, None, out=Kd)
Xs = eng.to_device(rs.rand(32768, d))
Kc = eng.empty((32768, n))
for _ in range(3):
  eng.kernel_matrix(spec, Xs, X, out=Kc)
Kc.free(); Xs.free()
X6, Xs6 = eng.to_device(rs.rand(4096, 6)), eng.to_device(rs.rand(65536, 6))
Kc6 = eng.empty((65536, 4096))
spec6 = KernelSpec('matern', 6, 1.3, 0.5 * np.ones(6), nu=2.5)
for _ in range(3):
  eng.kernel_matrix(spec6, Xs6, X6, out=Kc6)
Kc6.free(); X6.free(); Xs6.free()
M, K = 15872, 512
A = eng.to_device(rs.rand(M, K) - 0.5)
Cd = eng.empty((M, M))
for _ in range(3):
  eng.gemm(A, A, alpha=-1.0, beta=1.0, shape=(M, M, K), out=Cd, lower_only=True)
M = 8192
A2 = eng.to_device(rs.rand(M, M) - 0.5)
B2 = eng.to_device(rs.rand(M, M) - 0.5)
C2 = eng.empty((M, M))
for _ in range(2):
  eng.gemm(A2, B2, shape=(M, M, M), out=C2)
eng.sync()
print('pmc workload done')
# one n=4096 factorisation: the pivot-step kernel (diag_step64_kernel) for the LDS counters
Kc = eng.empty((4096, 4096))
eng.kernel_matrix(spec, eng.to_device(rs.rand(4096, d)), None, diag_add=0.05, out=Kc)
eng.cholesky(Kc)
# FETCH_SIZE calibration in the GEMM's own load pattern (16 B per lane): A [32768 x 16384] is read
# exactly once from HBM (4.295 GB, well past the 256 MB Infinity Cache), B [128 x 16384] (16.8 MB)
# stays cache-resident, C is 33.5 MB.
M, N, K = 32768, 128, 16384
A3 = eng.empty((M, K))
B3 = eng.to_device(rs.rand(N, K) - 0.5)
C3 = eng.empty((M, N))
for _ in range(2):
  eng.gemm(A3, B3, shape=(M, N, K), out=C3)
eng.sync()
print('pmc calibration gemm done')
# round 3: the symmetric Gram matrix of an additive kernel (config 5's shape: n = 4096, d = 100, 20 groups of
# 5 -> kernmat_symmulti_kernel) and a posterior over 65536 candidates at n = 4096, d = 6 Matern-2.5 (config 2:
# the cross-matrix strip kernel with the mean fused in, kernmat_strip_kernel<..., true>, + k_mu_finish)
n5, d5 = 4096, 100
groups = [list(range(i, i + 5)) for i in range(0, d5, 5)]
spec5 = KernelSpec('additive', d5, 1.7, groups=groups, sub_kinds=['se'] * 20, sub_scales=[1.0] * 20, sub_nus=[0.0] * 20,
                   sub_bandwidths=[np.full(5, 0.2 * np.sqrt(5.0))] * 20)
X5 = eng.to_device(rs.rand(n5, d5))
K5 = eng.empty((n5, n5))
for _ in range(3):
  eng.kernel_matrix(spec5, X5, None, diag_add=0.1, out=K5)
X6h = rs.rand(4096, 6)
Y6 = np.sin(3 * X6h.sum(axis=1)) + 0.05 * rs.randn(4096)
gp6 = eng.gp_fit(spec6, X6h, Y6 - np.median(Y6), float(Y6.var() / 20))
Xc6 = eng.to_device(rs.rand(65536, 6))
for _ in range(2):
  gp6.acq_argmax('ei', Xc6, params=(float(Y6.max()), 0.0), mean_const=float(np.median(Y6)))
eng.sync()
print('pmc round-3 kernels done')
