"""TRSM-shaped GEMM (32768 x 512 x 8192) and an 8192^3 GEMM: for rocprofv3 --pmc FETCH_SIZE comparisons."""
import os, sys
root = os.environ.get('DFH_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
from dragonfly_amd.engine import get_engine
from dragonfly_amd import _lib
print('library', _lib.LIB_PATH)
eng = get_engine()
rs = np.random.RandomState(0)
M, N, K = 32768, 512, 8192
A = eng.to_device(rs.rand(M, K) - 0.5); B = eng.to_device(rs.rand(N, K) - 0.5); Cd = eng.empty((M, N))
for _ in range(3):
  eng.gemm(A, B, shape=(M, N, K), out=Cd)
eng.sync()
