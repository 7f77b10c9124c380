"""Where does a block column of the team form of the tuning objective (lml_team_kernel) go?  Diagnostics build:
    python -m dragonfly_amd.build --debug-hooks;  DFH_LIB=dragonfly_amd/libdfhip_dbg.so python tools/dbg_lmlt.py n nb
Stamps are s_memrealtime (100 MHz), thread 0 of every workgroup, per block column:
  0 column entered | 1 B-row flag seen | 2 first products done | 3 look-ahead products done | 4 diag flag seen |
  5 image fetched | 6 first tile solved + stored | 7 look-ahead product + row flag out | 8 next diagonal tile factored + announced | 9 column left"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import get_engine, KernelSpec
eng = get_engine()
lib = eng.lib
lib.dfh_debug_lmlt_stamps.restype = C.c_int
lib.dfh_debug_lmlt_stamps.argtypes = [C.c_void_p]
n, nb = int(sys.argv[1]), int(sys.argv[2])
rs = np.random.RandomState(n)
d = 6
X = rs.rand(n, d); Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
Xd = eng.to_device(X)
specs = [KernelSpec('se', d, float(Y.var()) * (0.5 + rs.rand()), 0.2 + 0.6 * rs.rand(d)) for _ in range(nb)]
means = [0.0] * nb; noises = [float(Y.var() * 0.05)] * nb
eng.gp_lml_batch(specs, Xd, Y, means, noises)
nbt = (n + 1 + 63) // 64
T = 1
cap = int(os.environ.get('DFH_LML_TEAM', '8'))
while T * 2 <= cap and T * 2 * nb <= 256 and T * 2 <= nbt:
  T *= 2
nwg = nb * T
buf = eng.to_device(np.zeros(nwg * 32 * 16))
lib.dfh_debug_lmlt_stamps(buf.ptr)
eng.gp_lml_batch(specs, Xd, Y, means, noises)
lib.dfh_debug_lmlt_stamps(None)
st = buf.download().view(np.int64).reshape(nwg, 32, 16) * 0.01       # us
print('n', n, 'nb', nb, 'T', T, 'nbt', nbt)
c = 0                                                                # candidate 0's team
t0 = min(st[c * T + t, 0, 0] for t in range(T))
names = ['enter', 'brow', 'prod', 'la-prod', 'diag', 'image', 'solved', 'la+flag', 'factor', 'leave']
print('col owner | per member: ' + ' '.join(names))
for j in range(nbt):
  print('--- column %d (owner %d, next owner %d)' % (j, j % T, (j + 1) % T))
  for t in range(T):
    row = st[c * T + t, j]
    if row[0] == 0:
      continue
    print('  m%d: ' % t + ' '.join(('%7.1f' % (row[e] - t0)) if row[e] > 0 else '      -' for e in range(10)))
# the chain: time at which diag[j+1] went out (event 8 of the next owner in column j), differences
pub = []
for j in range(nbt - 1):
  t = (j + 1) % T
  pub.append(st[c * T + t, j, 8] - t0)
print('factor step of tile j (its owner): stage | barrier | factor64 (wave 0) | barrier (all waves) | image stores issued | announced')
for j in range(nbt):
  row = st[c * T + (j % T), j] if j == 0 else st[c * T + (j % T), j - 1]
  row2 = st[c * T + (j % T), j]
  use = row2 if row2[10] > 0 else None
  if use is not None:
    print('  tile %2d: ' % j + ' '.join('%6.1f' % (use[e + 1] - use[e]) for e in range(10, 15)) + '   total %.1f' % (use[15] - use[10]))
print('diag[j+1] announced at:', ' '.join('%.1f' % v for v in pub))
print('hops:', ' '.join('%.1f' % (b - a) for a, b in zip(pub[:-1], pub[1:])))
end = max(st[c * T + t, j, 9] for t in range(T) for j in range(nbt)) - t0
print('team done at %.1f us' % end)
