"""Where does a hop of the one-launch panel's diagonal chain go?  Needs the diagnostics build:
    python -m dragonfly_amd.build --debug-hooks;  DFH_LIB=dragonfly_amd/libdfhip_dbg.so python tools/dbg_panel.py [rows_below]
Stamps are s_memrealtime (100 MHz: 10 ns steps), thread 0 of each strip's workgroup."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import Engine
eng = Engine()
lib = eng.lib
lib.dfh_debug_panel_stamps.restype = C.c_int
lib.dfh_debug_panel_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = 20
nwg = 8 + (rows + 63) // 64
ms = (C.c_double * reps)(); st = (C.c_longlong * (nwg * 64))()
rc = lib.dfh_debug_panel_stamps(eng.ctx, reps, rows, ms, st)
# the published diagonal factor blocks against LAPACK
lib.dfh_debug_panel_data.restype = C.c_int
lib.dfh_debug_panel_data.argtypes = [C.c_void_p, C.c_void_p]
A = np.empty((512 + rows, 512)); Lf = np.empty((8, 64, 64))
lib.dfh_debug_panel_data(A.ctypes.data_as(C.c_void_p), Lf.ctypes.data_as(C.c_void_p))
Lref = np.linalg.cholesky(A[:512])
for s in range(8):
  blk = Lref[64 * s:64 * s + 64, 64 * s:64 * s + 64]
  err = np.abs(np.tril(Lf[s]) - blk)
  i, j = np.unravel_index(np.argmax(err), err.shape)
  print('factor block %d: max |L - LAPACK| = %.2e at (%d, %d)%s' % (s, err.max(), i, j, '' if err.max() < 1e-12 else '   <-- WRONG; first bad column %d'
        % int(np.argmax((err > 1e-12).any(axis=0)))))
t = np.array(list(st), dtype=np.int64).reshape(nwg, 64) * 0.01      # microseconds
if rc != 0:
  from dragonfly_amd import _lib
  print('ERROR:', _lib.last_error())
print('rc', rc, 'rows_below', rows, 'launch us:', ' '.join('%.1f' % (v * 1e3) for v in ms))
t0 = t[0, 0]
print('strip: start | per J: wait load solve update | staged factored(w0) imaged published   (us since strip 0 started)')
for s in range(8):
  line = 'strip %d: %6.2f |' % (s, t[s, 0] - t0)
  for J in range(s):
    line += ' J%d[%6.2f %6.2f %6.2f %6.2f]' % (J, *(t[s, 1 + 4 * J + i] - t0 for i in range(4)))
  line += ' | %6.2f %6.2f %6.2f %6.2f' % tuple(t[s, 40 + i] - t0 for i in range(4))
  print(line)
print('hop = published(s) - published(s-1), split along the critical step J = s-1:')
print('        flag->seen  load   solve  update  stage  factor  image  publish |  hop')
for s in range(1, 8):
  J = s - 1
  pub_prev = t[s - 1, 43]
  a = [t[s, 1 + 4 * J] - pub_prev, t[s, 2 + 4 * J] - t[s, 1 + 4 * J], t[s, 3 + 4 * J] - t[s, 2 + 4 * J],
       t[s, 4 + 4 * J] - t[s, 3 + 4 * J], t[s, 40] - t[s, 4 + 4 * J], t[s, 42] - t[s, 40], 0.0, t[s, 43] - t[s, 42]]
  print('strip %d: ' % s + ' '.join('%6.2f' % v for v in a) + ' | %6.2f' % (t[s, 43] - pub_prev))
print('last step in detail (thread 0 = wave 0): after-load -> stages 0-2 done | barrier | early product | flag2 seen | stage 3 + stores | own barrier | last product | end barrier')
for s in range(1, 8):
  J = s - 1
  pts = [t[s, 2 + 4 * J], t[s, 48], t[s, 49], t[s, 50], t[s, 51], t[s, 52], t[s, 53], t[s, 54], t[s, 4 + 4 * J]]
  print('strip %d: ' % s + ' '.join('%6.2f' % (b - a) for a, b in zip(pts[:-1], pts[1:])))
if rows > 0:
  for g in range(8, nwg):
    print('below %d: start %.2f' % (g, t[g, 0] - t0), ' '.join('J%d[%.2f..%.2f]' % (J, t[g, 1 + 4 * J] - t0, t[g, 4 + 4 * J] - t0) for J in range(8)))
