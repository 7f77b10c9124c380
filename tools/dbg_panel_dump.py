"""Block 0 of the one-launch panel's factor against LAPACK, entry by entry (diagnostics build; see tools/dbg_panel.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dragonfly_amd.engine import Engine
eng = Engine()
lib = eng.lib
lib.dfh_debug_panel_stamps.restype = C.c_int
lib.dfh_debug_panel_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
reps = 1
ms = (C.c_double * reps)(); st = (C.c_longlong * (8 * 64))()
rc = lib.dfh_debug_panel_stamps(eng.ctx, reps, 0, ms, st)
lib.dfh_debug_panel_data.restype = C.c_int
lib.dfh_debug_panel_data.argtypes = [C.c_void_p, C.c_void_p]
A = np.empty((512, 512)); Lf = np.empty((8, 64, 64))
lib.dfh_debug_panel_data(A.ctypes.data_as(C.c_void_p), Lf.ctypes.data_as(C.c_void_p))
Lref = np.linalg.cholesky(A)[:64, :64]
B = Lf[0]
np.set_printoptions(linewidth=250, precision=4)
bad = ~(np.abs(np.tril(B) - Lref) < 1e-10)
print('rc', rc)
print('wrong entries per column (rows >= col):', [int(bad[c:, c].sum()) for c in range(64)])
for c in range(3, 8):
  print('col', c, 'got ', B[c:c + 6, c])
  print('      want', Lref[c:c + 6, c])
print('NaN mask rows 0..20 x cols 0..20:')
print(np.isnan(B[:21, :21]).astype(int))
