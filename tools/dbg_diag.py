# needs a library built with the diagnostics hooks: python -m dragonfly_amd.build --force --debug-hooks (include/dfhip_debug.h)
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dragonfly_amd.engine import Engine
eng = Engine()
lib = eng.lib
lib.dfh_debug_diag_step.restype = C.c_int
lib.dfh_debug_diag_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
for rows in (0, 64, 448):
  for reps in (1, 10, 200):
    ms = C.c_double(0); cyc = (C.c_longlong * 7)()
    rc = lib.dfh_debug_diag_step(eng.ctx, reps, rows, C.byref(ms), cyc)
    print('rows_below=%d reps=%d rc=%d  %.1f us/launch  cycles load=%d factor=%d trsm=%d | wave3: own block %d..%d scaled %d inverse %d' % (rows, reps, rc, ms.value * 1e3, cyc[0], cyc[1], cyc[2], cyc[3], cyc[4], cyc[5], cyc[6]))
