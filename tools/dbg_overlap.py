# needs a library built with the diagnostics hooks: python -m dragonfly_amd.build --force --debug-hooks (include/dfhip_debug.h)
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dragonfly_amd.engine import Engine
eng = Engine()
lib = eng.lib
lib.dfh_debug_overlap.restype = C.c_int
lib.dfh_debug_overlap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
for which in (0, 1, 2):
  out = (C.c_double * 4)()
  rc = lib.dfh_debug_overlap(eng.ctx, which, 20, 500, out)
  tot = int(out[3] // 1e6) ; 
  print('which=%d rc=%d big alone %.1f ms, small alone %.1f ms, small-stream done at %.1f ms when together, packed=%r' % (which, rc, out[0], out[1], out[2], out[3]))
