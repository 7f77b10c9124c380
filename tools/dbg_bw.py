# needs a library built with the diagnostics hooks: python -m dragonfly_amd.build --force --debug-hooks (include/dfhip_debug.h)
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dragonfly_amd.engine import Engine
eng = Engine()
lib = eng.lib
lib.dfh_debug_write_bw.restype = C.c_int
lib.dfh_debug_write_bw.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double)]
for gb in (0.5, 2.147, 4.0):
  out = (C.c_double * 3)()
  rc = lib.dfh_debug_write_bw(eng.ctx, gb, out)
  print('%.3f GB: rc=%d fill(16B stores) %.2f TB/s, hipMemsetAsync %.2f TB/s, copy (read+write) %.2f TB/s' % (gb, rc, out[0], out[1], out[2]))
