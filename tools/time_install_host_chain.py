"""Host Python of ONE density evaluation of the slice sampler under install() (dragonfly_amd/install.py: _post_logp_batch ->
_lml_batch -> engine.gp_lml_batch), with the library call replaced by a no-op (tools/noop_lml_batch.c, built here with
gcc): what the mirrors and the binding cost per call, without a GPU.  Needs the Dragonfly checkout."""
import os, subprocess, sys, tempfile, time, warnings, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import numpy as np
from oracle.make_golden import import_reference
import_reference()
from dragonfly_amd import install, euclidean_gp, general_utils, gp_core, gpb_acquisitions, kernel, _lib
from dragonfly_amd import engine as E
_so = os.path.join(tempfile.mkdtemp(), 'libnoop.so')
subprocess.run(['gcc', '-O2', '-shared', '-fPIC', '-o', _so, os.path.join(ROOT, 'tools', 'noop_lml_batch.c')], check=True)
noop = C.CDLL(_so)
res, args = _lib.SIGNATURES['dfh_gp_lml_batch']
noop.dfh_gp_lml_batch.restype = res; noop.dfh_gp_lml_batch.argtypes = args
class FakeLib(object):
  dfh_gp_lml_batch = noop.dfh_gp_lml_batch
eng = E.Engine.__new__(E.Engine); eng.lib = FakeLib(); eng.ctx = C.c_void_p(1)
class DA(E.DeviceArray):
  def __init__(self, shape): self.shape = shape; self.ptr = C.c_void_p(4096)
  def __del__(self): pass
eng.to_device = lambda h: DA(np.shape(h))
for m in (E, euclidean_gp, general_utils, gp_core, kernel):
  m.get_engine = (lambda _e=eng: _e)
install.install()
import dragonfly.opt.gp_bandit as GB
from dragonfly.gp.euclidean_gp import euclidean_gp_args
from dragonfly.utils.option_handler import load_options
rs = np.random.RandomState(0)
X = [rs.rand(6) for _ in range(40)]; Y = [float(np.sin(x).sum()) for x in X]
opts = load_options(euclidean_gp_args, partial_options={'hp_tune_criterion': 'post_sampling', 'kernel_type': 'matern'})
with warnings.catch_warnings():
  warnings.simplefilter('ignore')
  f = GB.EuclideanGPFitter(X, Y, options=opts); f._set_up()
lo = np.array([b[0] for b in f.cts_hp_bounds]); hi = np.array([b[1] for b in f.cts_hp_bounds])
f.hps = list(lo + (hi - lo) * rs.rand(len(lo))); f.curr_hp = 3; f.other_gp_params = None
def T(fn, reps=20000):
  fn(); t0=time.perf_counter()
  for _ in range(reps): fn()
  return (time.perf_counter()-t0)/reps*1e6
xs3 = list(lo[3] + (hi[3]-lo[3]) * rs.rand(3)); xs1 = xs3[:1]
print('_post_logp_batch, 3 values: %.1f us; 1 value: %.1f us' % (T(lambda: f._post_logp_batch(xs3)), T(lambda: f._post_logp_batch(xs1))))
cands = [np.array(f.hps, dtype=float)[:len(lo)] for _ in range(3)]
print('_lml_batch, 3 candidates  : %.1f us' % T(lambda: f._lml_batch(cands, [[]]*3, None)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20000): f._post_logp_batch(xs3)
pr.disable(); pstats.Stats(pr).sort_stats('tottime').print_stats(14)
