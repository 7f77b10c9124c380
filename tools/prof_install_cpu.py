"""Where the Python of dragonfly_amd.install() goes in a real run -- on CPU, with the NumPy stand-in engine of the tests
behind the mirrors (tests/oracle_engine.py), so that what is left after taking the stand-in's own time out is the host
code the GPU run pays too:   python tools/prof_install_cpu.py [evals] [profile_out]
(needs the Dragonfly checkout: DRAGONFLY_REFERENCE or /root/reference)"""
import cProfile
import os
import pstats
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle.make_golden import import_reference

evals = int(sys.argv[1]) if len(sys.argv) > 1 else 60
out = sys.argv[2] if len(sys.argv) > 2 else None
import_reference()
from dragonfly import maximise_function
import dragonfly.utils.euclidean_synthetic_functions as S
from oracle_engine import OracleEngine
from dragonfly_amd import install, euclidean_gp, general_utils, gp_core, gpb_acquisitions, kernel
from dragonfly_amd import engine as engine_mod

_sf = S.get_syn_func_caller('hartmann6', noise_type='no_noise')
objective = lambda x: float(_sf.func(np.asarray(x, dtype=float)))
bounds = [[float(a), float(b)] for a, b in _sf.domain.bounds]
eng = OracleEngine()
for m in (engine_mod, euclidean_gp, general_utils, gp_core, kernel):
  m.get_engine = (lambda _e=eng: _e)
gpb_acquisitions.DEVICE_CANDIDATES = False
install.install()
np.random.seed(11)
prof = cProfile.Profile()
t0 = time.perf_counter()
prof.enable()
with warnings.catch_warnings():
  warnings.simplefilter('ignore')
  val, pt, history = maximise_function(objective, bounds, evals)
prof.disable()
wall = time.perf_counter() - t0
st = pstats.Stats(prof)
standin = sum(v[3] for k, v in st.stats.items() if k[0].endswith('oracle_engine.py') and k[2] in
              ('gp_lml_batch', 'gp_fit', 'predict', 'acq_argmax', 'thompson', 'kernel_matrix', 'stable_cholesky', 'append', 'add_ucb_all', 'add_ucb_group'))
print('wall %.1f s under cProfile; cumulative time inside the stand-in engine\'s entry points %.1f s' % (wall, standin))
fh = open(out, 'w') if out else sys.stdout
pstats.Stats(prof, stream=fh).sort_stats('tottime').print_stats(40)
pstats.Stats(prof, stream=fh).sort_stats('cumulative').print_stats(60)
