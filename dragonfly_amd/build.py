"""Build libdfhip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m dragonfly_amd.build [--force] [--debug-hooks]

--debug-hooks builds a SECOND library, dragonfly_amd/libdfhip_dbg.so (objects in csrc/_obj_dbg), that
also holds the diagnostics entry points of include/dfhip_debug.h (tools/dbg_*.py; selected at run time
with DFH_LIB=/path/to/libdfhip_dbg.so); libdfhip.so exports exactly the C-ABI of include/dfhip.h.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container;
the resulting dragonfly_amd/libdfhip.so travels with the repository snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ_DIR = os.path.join(HERE, 'csrc', '_obj')
LIB_PATH = os.path.join(HERE, 'libdfhip.so')
SOURCES = ['runtime.hip', 'gemm_f64.hip', 'kernmat.hip', 'chol.hip', 'api.hip', 'rng.hip', 'mgpu.hip', 'psdproj.hip', 'mtjump.hip']
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "factor64.h"), os.path.join(CSRC, "kerneval.h"),
           os.path.join(HERE, '..', 'include', 'dfhip.h')]
# -ffp-contract=off: no implicit a*b+c fusion, so the elementwise epilogues round exactly where the
# NumPy expressions they replace round; fused multiply-adds are written explicitly (fma) where wanted.
CXXFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
            '-Wall', '-Wno-unused-function', '-Wno-unused-result']


def _hipcc():
  for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
    if cand and os.path.exists(cand):
      return cand
  raise RuntimeError('hipcc not found (set HIPCC=/path/to/hipcc)')


def _stale(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, debug_hooks=False):
  """Compile every HIP translation unit for gfx950 and link libdfhip.so. Returns its path."""
  hipcc = _hipcc()
  OBJ_DIR = os.path.join(CSRC, '_obj_dbg' if debug_hooks else '_obj')         # pylint: disable=invalid-name
  LIB_PATH = os.path.join(HERE, 'libdfhip_dbg.so' if debug_hooks else 'libdfhip.so')   # pylint: disable=invalid-name
  os.makedirs(OBJ_DIR, exist_ok=True)
  flags = CXXFLAGS + (['-DDFH_DEBUG_HOOKS'] if debug_hooks else []) + os.environ.get('DFH_EXTRA_CXXFLAGS', '').split()
  headers = HEADERS + ([os.path.join(HERE, '..', 'include', 'dfhip_debug.h')] if debug_hooks else [])
  stamp = os.path.join(OBJ_DIR, 'flags.txt')
  if not os.path.exists(stamp) or open(stamp).read() != ' '.join(flags):
    force = True                   # different flags than the objects on disk were built with
    with open(stamp, 'w') as f:
      f.write(' '.join(flags))
  jobs = []
  for src in SOURCES:
    src_path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ_DIR, src.replace('.hip', '.o'))
    if force or _stale(obj, [src_path] + headers):
      jobs.append((src_path, obj))

  def _compile(job):
    src_path, obj = job
    cmd = [hipcc] + flags + ['-c', src_path, '-o', obj]
    if verbose:
      print(' '.join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
      raise RuntimeError('hipcc failed on %s:\n%s\n%s' % (src_path, res.stdout, res.stderr))
    if verbose and res.stderr.strip():
      print(res.stderr, file=sys.stderr)

  with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as pool:
    list(pool.map(_compile, jobs))
  objs = [os.path.join(OBJ_DIR, s.replace('.hip', '.o')) for s in SOURCES]
  if force or jobs or _stale(LIB_PATH, objs):
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    if verbose:
      print(' '.join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
      raise RuntimeError('link failed:\n%s\n%s' % (res.stdout, res.stderr))
  return LIB_PATH


if __name__ == '__main__':
  print(build(force='--force' in sys.argv, debug_hooks='--debug-hooks' in sys.argv))
