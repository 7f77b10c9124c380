"""Builds oracle/_build/libldtruth.so from oracle/ld_truth.c (gcc, x87 long double, OpenMP).

    python oracle/build_truth.py [--force]

TEST INFRASTRUCTURE ONLY (see oracle/ld_truth.c).  __graft_entry__.build() calls build(); the
library is git-ignored but travels with the snapshot to the GPU box, where ref_longdouble.py
rebuilds it on demand if it is missing (gcc is part of the image).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'ld_truth.c')
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'libldtruth.so')


def build(force=False):
  os.makedirs(OUT_DIR, exist_ok=True)
  if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
    return LIB
  cmd = ['gcc', '-O2', '-fopenmp', '-shared', '-fPIC', '-o', LIB, SRC, '-lm']
  res = subprocess.run(cmd, capture_output=True, text=True)
  if res.returncode != 0:
    raise RuntimeError('gcc failed on %s:\n%s\n%s' % (SRC, res.stdout, res.stderr))
  return LIB


if __name__ == '__main__':
  print(build(force='--force' in sys.argv))
