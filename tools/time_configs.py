"""BASELINE configs 2 and 5 at full size (bench.other_configs): ms, sections, row-solve fraction.
   python tools/time_configs.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dragonfly_amd.engine import get_engine
eng = get_engine()
out = bench.other_configs(eng)
for c in ('C2', 'C5'):
  r = out[c]
  print(c, 'ms', r['ms'], 'trsm_frac', r['trsm_frac_of_fp64_mfma_peak'], 'sections', r['sections_ms'], 'argmax', r.get('argmax', r.get('argmax_per_group', '-')) if not isinstance(r.get('argmax_per_group'), list) else 'groups')
