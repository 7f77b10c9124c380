"""Idle gaps and long kernels in a rocprofv3 kernel trace (rocpd sqlite):
   python tools/rocpd_gaps.py trace.db [min_gap_us=500] [min_kernel_us=5000]"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_timeline import short   # noqa: E402

c = sqlite3.connect(sys.argv[1])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
min_k = float(sys.argv[3]) if len(sys.argv) > 3 else 5000.0
cols = [d[1] for d in c.execute('pragma table_info(kernels)')]
key = 'stream_id' if 'stream_id' in cols else 'queue_id'
rows = c.execute('select %s, start, end, name from kernels order by start' % key).fetchall()
t0 = rows[0][1]
busy_until = rows[0][2]
for i, r in enumerate(rows):
  if (r[1] - busy_until) / 1e3 > min_gap:
    p = rows[i - 1]
    print('%10.2f ms  GAP %8.1f us  after %s (s%s), before %s (s%s)' % ((busy_until - t0) / 1e6, (r[1] - busy_until) / 1e3, short(p[3]), p[0], short(r[3]), r[0]))
  if (r[2] - r[1]) / 1e3 > min_k:
    print('%10.2f ms  KERNEL %8.1f us  %s (s%s)' % ((r[1] - t0) / 1e6, (r[2] - r[1]) / 1e3, short(r[3]), r[0]))
  busy_until = max(busy_until, r[2])
print('%d launches, %.2f ms' % (len(rows), (busy_until - t0) / 1e6))
