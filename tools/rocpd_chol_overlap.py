"""The n = 16384 factorisation of one bench step in a rocprofv3 kernel trace (rocpd sqlite): for
every look-ahead trailing update U(k) (gemm_f64_la_kernel) the kernels of the NEXT panel that ran
inside its interval, then the raw timeline of a few panels.
  python tools/rocpd_chol_overlap.py trace.db [step_index=2] [first_row=0] [rows=90]"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_timeline import short   # noqa: E402


def main(path, step=2, first=0, nrows=90):
  c = sqlite3.connect(path)
  cols = [d[1] for d in c.execute('pragma table_info(kernels)')]
  key = 'stream_id' if 'stream_id' in cols else 'queue_id'
  rows = c.execute('select %s, start, end, name from kernels order by start' % key).fetchall()
  grams = [i for i, r in enumerate(rows) if 'kernmat_sym_kernel' in r[3] and r[2] - r[1] > 350e3]
  lo = grams[step]
  hi = next(i for i in range(lo, len(rows)) if 'kernmat_strip_kernel' in rows[i][3])
  seg = rows[lo:hi]
  t0 = seg[0][1]
  print('n = 16384 fit of bench step %d: %d launches, %.2f ms from the Gram kernel to the first cross matrix' %
        (step, len(seg), (seg[-1][2] - t0) / 1e6))
  print()
  print('look-ahead updates and what ran inside them (start offset, duration; D = panel_fused_kernel of the next panel,')
  print('I = trtri64 + inverse-assembly GEMMs, T = the row-solve GEMM of the next panel, cond = conditional refinement):')
  las = [r for r in seg if 'gemm_f64_la_kernel' in r[3]]
  for k, u in enumerate(las):
    inside = [r for r in seg if r is not u and r[1] >= u[1] and r[2] <= u[2]]
    names = {}
    for r in inside:
      names.setdefault(short(r[3]), []).append((r[2] - r[1]) / 1e3)
    desc = ', '.join('%s x%d (%.0f us)' % (n, len(v), sum(v)) for n, v in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:6])
    print('  U(%2d) +%8.1f us  %7.1f us : %s' % (k, (u[1] - t0) / 1e3, (u[2] - u[1]) / 1e3, desc))
  print()
  print('raw timeline, rows %d..%d (offset from the Gram kernel, duration, stream, kernel):' % (first, first + nrows))
  for r in seg[first:first + nrows]:
    print('%10.1f us  +%8.1f us  s%-3s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0], short(r[3])))


if __name__ == '__main__':
  a = sys.argv
  main(a[1], int(a[2]) if len(a) > 2 else 2, int(a[3]) if len(a) > 3 else 0, int(a[4]) if len(a) > 4 else 90)
