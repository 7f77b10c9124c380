"""Per-kernel summary (count / total / avg / min / max) of a rocprofv3 rocpd sqlite database,
the default output of `rocprofv3 --kernel-trace --stats` on ROCm 7.2.
Usage: python tools/rocpd_stats.py results.db [substring-filter]"""
import sqlite3
import sys


def main(path, filt=None):
  c = sqlite3.connect(path)
  rows = c.execute('select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, '
                   'min(end-start)/1e3, max(end-start)/1e3 from kernels group by name '
                   'order by 3 desc').fetchall()
  tot = sum(r[2] for r in rows)
  print('%-72s %7s %11s %6s %10s %9s %10s' % ('kernel', 'calls', 'total_ms', '%', 'avg_us', 'min_us', 'max_us'))
  for r in rows:
    if filt and filt not in r[0]:
      continue
    print('%-72s %7d %11.3f %6.1f %10.1f %9.1f %10.1f' % (r[0][:72], r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5]))
  print('total kernel time %.3f ms' % tot)


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
