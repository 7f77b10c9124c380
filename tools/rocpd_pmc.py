"""Per-kernel PMC counter summary of a rocprofv3 rocpd database (`--pmc ...` run).
Usage: python tools/rocpd_pmc.py results.db"""
import sqlite3
import sys


def main(path):
  c = sqlite3.connect(path)
  views = [r[0] for r in c.execute("select name from sqlite_master where type='view'")]
  if 'counters_collection' not in views:
    print('no counters_collection view; views =', views)
    return
  cols = [d[1] for d in c.execute('pragma table_info(counters_collection)')]
  print('columns:', cols)
  name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
  q = ('select %s, counter_name, count(*), avg(value), sum(value), avg(end-start)/1e3 from counters_collection '
       'group by %s, counter_name order by 1, 2' % (name_col, name_col))
  print('%-60s %-28s %6s %16s %10s' % ('kernel', 'counter', 'calls', 'avg_value', 'avg_us'))
  for r in c.execute(q):
    print('%-60s %-28s %6d %16.1f %10.1f' % (str(r[0])[:60], r[1], r[2], r[3], r[5] or 0.0))


if __name__ == '__main__':
  main(sys.argv[1])
