"""Timeline of the LAST factorisation in a rocprofv3 kernel trace (rocpd sqlite): every kernel launch
from the last Gram-matrix kernel on, with start offset, duration, stream and a short name.
  python tools/rocpd_timeline.py trace.db [max_rows]"""
import re
import sqlite3
import sys


def short(name):
  name = re.sub(r'\(anonymous namespace\)::', '', name)
  m = re.match(r'(?:void )?([A-Za-z0-9_]+)(<[^(]*>)?', name)
  base = m.group(1) if m else name[:30]
  targs = (m.group(2) or '') if m else ''
  return (base + targs)[:44]


def main(path, max_rows=100000):
  c = sqlite3.connect(path)
  tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
  kt = [t for t in tabs if t == 'kernels'] or [t for t in tabs if 'kernel' in t.lower()]
  cols = [d[1] for d in c.execute('pragma table_info(%s)' % kt[0])]
  key = 'stream_id' if 'stream_id' in cols else 'queue_id'
  rows = c.execute('select %s, start, end, name from %s order by start' % (key, kt[0])).fetchall()
  last_gram = max(i for i, r in enumerate(rows) if 'kernmat' in r[3])
  rows = rows[last_gram:]
  t0 = rows[0][1]
  for r in rows[:max_rows]:
    print('%10.1f us  +%8.1f us  s%-3s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[0], short(r[3])))
  print('total %.1f us over %d launches' % ((max(r[2] for r in rows) - t0) / 1e3, len(rows)))


if __name__ == '__main__':
  main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 100000)
