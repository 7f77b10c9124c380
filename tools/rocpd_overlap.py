"""Concurrency analysis of a rocprofv3 kernel trace (rocpd sqlite): per stream/queue busy time,
pairwise overlap, and union busy time over the last `window_s` seconds of the trace."""
import sqlite3
import sys


def main(path, window_s=2.5):
  c = sqlite3.connect(path)
  cols = [d[1] for d in c.execute('pragma table_info(kernels)')]
  key = 'stream_id' if 'stream_id' in cols else 'queue_id'
  rows = c.execute('select %s, start, end, name from kernels order by start' % key).fetchall()
  t_end = max(r[2] for r in rows)
  t0 = t_end - int(window_s * 1e9)
  rows = [r for r in rows if r[1] >= t0]
  print('key column:', key, ' kernels in window:', len(rows))
  streams = sorted(set(r[0] for r in rows))
  iv = {s: [(r[1], r[2]) for r in rows if r[0] == s] for s in streams}

  def merge(ivs):
    ivs = sorted(ivs)
    out = []
    for a, b in ivs:
      if out and a <= out[-1][1]:
        out[-1][1] = max(out[-1][1], b)
      else:
        out.append([a, b])
    return out

  def total(ivs):
    return sum(b - a for a, b in ivs) / 1e6

  def inter(x, y):
    i = j = 0
    tot = 0
    while i < len(x) and j < len(y):
      a = max(x[i][0], y[j][0]); b = min(x[i][1], y[j][1])
      if b > a:
        tot += b - a
      if x[i][1] < y[j][1]:
        i += 1
      else:
        j += 1
    return tot / 1e6

  merged = {s: merge(v) for s, v in iv.items()}
  for s in streams:
    names = {}
    for r in rows:
      if r[0] == s:
        names[r[3][:40]] = names.get(r[3][:40], 0) + (r[2] - r[1]) / 1e6
    top = sorted(names.items(), key=lambda kv: -kv[1])[:3]
    print('%s %s: busy %.1f ms over %d kernels; top %s' % (key, s, total(merged[s]), len(iv[s]), top))
  for i, a in enumerate(streams):
    for b in streams[i + 1:]:
      print('overlap %s & %s: %.1f ms' % (a, b, inter(merged[a], merged[b])))
  allm = merge([x for v in iv.values() for x in v])
  print('union busy %.1f ms of window %.1f ms' % (total(allm), (t_end - t0) / 1e6))


if __name__ == '__main__':
  main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 2.5)
