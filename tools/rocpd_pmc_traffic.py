"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, one counter
per pass: they do not fit the TCC slots together) over the same command.
Usage: python tools/rocpd_pmc_traffic.py fetch.db write.db [kernel-substring] [out.json]
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of wide (16 B per
lane) coalesced reads (MI355X_MICROARCH.md, HBM section) -- the factor is re-measured on this
kernel's own access pattern by tools/pmc_workload.py's calibration GEMM and passed in as
FETCH_FACTOR (default 2.0); WRITE_SIZE*1024 is exact on the GEMM's stores (same calibration).
FIRST_LAUNCHES=N restricts the JSON figure to the kernel's first N dispatches: with
`bench.py --steps 1 --warmup 0` those are the launches of the timed step (N = the bench line's
roofline.launches_per_step); the later ones belong to the untimed extras (configs 2 and 5,
conditioning check) and have different shapes."""
import json
import os
import sqlite3
import sys


def per_kernel(path, counter):
  c = sqlite3.connect(path)
  cols = [d[1] for d in c.execute('pragma table_info(counters_collection)')]
  name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
  q = ('select %s, count(*), avg(value), avg(end-start)/1e3 from counters_collection where counter_name=? '
       'group by %s' % (name_col, name_col))
  return {r[0]: (r[1], r[2], r[3]) for r in c.execute(q, (counter,))}


def first_launches(path, counter, filt, num):
  """ (count, mean counter value) over the first `num` dispatches of the kernel matching filt """
  c = sqlite3.connect(path)
  cols = [d[1] for d in c.execute('pragma table_info(counters_collection)')]
  name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
  rows = [r[0] for r in c.execute('select value from counters_collection where counter_name=? and '
                                  'instr(%s, ?) > 0 order by start limit ?' % (name_col), (counter, filt, num))]
  return len(rows), (sum(rows) / len(rows) if rows else 0.0)


def main(fetch_db, write_db, filt='gemm_f64_kernel<false, false, 4>', out=None):
  factor = float(os.environ.get('FETCH_FACTOR', '2.0'))
  first = int(os.environ.get('FIRST_LAUNCHES', '0'))
  f = per_kernel(fetch_db, 'FETCH_SIZE')
  w = per_kernel(write_db, 'WRITE_SIZE')
  print('%-64s %7s %14s %14s %10s' % ('kernel', 'calls', 'read_MB/launch', 'write_MB/launch', 'avg_us'))
  res = None
  for k in sorted(f, key=lambda k: -f[k][0] * f[k][1]):
    rd = f[k][1] * 1024 * factor
    wr = w.get(k, (0, 0.0, 0.0))[1] * 1024
    print('%-64s %7d %14.2f %14.2f %10.1f' % (k[:64], f[k][0], rd / 1e6, wr / 1e6, f[k][2]))
    if filt in k and res is None:
      res = dict(kernel=k, launches=f[k][0], read_bytes_per_launch=rd, write_bytes_per_launch=wr,
                 hbm_bytes_per_launch=rd + wr, fetch_factor=factor)
  if res and first > 0:
    nf, vf = first_launches(fetch_db, 'FETCH_SIZE', filt, first)
    nw, vw = first_launches(write_db, 'WRITE_SIZE', filt, first)
    assert nf == nw == first, (nf, nw, first)
    res.update(launches=nf, read_bytes_per_launch=vf * 1024 * factor, write_bytes_per_launch=vw * 1024,
               hbm_bytes_per_launch=vf * 1024 * factor + vw * 1024,
               launch_subset='first %d dispatches = the timed step of bench.py --steps 1 --warmup 0' % first)
    print('first %d launches of %s: read %.2f MB, write %.2f MB per launch' % (
        first, filt, res['read_bytes_per_launch'] / 1e6, res['write_bytes_per_launch'] / 1e6))
  if out and res:
    with open(out, 'w') as fh:
      json.dump(res, fh, indent=1)
    print('wrote', out, res)


if __name__ == '__main__':
  main(*sys.argv[1:])
