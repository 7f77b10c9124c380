"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, one counter
per pass: they do not fit the TCC slots together) over the same command.
Usage: python tools/rocpd_pmc_traffic.py fetch.db write.db [kernel-substring] [out.json]
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of wide (16 B per
lane) coalesced reads (MI355X_MICROARCH.md, HBM section) -- the factor is re-measured on this
kernel's own access pattern by tools/pmc_workload.py's calibration GEMM and passed in as
FETCH_FACTOR (default 2.0); WRITE_SIZE*1024 is exact on the GEMM's stores (same calibration)."""
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


def main(fetch_db, write_db, filt='gemm_f64_kernel<false, false, 4>', out=None):
  factor = float(os.environ.get('FETCH_FACTOR', '2.0'))
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
  if out and res:
    with open(out, 'w') as fh:
      json.dump(res, fh, indent=1)
    print('wrote', out, res)


if __name__ == '__main__':
  main(*sys.argv[1:])
