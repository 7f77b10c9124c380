"""Per-dispatch PMC listing of a rocprofv3 --pmc rocpd database, with the derived figures used in
profiles/ (MFMA utilisation, effective clock, executed TF/s, HBM bytes).
Usage: python tools/rocpd_pmc_dispatch.py results.db [kernel-substring ...]"""
import sqlite3
import sys
from collections import OrderedDict


def main(path, filts):
  c = sqlite3.connect(path)
  rows = c.execute('select dispatch_id, kernel_name, grid_size, counter_name, value, (end-start)/1e3 '
                   'from counters_collection order by dispatch_id').fetchall()
  disp = OrderedDict()
  for did, name, grid, cname, val, us in rows:
    d = disp.setdefault(did, dict(name=name, grid=grid, us=us, c={}))
    d['c'][cname] = d['c'].get(cname, 0.0) + val
  shown = {}
  for did, d in disp.items():
    if filts and not any(f in d['name'] for f in filts):
      continue
    key = (d['name'], d['grid'] // 65536 if 'diag_step' in d['name'] else d['grid'])
    shown[key] = shown.get(key, 0) + 1
    if shown[key] > 3:          # at most three dispatches per kernel and launch shape
      continue
    if 'gemm_f64' in d['name'] and d['us'] < 500.0:     # the factorisation's many small GEMMs
      continue
    cs = d['c']
    extra = ''
    if 'GRBM_GUI_ACTIVE' in cs and 'SQ_VALU_MFMA_BUSY_CYCLES' in cs:
      cyc = cs['GRBM_GUI_ACTIVE'] / 8.0                         # summed over the 8 XCDs
      util = cs['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0)    # 1024 SIMDs
      tf = cs.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0.0) * 512.0 / (d['us'] * 1e-6) / 1e12
      extra = ' | clock %.2f GHz MfmaUtil %.1f%% executed %.1f TF/s' % (cyc / (d['us'] * 1e3), 100 * util, tf)
    if 'SQ_LDS_BANK_CONFLICT' in cs and cs.get('SQ_LDS_IDX_ACTIVE', 0) > 0:
      extra = ' | LDS bank-conflict cycles / active cycles = %.3f' % (cs['SQ_LDS_BANK_CONFLICT'] / cs['SQ_LDS_IDX_ACTIVE'])
    if 'FETCH_SIZE' in cs:
      b = cs['FETCH_SIZE'] * 1024 * 2
      extra = ' | 2*FETCH_SIZE*1024 = %.3f GB -> %.2f TB/s' % (b / 1e9, b / (d['us'] * 1e-6) / 1e12)
    if 'WRITE_SIZE' in cs:
      b = cs['WRITE_SIZE'] * 1024
      extra = ' | WRITE_SIZE*1024 = %.3f GB -> %.2f TB/s' % (b / 1e9, b / (d['us'] * 1e-6) / 1e12)
    print('disp %4d %-58s grid %9d %9.1f us %s%s' % (did, d['name'][:58], d['grid'], d['us'],
                                                   ' '.join('%s=%g' % kv for kv in sorted(cs.items())), extra))


if __name__ == '__main__':
  main(sys.argv[1], sys.argv[2:])
