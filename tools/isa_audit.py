"""Static audit of the gfx950 machine code of libdfhip.so: registers that are written and never read.

    python tools/isa_audit.py [path/to/libdfhip.so | file.s ...]      (exit status 1 if anything is flagged)

Why this exists (docs/NOTES_r05.md section 2): the "scalar wave index" build of panel_fused_kernel<true> that
produced NaN from column 4 on was a register-allocation fault of the compiler, not of the source or the
hardware.  Under the kernel's register pressure (512 of 512) the allocator split the spill of one 8-dword
accumulator tuple into  scratch_store_dwordx3 (dwords 0-2) + v_accvgpr_write a191 (dword 3, "Reload Reuse")
+ four more AGPR copies (dwords 4-7), and the reload put back seven of the eight pieces:  a191 is written
once and read nowhere, dword 3 -- the high half of the second element of the block's double4, i.e. columns
4..7 of strip 0 -- came back as whatever a195 last held.  The signature is a register with a write and no
read anywhere in the function; a correct compilation has none (such a write would have been deleted as dead
code), so that is what this looks for, kernel by kernel:
  * accumulator registers (aN) in every function -- the unified register file's AGPRs are the allocator's
    spill space on gfx90a+, they never carry call arguments;
  * vector registers (vN) in functions that make no call (arguments of a call are written and not read).
and, for the variant of the fault that parks the piece in scratch, fixed-offset scratch reloads of bytes that no
fixed-offset scratch store of the function writes (audit_scratch).
The checks are flow-insensitive (a read / a store anywhere counts), so they cannot prove a build right; they find this fault.

`python tools/isa_audit.py --resources [lib]` prints every kernel's registers, scratch bytes per lane, LDS and spill counts
from the code objects' metadata (tests/test_isa_audit.py holds the latency-chain kernels to their recorded scratch).

Inputs: the shipped library (the .hip_fatbin section is unbundled and disassembled with the ROCm LLVM tools)
or assembly files from `hipcc -save-temps` (the *-hip-amdgcn-*.s ones).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get('ROCM_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'

_REG = re.compile(r'\b([av])(\d+)\b|\b([av])\[(\d+):(\d+)\]')
# no destination register in the first operand: stores, LDS writes, scalar / control instructions, VOPC to vcc
# (matched against the WHOLE instruction text: whether an atomic returns a value is said by `sc0` among its modifiers
#  -- gfx940+ -- or by `_rtn` in an LDS atomic's mnemonic, not by the mnemonic's stem)
_NO_DEST = re.compile(r'^(global_store|scratch_store|flat_store|buffer_store|ds_write|ds_store|ds_gws|s_|v_cmpx?_\w+_e32|'
                      r'v_nop|exp|buffer_wbl2|buffer_inv|'
                      r'(global|flat|buffer)_atomic_(?!.*\bsc0\b)|'
                      r'ds_(add|sub|rsub|inc|dec|min|max|and|or|xor|mskor|cmpst|pk_add)_(?!rtn)\w*\s)')
_CALL = re.compile(r'^s_(swappc|setpc)_b64')


def _regs(text):
  out = set()
  for m in _REG.finditer(text):
    if m.group(1):
      out.add((m.group(1), int(m.group(2))))
    else:
      out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
  return out


def audit_function(lines):
  """lines: instruction texts ('mnemonic operands') of one function. Returns sorted [(reg, n_writes)]."""
  written, read = {}, set()
  tuples = []
  calls = False
  for ins in lines:
    ins = ins.split(';')[0].strip()
    if not ins:
      continue
    mnem, _, ops = ins.partition(' ')
    if _CALL.match(mnem):
      calls = True
    operands = [o.strip() for o in ops.split(',')] if ops else []
    if not operands:
      continue
    if _NO_DEST.match(ins):
      for o in operands:
        read |= _regs(o)
      continue
    dest = _regs(operands[0])
    for r in dest:
      written[r] = written.get(r, 0) + 1
    if len(dest) > 1:
      tuples.append(dest)
    for o in operands[1:]:
      read |= _regs(o)
  # a tuple destination (a dwordx4 load, an MFMA result) is one write: its registers are "never read" only if NONE of
  # them is -- a partly used tuple is ordinary code, not the fault
  partly_used = set()
  for dest in tuples:
    if any(r in read for r in dest):
      partly_used |= dest
  flagged = [(r, n) for r, n in written.items()
             if r not in read and r not in partly_used and (r[0] == 'a' or not calls)]
  return sorted(flagged)


def severity(reg):
  """ 'error' for what was root-caused (docs/NOTES_r05.md section 2): an ACCUMULATOR register written and never read --
      on this register file the AGPRs of a kernel under pressure are the allocator's spill space, a dead write there is
      a lost spill piece -- and a reload of unspilled scratch.  A written-never-read VECTOR register is a 'warning':
      the heuristic is flow-insensitive and an odd but correct code sequence can show it. """
  return 'error' if reg.startswith('a') or reg.startswith('scratch') else 'warning'


_SCRATCH = re.compile(r'^scratch_(load|store)_(dword(?:x(\d))?|[su]?byte|[su]?short)\w*\s+(.*)$')


def audit_scratch(lines):
  """ The same fault with the lost piece parked in scratch instead of an AGPR would read as a reload of bytes no spill
      wrote: scratch_load from a fixed offset (`off, off offset:K`) covering bytes that no fixed-offset scratch_store
      of the function writes.  Register-addressed scratch (real local arrays) is left alone.  Returns [(offset, text)]. """
  stored, loads = set(), []
  for ins in lines:
    ins = ins.split(';')[0].strip()
    m = _SCRATCH.match(ins)
    if not m:
      continue
    nbytes = 4 * int(m.group(3) or 1) if m.group(2).startswith('dword') else (1 if 'byte' in m.group(2) else 2)
    ops = [re.sub(r'\s*offset:.*', '', o.strip()) for o in m.group(4).split(',')]
    offm = re.search(r'offset:(-?\d+)', ins)
    off = int(offm.group(1)) if offm else 0
    addr = (ops[0:1] + ops[2:3]) if m.group(1) == 'store' else ops[1:3]
    if any(o != 'off' for o in addr):
      continue
    if m.group(1) == 'store':
      stored.update(range(off, off + nbytes))
    else:
      loads.append((off, nbytes, ins))
  return [(off, ins) for off, nbytes, ins in loads if any(b not in stored for b in range(off, off + nbytes))]


def functions_of_asm(path):
  """-save-temps assembly: yield (name, [instruction text]) per function."""
  name, body = None, []
  for line in open(path, errors='replace'):
    s = line.rstrip('\n')
    m = re.match(r'^([A-Za-z_][\w$.]*):\s*(;.*)?$', s)
    if m and not s.startswith('.L'):
      if name and body:
        yield name, body
      name, body = m.group(1), []
      continue
    if s.startswith('.Lfunc_end'):
      if name and body:
        yield name, body
      name, body = None, []
      continue
    if name and s.startswith('\t') and not s.lstrip().startswith(('.', ';')):
      body.append(s.strip())
  if name and body:
    yield name, body


def _code_objects(path, tmp):
  """Unbundle every gfx950 code object of a hipcc-built shared library into tmp; yields their paths."""
  fat = os.path.join(tmp, 'fat.bin')
  subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', path, fat], check=True)
  blob = open(fat, 'rb').read()
  starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
  for i, st in enumerate(starts):
    piece = os.path.join(tmp, 'bundle%d.bin' % i)
    with open(piece, 'wb') as f:
      f.write(blob[st:starts[i + 1] if i + 1 < len(starts) else len(blob)])
    co = os.path.join(tmp, 'co%d.o' % i)
    subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--targets=' + TARGET,
                    '--input=' + piece, '--output=' + co], check=True, capture_output=True)
    yield co


def functions_of_library(path):
  """Shared library built by hipcc: yield (name, [instruction text]) for every gfx950 function in it."""
  with tempfile.TemporaryDirectory() as tmp:
    for co in _code_objects(path, tmp):
      dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', '--no-leading-addr', co],
                           check=True, capture_output=True, text=True).stdout
      name, body = None, []
      for line in dis.splitlines():
        m = re.match(r'^(?:[0-9a-f]+ )?<([^>]+)>:$', line.strip())
        if m:
          if name and body:
            yield name, body
          name, body = m.group(1), []
        elif name and line.startswith(('\t', ' ')) and line.strip():
          body.append(re.sub(r'\s*//.*$', '', line.strip()))
      if name and body:
        yield name, body


_META = ('.agpr_count', '.vgpr_count', '.sgpr_count', '.private_segment_fixed_size', '.group_segment_fixed_size',
         '.vgpr_spill_count', '.sgpr_spill_count')


def resource_usage(path):
  """Kernel -> {vgpr, agpr, sgpr, scratch (bytes per lane), lds, vgpr_spill, sgpr_spill}, from the code objects' metadata
  notes (what -Rpass-analysis=kernel-resource-usage prints at compile time, read back from what was built)."""
  out = {}
  with tempfile.TemporaryDirectory() as tmp:
    for co in _code_objects(path, tmp):
      notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], check=True, capture_output=True, text=True).stdout
      cur = None
      for line in notes.splitlines():
        m = re.match(r'^(  - |    )(\.[a-z_]+):\s+(\S+)\s*$', line)        # kernel-level keys only (arguments sit deeper)
        if not m:
          continue
        if m.group(1) == '  - ':
          cur = {}
          out[len(out)] = cur
        if cur is None:
          continue
        key, val = m.group(2), m.group(3)
        if key in _META:
          cur[key] = int(val)
        elif key == '.name':
          cur['.name'] = val
  res = {}
  for k in out.values():
    if '.name' in k:
      res[k['.name']] = {'vgpr': k.get('.vgpr_count', 0), 'agpr': k.get('.agpr_count', 0), 'sgpr': k.get('.sgpr_count', 0),
                         'scratch': k.get('.private_segment_fixed_size', 0), 'lds': k.get('.group_segment_fixed_size', 0),
                         'vgpr_spill': k.get('.vgpr_spill_count', 0), 'sgpr_spill': k.get('.sgpr_spill_count', 0)}
  return res


def demangle(names):
  import shutil
  filt = shutil.which('llvm-cxxfilt') or shutil.which('c++filt')
  if not filt:
    return list(names)
  p = subprocess.run([filt], input='\n'.join(names), capture_output=True, text=True, check=True)
  return p.stdout.split('\n')[:len(names)]


def audit(path):
  funcs = functions_of_asm(path) if path.endswith('.s') else functions_of_library(path)
  n, bad = 0, []
  for name, body in funcs:
    if name.startswith('.L') or '$local' in name:
      continue
    n += 1
    for reg, writes in audit_function(body):
      bad.append((name, '%s%d' % reg, writes))
    for off, ins in audit_scratch(body):
      bad.append((name, 'scratch+%d (%s)' % (off, ins), 0))
  return n, bad


def main(argv):
  if argv and argv[0] == '--resources':
    lib = argv[1] if len(argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'dragonfly_amd', 'libdfhip.so')
    ru = resource_usage(lib)
    names = sorted(ru, key=lambda k: (-ru[k]['scratch'], k))
    print('%-6s %-5s %-5s %-8s %-7s %-7s %-7s  kernel' % ('vgpr', 'agpr', 'sgpr', 'scratch', 'lds', 'vspill', 'sspill'))
    for k, d in zip(names, demangle(names)):
      r = ru[k]
      print('%-6d %-5d %-5d %-8d %-7d %-7d %-7d  %s' % (r['vgpr'], r['agpr'], r['sgpr'], r['scratch'], r['lds'], r['vgpr_spill'],
                                                      r['sgpr_spill'], d.replace('(anonymous namespace)::', '')))
    return 0
  paths = argv or [os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'dragonfly_amd', 'libdfhip.so')]
  rc = 0
  for p in paths:
    n, bad = audit(p)
    print('%s: %d functions, %d findings (registers written and never read, scratch reloads never spilled)' % (os.path.relpath(p), n, len(bad)))
    for name, reg, writes in bad:
      print('  %-7s %s  in  %s  (%s)' % (severity(reg), reg, name, 'reloaded, never spilled' if writes == 0 else '%d write%s, no read' % (writes, '' if writes == 1 else 's')))
      if severity(reg) == 'error':
        rc = 1
  return rc


if __name__ == '__main__':
  sys.exit(main(sys.argv[1:]))
