"""The built library's machine code holds no register that is written and never read.

That is the signature of the compiler fault behind the "scalar wave index" build of panel_fused_kernel<true>
(docs/NOTES_r05.md section 2, profiles/r05_scalar_w_root_cause.txt): one dword of a spilled accumulator tuple
parked in an AGPR and never put back.  It depends on the register allocation, not on the source, so it is checked
on what was actually built -- here on CPU, with the ROCm LLVM tools.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_audit  # noqa: E402  pylint: disable=wrong-import-position

LIB = os.path.join(ROOT, 'dragonfly_amd', 'libdfhip.so')

# the fault as it stood in the faulty build, cut to the registers involved: the spill site, and the reload of the strip-0 leaf
_FAULTY = '''
v_mov_b64_e32 v[10:11], v[82:83]
v_mov_b64_e32 v[12:13], v[84:85]
v_mov_b64_e32 v[14:15], v[86:87]
v_mov_b64_e32 v[16:17], v[88:89]
v_accvgpr_write_b32 a191, v13           ;  Reload Reuse
scratch_store_dwordx3 off, v[10:12], off offset:384 ; 12-byte Folded Spill
v_accvgpr_write_b32 a187, v17           ;  Reload Reuse
v_accvgpr_write_b32 a188, v16           ;  Reload Reuse
v_accvgpr_write_b32 a189, v15           ;  Reload Reuse
v_accvgpr_write_b32 a190, v14           ;  Reload Reuse
scratch_load_dwordx3 a[192:194], off, off offset:384 ; 12-byte Folded Reload
s_waitcnt vmcnt(0)
v_accvgpr_mov_b32 a199, a187
v_accvgpr_mov_b32 a198, a188
v_accvgpr_mov_b32 a197, a189
v_accvgpr_mov_b32 a196, a190
v_accvgpr_read_b32 v0, a192
v_accvgpr_read_b32 v1, a193
v_accvgpr_read_b32 v2, a194
v_accvgpr_read_b32 v3, a195
v_accvgpr_read_b32 v4, a196
v_accvgpr_read_b32 v5, a197
v_accvgpr_read_b32 v6, a198
v_accvgpr_read_b32 v7, a199
ds_write2_b64 v20, v[0:1], v[2:3] offset1:4
ds_write2_b64 v20, v[4:5], v[6:7] offset0:8 offset1:12
s_endpgm
'''.strip().splitlines()


def test_the_fault_is_what_the_audit_finds():
  assert isa_audit.audit_function(_FAULTY) == [(('a', 191), 1)]
  repaired = list(_FAULTY)
  repaired.insert(repaired.index('v_accvgpr_mov_b32 a196, a190') + 1, 'v_accvgpr_mov_b32 a195, a191')
  assert isa_audit.audit_function(repaired) == []


def test_a_reload_of_bytes_nobody_spilled_is_found_too():
  spill = ['scratch_store_dwordx3 off, v[10:12], off offset:384', 'scratch_store_dwordx4 off, v[98:101], off offset:352']
  assert isa_audit.audit_scratch(spill + ['scratch_load_dwordx3 a[192:194], off, off offset:384']) == []
  assert [o for o, _ in isa_audit.audit_scratch(spill + ['scratch_load_dwordx4 a[192:195], off, off offset:384'])] == [384]
  assert isa_audit.audit_scratch(['scratch_load_dword v1, v2, off offset:16']) == []          # a local array: not a spill slot
  assert isa_audit.audit_scratch(['scratch_store_dword off, v1, off', 'scratch_load_dword v3, off, off']) == []


def test_operands_are_sorted_into_reads_and_writes():
  # stores and LDS writes have no destination; a call makes written-only vector registers legitimate (arguments)
  assert isa_audit.audit_function(['v_mov_b32_e32 v1, 0', 'global_store_dword v2, v1, s[0:1]']) == []
  assert isa_audit.audit_function(['v_mov_b32_e32 v1, 0', 's_endpgm']) == [(('v', 1), 1)]
  assert isa_audit.audit_function(['v_mov_b32_e32 v1, 0', 's_swappc_b64 s[30:31], s[4:5]']) == []
  assert isa_audit.audit_function(['v_accvgpr_write_b32 a7, v1', 's_swappc_b64 s[30:31], s[4:5]']) == [(('a', 7), 1)]
  assert isa_audit.audit_function(['v_mfma_f64_16x16x4_f64 a[0:7], v[0:1], v[2:3], a[0:7]',
                                   'v_accvgpr_read_b32 v4, a3', 'ds_write_b32 v5, v4']) == []   # read as the addend


@pytest.mark.skipif(not os.path.exists(os.path.join(isa_audit.LLVM, 'llvm-objdump')), reason='ROCm LLVM tools not found')
def test_built_library_has_no_written_never_read_register():
  assert os.path.exists(LIB), 'libdfhip.so is not built (python -m dragonfly_amd.build)'
  n, bad = isa_audit.audit(LIB)
  assert n >= 100, 'only %d functions disassembled' % n
  errors = [b for b in bad if isa_audit.severity(b[1]) == 'error']
  assert not errors, 'written-never-read accumulator registers / unspilled reloads (tools/isa_audit.py): %r' % (errors,)
  assert not bad, 'written-never-read vector registers (a warning for build(), recorded as clean here): %r' % (bad,)


def test_atomics_and_partly_used_tuples_are_not_findings():
  """ round 6 (advisor): whether an atomic has a destination is said by its modifiers / `_rtn`, not by the mnemonic's
      stem; a tuple destination of which SOME registers are read is ordinary code. """
  # a returning global atomic writes v5 (and v5 is used); a non-returning one only reads its operands
  assert isa_audit.audit_function(['global_atomic_add_u32 v5, v1, v2, s[0:1] sc0', 'global_store_dword v1, v5, s[2:3]']) == []
  assert isa_audit.audit_function(['global_atomic_add_u32 v5, v1, v2, s[0:1] sc0', 's_endpgm']) == [(('v', 5), 1)]
  assert isa_audit.audit_function(['v_mov_b32_e32 v1, 0', 'global_atomic_add_u32 v0, v1, s[0:1]', 's_endpgm']) == []
  # flat / LDS atomics without return: the address register is a READ, not a destination
  assert isa_audit.audit_function(['v_mov_b32_e32 v3, 0', 'v_mov_b32_e32 v4, 1', 'flat_atomic_add v[3:4], v4', 's_endpgm']) == []
  assert isa_audit.audit_function(['v_mov_b32_e32 v3, 0', 'v_mov_b32_e32 v4, 1', 'ds_add_u32 v3, v4', 's_endpgm']) == []
  assert isa_audit.audit_function(['v_mov_b32_e32 v3, 0', 'ds_add_rtn_u32 v7, v3, v3', 's_endpgm']) == [(('v', 7), 1)]
  # a dwordx4 load of which two dwords are used
  assert isa_audit.audit_function(['global_load_dwordx4 v[8:11], v0, s[0:1]', 'global_store_dwordx2 v0, v[8:9], s[2:3]']) == []
  assert isa_audit.severity('a191') == 'error' and isa_audit.severity('v7') == 'warning' and isa_audit.severity('scratch+384 (x)') == 'error'


# scratch bytes per lane as recorded in profiles/r05_kernel_resource_usage.txt: the throughput kernels have none, the
# latency-chain kernels of the factorisation must not get worse than what the round's timings were measured with
_NO_SCRATCH = ('gemm_f64_kernel', 'lml_wg_kernel', 'kernmat_sym', 'kernmat_strip', 'trtri64_kernel', 'k_lml_tiny', 'k_pack_fused')
_SCRATCH_CEILING = {'panel_fused_kernelILb1E': 388, 'panel_fused_kernelILb0E': 188, 'panel_strip_kernel': 228,
                    'lml_team_kernel': 52, 'diag_step64_kernel': 48, 'gemm_f64_cond_kernel': 44, 'gemm_f64_la_kernel': 12}


@pytest.mark.skipif(not os.path.exists(os.path.join(isa_audit.LLVM, 'llvm-readelf')), reason='ROCm LLVM tools not found')
def test_scratch_of_the_built_kernels_is_what_was_recorded():
  usage = isa_audit.resource_usage(LIB)
  assert len(usage) >= 80, 'only %d kernels found in the metadata' % len(usage)
  seen = set()
  for name, r in usage.items():
    for key in _NO_SCRATCH:
      if key in name:
        seen.add(key)
        assert r['scratch'] == 0, '%s: %d B/lane of scratch (was 0)' % (name, r['scratch'])
    for key, ceiling in _SCRATCH_CEILING.items():
      if key in name:
        seen.add(key)
        assert r['scratch'] <= ceiling, '%s: %d B/lane of scratch (recorded: %d)' % (name, r['scratch'], ceiling)
  missing = (set(_NO_SCRATCH) | set(_SCRATCH_CEILING)) - seen
  assert not missing, 'kernels not found in the library: %r' % sorted(missing)
