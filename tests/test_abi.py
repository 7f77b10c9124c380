"""CPU: the C-ABI shared library loads and exports every symbol include/dfhip.h declares; the
ctypes table binds exactly that set; no compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from dragonfly_amd import _lib

HEADER = os.path.join(ROOT, 'include', 'dfhip.h')


def declared_functions():
  text = open(HEADER).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(dfh_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_functions():
  names = declared_functions()
  assert 'dfh_gp_fit' in names and 'dfh_kernel_matrix' in names and len(names) >= 25


def test_library_exports_every_declared_symbol():
  lib = _lib.load()
  for name in declared_functions():
    assert hasattr(lib, name), 'libdfhip.so does not export %s' % name
    assert isinstance(getattr(lib, name), ctypes._CFuncPtr)    # pylint: disable=protected-access


def test_ctypes_table_matches_header():
  assert sorted(_lib.SIGNATURES.keys()) == declared_functions()


def test_abi_version_and_error_string():
  lib = _lib.load()
  assert lib.dfh_abi_version() == 2
  assert isinstance(_lib.last_error(), str)


def test_kernel_desc_layout():
  """ struct dfh_kernel_desc: 2 x int32, 2 x double, pointer, int32 (+pad), 6 pointers; ABI version 2: + the
      three additive-factor pointers at the end """
  assert ctypes.sizeof(_lib.KernelDesc) == 112
  assert _lib.KernelDesc.group_factor.offset == 88 and _lib.KernelDesc.factor_scale.offset == 104
  assert _lib.KernelDesc.scale.offset == 8 and _lib.KernelDesc.bw.offset == 24
  assert _lib.KernelDesc.n_groups.offset == 32 and _lib.KernelDesc.group_off.offset == 40


def test_no_device_fails_loudly():
  """ there is no CPU fallback: without a GPU creating an engine raises """
  if _lib.device_count() > 0:
    pytest.skip('a GPU is present')
  from dragonfly_amd.engine import Engine
  with pytest.raises(_lib.DfhipError):
    Engine()
  from dragonfly_amd import kernel as K
  import numpy as np
  with pytest.raises(_lib.DfhipError):
    K.SEKernel(2, 1.0, [0.3, 0.3])(np.zeros((3, 2)))


def test_library_exports_nothing_beyond_the_header():
  """ every exported dfh_* symbol is declared in include/dfhip.h (diagnostics hooks live behind
      -DDFH_DEBUG_HOOKS / include/dfhip_debug.h and are absent from the product build) """
  import shutil
  import subprocess
  nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
  out = subprocess.run([nm, '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
  exported = sorted(set(re.findall(r'\bT (dfh_[a-z0-9_]+)\b', out)))
  assert exported == declared_functions()


def test_array_addresses_go_to_void_pointer_parameters_only():
  """ engine._ptr hands a NumPy array to the C-ABI as its address, an integer (no ctypes view per argument): ctypes takes
      an integer for a c_void_p parameter and for no typed pointer, so every _ptr(...) argument of every call of the
      library in the package must land on a c_void_p of the signature table -- and the array must be a named local of the
      calling function, alive across the call (an integer keeps nothing alive) """
  import ast
  pkg = os.path.join(ROOT, 'dragonfly_amd')
  seen = 0
  for fn in sorted(os.listdir(pkg)):
    if not fn.endswith('.py'):
      continue
    tree = ast.parse(open(os.path.join(pkg, fn)).read())
    for node in ast.walk(tree):
      if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in _lib.SIGNATURES:
        _, argtypes = _lib.SIGNATURES[node.func.attr]
        for i, arg in enumerate(node.args):
          if isinstance(arg, ast.Call) and isinstance(arg.func, ast.Name) and arg.func.id in ('_ptr', '_engine_ptr'):
            seen += 1
            assert argtypes[i] is ctypes.c_void_p, '%s:%d %s argument %d' % (fn, node.lineno, node.func.attr, i)
            assert len(arg.args) == 1 and isinstance(arg.args[0], ast.Name), \
                '%s:%d %s argument %d: a temporary would be freed before the call' % (fn, node.lineno, node.func.attr, i)
  assert seen >= 60


def test_flag_constants_of_the_binding_are_the_headers():
  """ the bits a caller ORs into `flags` (fit modes, round 6: the pointer-kind hints of dfh_gp_lml_batch) have one
      definition, the header's; the binding's copies must equal it """
  text = open(HEADER).read()
  defs = {m.group(1): int(m.group(2), 0) for m in re.finditer(r'^#define\s+(DFH_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)\b', text, flags=re.M)}
  assert defs['DFH_FIT_NO_JITTER'] == _lib.FIT_NO_JITTER
  assert defs['DFH_FIT_PROJECT_FIRST'] == _lib.FIT_PROJECT_FIRST and defs['DFH_FIT_TRY_BEFORE_PROJECT'] == _lib.FIT_TRY_BEFORE_PROJECT
  assert defs['DFH_LML_X_IS_DEVICE'] == _lib.LML_X_IS_DEVICE and defs['DFH_LML_Y_IS_HOST'] == _lib.LML_Y_IS_HOST
  # the hint bits do not collide with the fit modes
  assert (_lib.LML_X_IS_DEVICE | _lib.LML_Y_IS_HOST) & (_lib.FIT_NO_JITTER | _lib.FIT_PROJECT_FIRST | _lib.FIT_TRY_BEFORE_PROJECT) == 0
