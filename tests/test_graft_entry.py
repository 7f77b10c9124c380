"""CPU: the driver's hooks.  build() must return cleanly from this tree (round 3 shipped a hook that
asserted a stale ABI literal), and smoke()'s acceptance table must be the contract's 1e-10."""
import inspect

import __graft_entry__ as entry
from dragonfly_amd import _lib


def test_build_hook_returns_cleanly():
  entry.build()
  assert _lib.load().dfh_abi_version() == entry.header_abi_version()


def test_smoke_tolerances_are_the_contract():
  assert set(entry.SMOKE_TOL) == {'alpha', 'lml', 'mu', 'sd', 'ei', 'ts'}
  assert all(tol <= 1e-10 for tol in entry.SMOKE_TOL.values())
  src = inspect.getsource(entry.smoke)
  assert 'SMOKE_TOL' in src and 'argmax' in src
  # no literal tolerance besides the table
  assert '1e-' not in src
