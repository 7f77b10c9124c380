"""CPU: the oracle's restatement of the polynomial / exponential-decay kernels against the real
reference's matrices, and the mirrors' kernels / GP / multi-fidelity GP over the stand-in engine
against the real reference's posteriors."""
import numpy as np

from conftest import load_golden, relerr
from oracle import ref_numpy as O
from polyexp_replay import check


def test_oracle_poly_and_expdecay_match_the_reference():
  g = load_golden('poly_expdecay')
  for order in (1, 2, 3, 5):
    spec = O.KernelSpec('poly', 3, 1.7, g['scalings'], nu=order)
    assert np.array_equal(spec(g['X1'], g['X2']), g['poly%d_K12' % order])
    assert np.array_equal(spec(g['X1']), g['poly%d_K11' % order])
  spec = O.KernelSpec('expdecay', 2, float(g['ed_scale']), g['powers'], nu=float(g['ed_offset']))
  assert np.array_equal(spec(g['Z1'], g['Z2']), g['ed_K12'])
  assert np.array_equal(spec(g['Z1']), g['ed_K11'])
  # the multi-fidelity GP of the fixture through the oracle's GP
  fd, dd = g['ZZ'].shape[1], g['XX'].shape[1]
  subs = [O.KernelSpec('expdecay', fd, 1.0, g['f_powers'], nu=float(g['f_offset'])), O.KernelSpec('se', dd, 1.0, g['dbw'])]
  prod = O.KernelSpec('product', fd + dd, float(g['scale']), groups=[list(range(fd)), list(range(fd, fd + dd))], subs=subs)
  ZX = np.concatenate([g['ZZ'], g['XX']], axis=1)
  og = O.GPOracle(ZX, g['YY'], prod, float(g['mean_c']), float(g['noise']))
  assert np.array_equal(prod(ZX), g['K'])
  assert relerr(og.alpha, g['alpha']) < 1e-12
  mu, sd = og.eval(np.concatenate([g['Zs'], g['Xs']], axis=1), 'std')
  assert relerr(mu, g['mu']) < 1e-12 and relerr(sd, g['sd']) < 1e-11


def test_mirrors_over_the_stand_in_engine(monkeypatch):
  from oracle_engine import patch_engine
  from dragonfly_amd import mf_gp    # noqa: F401  (imports gp_core before patching)
  patch_engine(monkeypatch)
  check(tol=1e-10)
