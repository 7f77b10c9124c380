"""dragonfly_amd.install: the batched fitter's direct decode of a tuning candidate (install.py: _decode_candidates)
against what the reference's own build_gp makes of the same vector (dragonfly/gp/gp_core.py:501-543,
gp/euclidean_gp.py:325-339, 796-897) -- kernel description, constant mean and noise variance equal to the last bit,
over the fitter options that change the layout of the hyper-parameter vector.  CPU, needs the reference tree."""
import os
import warnings

from argparse import Namespace

import numpy as np
import pytest

from oracle_engine import patch_engine

REF = os.environ.get('DRAGONFLY_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'dragonfly')),
                                reason='needs the reference tree (build container only)')

DIM = 5


def _fitter(monkeypatch, **options):
  from oracle.make_golden import import_reference
  import_reference()
  patch_engine(monkeypatch)
  from dragonfly_amd import install
  install.install()
  import dragonfly.opt.gp_bandit as GB
  from dragonfly.gp.euclidean_gp import euclidean_gp_args
  from dragonfly.utils.option_handler import load_options
  rs = np.random.RandomState(3)
  X = [rs.rand(DIM) for _ in range(30)]
  Y = [float(np.sin(3 * x).sum() + 0.1 * rs.randn()) for x in X]
  opts = load_options(euclidean_gp_args, partial_options=options)
  if options.get('noise_var_type') == 'label':
    Y = np.array(Y)           # gp_core.py:536 calls self.Y.std(): labels as an array, or the reference itself fails
  with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    fitter = GB.EuclideanGPFitter(X, Y, options=opts)
    fitter._set_up()          # pylint: disable=protected-access
  return fitter, install


def _candidates(fitter, num, rs, groupings=None):
  lo = np.array([b[0] for b in fitter.cts_hp_bounds], dtype=float)
  hi = np.array([b[1] for b in fitter.cts_hp_bounds], dtype=float)
  cts = [lo + (hi - lo) * rs.rand(len(lo)) for _ in range(num)]
  dscr = [[vals[rs.randint(len(vals))] for vals in fitter.dscr_hp_vals] for _ in range(num)]
  return cts, dscr


OPTIONS = [
  dict(kernel_type='se'),
  dict(kernel_type='matern'),                                     # nu = 2.5, the option's default
  dict(kernel_type='matern', matern_nu=-1.0),                     # nu tuned: a discrete hyper-parameter
  dict(kernel_type='matern', matern_nu=2.5),
  dict(kernel_type='se', use_same_bandwidth=True),
  dict(kernel_type='matern', matern_nu=1.5, use_same_bandwidth=True),
  dict(kernel_type='se', mean_func_type='median', noise_var_type='label', noise_var_label=0.03),
  dict(kernel_type='matern', mean_func_type='mean', noise_var_type='value', noise_var_value=0.02),
  dict(kernel_type='se', mean_func_type='const', mean_func_const=0.7),
  dict(kernel_type='se', mean_func_type='zero'),
]


@pytest.mark.parametrize('options', OPTIONS, ids=lambda o: '-'.join('%s=%s' % kv for kv in sorted(o.items())))
def test_decode_equals_build_gp(monkeypatch, options):
  fitter, install = _fitter(monkeypatch, **options)
  try:
    rs = np.random.RandomState(17)
    cts, dscr = _candidates(fitter, 40, rs)
    got = fitter._decode_candidates(cts, dscr, True, None)              # pylint: disable=protected-access
    assert got is not None and fitter._amd_decode['ok']                  # pylint: disable=protected-access
    specs, means, noises = got
    for c, ds, sp, mean, noise in zip(cts, dscr, specs, means, noises):
      want = fitter._build_one(c, list(ds))                              # pylint: disable=protected-access
      assert want == (sp.signature(), mean, noise)
    # one list of discrete values for all candidates (the tuners' other calling convention)
    got1 = fitter._decode_candidates(cts[:5], dscr[0], False, None)     # pylint: disable=protected-access
    assert [s.signature() for s in got1[0]] == [fitter._build_one(c, list(dscr[0]))[0] for c in cts[:5]]    # pylint: disable=protected-access
  finally:
    install.uninstall()


@pytest.mark.parametrize('kernel_type', ['se', 'matern'])
def test_decode_equals_build_gp_additive(monkeypatch, kernel_type):
  fitter, install = _fitter(monkeypatch, kernel_type=kernel_type, use_additive_gp=True)
  try:
    rs = np.random.RandomState(23)
    cts, dscr = _candidates(fitter, 25, rs)
    other = Namespace(add_gp_groupings=[[np.int64(3), np.int64(0)], [np.int64(4)], [np.int64(1), np.int64(2)]])
    specs, means, noises = fitter._decode_candidates(cts, dscr, True, other)     # pylint: disable=protected-access
    for c, ds, sp, mean, noise in zip(cts, dscr, specs, means, noises):
      assert fitter._build_one(c, list(ds), other) == (sp.signature(), mean, noise)   # pylint: disable=protected-access
    assert specs[0].kind == 'additive' and specs[0].groups == [[3, 0], [4], [1, 2]]
  finally:
    install.uninstall()


def test_what_the_decode_leaves_to_the_general_route(monkeypatch):
  fitter, install = _fitter(monkeypatch, kernel_type='matern', matern_nu=-1.0)
  try:
    rs = np.random.RandomState(5)
    cts, dscr = _candidates(fitter, 3, rs)
    assert len(dscr[0]) == 1 and fitter._decode_candidates(cts, dscr, True, None) is not None
    assert fitter._decode_candidates([c[:-1] for c in cts], dscr, True, None) is None      # a vector of the wrong length
    assert fitter._decode_candidates([[float(v) for v in c] for c in cts], dscr, True, None) is None   # plain floats: no .item()
    assert fitter._decode_candidates(cts, [[2.0]] * 3, True, None) is None                 # nu not a half-integer
    assert fitter._amd_decode['ok']                                                         # none of these is a verdict
    # a difference against build_gp in one of the first calls switches the decode off for this fitter
    fitter._amd_decode['checks'] = 0
    monkeypatch.setattr(type(fitter), '_build_one', lambda self, *a: None)
    assert fitter._decode_candidates(cts, dscr, True, None) is None and not fitter._amd_decode['ok']
    assert fitter._decode_candidates(cts, dscr, True, None) is None
  finally:
    install.uninstall()


def test_a_user_mean_function_keeps_the_general_route(monkeypatch):
  fitter, install = _fitter(monkeypatch, kernel_type='se')
  try:
    fitter.options.mean_func = lambda x: np.zeros(len(x))
    fitter._amd_decode = None                                            # pylint: disable=protected-access
    rs = np.random.RandomState(5)
    cts, dscr = _candidates(fitter, 2, rs)
    assert fitter._decode_candidates(cts, dscr, True, None) is None     # pylint: disable=protected-access
  finally:
    install.uninstall()
