"""CPU: host-side logic of the reference-interface mirror (no device compute)."""
from argparse import Namespace

import numpy as np
import pytest

from dragonfly_amd import kernel as K
from dragonfly_amd import parallel
from dragonfly_amd.engine import KernelSpec
from dragonfly_amd import _lib
from dragonfly_amd.option_handler import get_option_specs, load_options
from dragonfly_amd.oper_utils import EuclideanDomain, random_sample


def test_kernel_hyperparams_and_errors():
  k = K.SEKernel(2, 2.0, [0.1, 1.0])
  assert k.hyperparams['scale'] == 2.0 and list(k.hyperparams['dim_bandwidths']) == [0.1, 1.0]
  assert k.is_guaranteed_psd() and k.dim == 2
  with pytest.raises(ValueError):
    K.SEKernel(3, 1.0, [0.1, 1.0])                    # kernel.py:151
  k1 = K.SEKernel(3, 1.0, 0.5)                        # single bandwidth
  assert list(k1.hyperparams['dim_bandwidths']) == [0.5] * 3
  with pytest.raises(ValueError):
    K.MaternKernel(2, 2.0, 1.0, [1.0, 1.0])           # kernel.py:244 nu must be p + 1/2
  m = K.MaternKernel(2, 2.5, 2.1, [0.1, 1.0])
  assert m.p == 2 and m.hyperparams['nu'] == 2.5
  with pytest.raises(ValueError):
    K.AdditiveKernel(1.0, [k], [[0], [1]])            # kernel.py:471
  assert 'SE: ' in str(k) and 'Matern: nu=2.5' in str(m)


def test_empty_inputs_return_empty_matrix():
  """ kernel.py:81-82 -- decided on the host, no device needed """
  k = K.SEKernel(2, 2.0, [0.1, 1.0])
  assert k(np.zeros((0, 2)), np.zeros((5, 2))).shape == (0, 5)
  assert k([], []).shape == (0, 0)


def test_kernel_desc_marshalling():
  d = K.MaternKernel(3, 1.5, 0.7, [0.2, 0.3, 0.4]).to_spec().to_desc()
  assert d.kind == _lib.KERNEL_MATERN and d.dim == 3 and d.nu == 1.5 and d.scale == 0.7
  assert [d.bw[i] for i in range(3)] == [0.2, 0.3, 0.4]
  subs = [K.SEKernel(2, 1.0, [0.5, 0.6]), K.MaternKernel(1, 2.5, 1.0, [0.7])]
  spec = K.AdditiveKernel(3.0, subs, [[2, 0], [1]]).to_spec()
  d = spec.to_desc()
  assert d.kind == _lib.KERNEL_ADDITIVE and d.n_groups == 2 and d.dim == 3 and d.scale == 3.0
  assert [d.group_off[i] for i in range(3)] == [0, 2, 3]
  assert [d.group_dims[i] for i in range(3)] == [2, 0, 1]
  assert [d.sub_kind[i] for i in range(2)] == [_lib.KERNEL_SE, _lib.KERNEL_MATERN]
  assert [d.sub_bw[i] for i in range(3)] == [0.5, 0.6, 0.7]
  d = K.PolyKernel(2, 3, 1.5, [0.5, 0.7]).to_spec().to_desc()        # nu carries the order, bw the scalings
  assert d.kind == _lib.KERNEL_POLY and d.nu == 3.0 and [d.bw[i] for i in range(2)] == [0.5, 0.7]
  prod = K.CoordinateProductKernel(3, 2.0, [K.ExpDecayKernel(1, 1.0, 0.1, [2.0]), K.SEKernel(2, 1.0, [0.5, 0.6])],
                                   [[0], [1, 2]]).to_spec().to_desc()
  assert prod.kind == _lib.KERNEL_PRODUCT and [prod.sub_kind[i] for i in range(2)] == [_lib.KERNEL_EXPDECAY, _lib.KERNEL_SE]
  assert prod.sub_nu[0] == 0.1 and [prod.sub_bw[i] for i in range(3)] == [2.0, 0.5, 0.6]    # offset; power, bandwidths
  with pytest.raises(ValueError):
    KernelSpec('spline', 2, 1.0, [1, 1]).to_desc()


def test_batch_descriptors_equal_the_per_candidate_ones():
  """ engine._single_kind_descs (a tuning batch's struct array filled column-wise) == one to_desc per candidate, field by field """
  from dragonfly_amd import engine as E
  rs = np.random.RandomState(3)
  d = 4
  specs = []
  for i in range(37):
    kind = ['se', 'matern', 'poly', 'expdecay'][i % 4]
    nu = {'se': 0.0, 'matern': [0.5, 1.5, 2.5][i % 3], 'poly': 2.0, 'expdecay': 0.3}[kind]
    specs.append(KernelSpec(kind, d, 0.5 + rs.rand(), 0.2 + rs.rand(d), nu=nu))
  keep = []
  arr = E._single_kind_descs(specs, d, keep)      # pylint: disable=protected-access
  assert arr is not None and len(keep) == 2
  for i, sp in enumerate(specs):
    one = sp.to_desc()
    got = arr[i]
    assert (got.kind, got.dim, got.scale, got.nu, got.n_groups) == (one.kind, one.dim, one.scale, one.nu, 0)
    assert [got.bw[k] for k in range(d)] == [one.bw[k] for k in range(d)]
    assert not got.group_off and not got.sub_bw and not got.factor_scale
  # anything with groups, or a bandwidth vector of another length, takes the per-candidate path (which reports it)
  add = K.AdditiveKernel(3.0, [K.SEKernel(2, 1.0, [0.5, 0.6]), K.MaternKernel(2, 2.5, 1.0, [0.7, 0.8])], [[2, 0], [1, 3]]).to_spec()
  assert E._single_kind_descs(specs[:3] + [add], d, []) is None      # pylint: disable=protected-access
  assert E._single_kind_descs([KernelSpec('se', d, 1.0, [0.1, 0.2])], d, []) is None      # pylint: disable=protected-access
  assert E._single_kind_descs([], d, []) is None      # pylint: disable=protected-access


def test_option_handler():
  specs = [get_option_specs('a', False, 1, ''), get_option_specs('b', False, 'x', '')]
  o = load_options(specs)
  assert o.a == 1 and o.b == 'x'
  o = load_options(specs, partial_options=Namespace(a=5))
  assert o.a == 5 and o.b == 'x'
  o = load_options(specs, partial_options={'b': 'y'})
  assert o.b == 'y'


def test_random_sample_draw_order_matches_reference():
  """ oper_utils.py:59-67: candidates = map_to_bounds(np.random.random((m, d)), bounds) """
  bounds = np.array([[-5.0, 10.0], [0.0, 15.0]])
  np.random.seed(7)
  pts, vals = random_sample(lambda x: x.sum(axis=1), bounds, 11)
  np.random.seed(7)
  ref = np.random.random((11, 2)) * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0]
  assert np.array_equal(pts, ref) and np.array_equal(vals, ref.sum(axis=1))


def test_shard_bounds_cover_and_align():
  for m, w, align in ((10, 3, 1), (2097152, 8, 4096), (1000, 8, 4096), (5, 8, 1), (12289, 4, 4096)):
    spans = [parallel.shard_bounds(m, r, w, align) for r in range(w)]
    assert spans[0][0] == 0 and spans[-1][1] == m
    for (lo, hi), (lo2, _) in zip(spans[:-1], spans[1:]):
      assert hi == lo2 and lo <= hi
    for lo, hi in spans:
      assert lo % align == 0 or lo == m


def test_reduce_argmax_numpy_semantics():
  rs = np.random.RandomState(0)
  for trial in range(200):
    vals = rs.randint(0, 4, size=9).astype(float)
    if trial % 3 == 0:
      vals[rs.randint(0, 9, size=2)] = np.nan
    # split into 3 shards, each reports its own np.argmax
    shard_v, shard_i = [], []
    for lo, hi in ((0, 3), (3, 3), (3, 9)):          # middle shard empty
      if hi > lo:
        j = int(np.argmax(vals[lo:hi]))
        shard_v.append(vals[lo + j]); shard_i.append(lo + j)
      else:
        shard_v.append(float('nan')); shard_i.append(-1)
    v, i = parallel.reduce_argmax(shard_v, shard_i)
    assert i == int(np.argmax(vals))
    assert (v != v) if np.isnan(vals[i]) else v == vals[i]


def test_mt19937_jump_ahead_lands_in_numpys_state():
  """ dfh_mt19937_advance (host code, polynomial jump-ahead) against NumPy walking the stream:
      same key, same position, same draws afterwards -- inside a block, across one block edge,
      across many, and 2^26 words away """
  from dragonfly_amd.parallel import advance_mt19937
  for seed, before, n in ((1, 0, 1), (2, 5, 311), (3, 0, 312), (4, 7, 313), (5, 100, 1872), (6, 3, 100001),
                          (7, 11, (1 << 23) + 5), (8, 0, 1 << 25)):
    walked, jumped = np.random.RandomState(seed), np.random.RandomState(seed)
    walked.random_sample(before)
    jumped.random_sample(before)
    left = n
    while left > 0:
      walked.random_sample(min(left, 1 << 22))
      left -= min(left, 1 << 22)
    advance_mt19937(jumped, n)
    a, b = walked.get_state(), jumped.get_state()
    assert np.array_equal(a[1], b[1]) and a[2] == b[2], (seed, n)
    assert np.array_equal(walked.random_sample(7), jumped.random_sample(7))
    assert walked.standard_normal() == jumped.standard_normal()


def test_gp_kernel_modes_and_bad_lengths():
  from dragonfly_amd.gp_core import GP
  class Foreign(object):                     # a kernel the host evaluates (no device description)
    def is_guaranteed_psd(self):
      return True
  class NotPsd(Foreign):
    def is_guaranteed_psd(self):
      return False
  g = GP([np.zeros(2)], [0.0], Foreign(), lambda x: np.zeros(len(x)), 0.1, build_posterior=False)
  assert g._generic
  g2 = GP([np.zeros(2)], [0.0], K.SEKernel(2, 1.0, [1, 1]), lambda x: np.zeros(len(x)), 0.1, build_posterior=False)
  assert not g2._generic
  with pytest.raises(AssertionError):        # the reference's check (gp_core.py:116-118)
    GP([np.zeros(2)], [0.0], NotPsd(), lambda x: np.zeros(len(x)), 0.1, build_posterior=False)
  g3 = GP([np.zeros(2)], [0.0], NotPsd(), lambda x: np.zeros(len(x)), 0.1, build_posterior=False,
          handle_non_psd_kernels='project_first')
  assert g3._generic
  with pytest.raises(ValueError):
    GP([np.zeros(2)], [0.0], NotPsd(), lambda x: np.zeros(len(x)), 0.1, build_posterior=False,
       handle_non_psd_kernels='something_else')
  with pytest.raises(ValueError):
    GP([np.zeros(2)], [0.0, 1.0], K.SEKernel(2, 1.0, [1, 1]), lambda x: np.zeros(len(x)), 0.1,
       build_posterior=False)


def test_fitter_set_up_matches_reference_bounds():
  """ gp_core.py:393-416, euclidean_gp.py:254-268: hyper-parameter boxes """
  from dragonfly_amd.euclidean_gp import EuclideanGPFitter
  rs = np.random.RandomState(1)
  X = rs.rand(20, 3)
  Y = rs.randn(20)
  f = EuclideanGPFitter(list(X), list(Y), options=Namespace(kernel_type='se', ml_hp_tune_opt='rand'))
  Yvar = Y.std() ** 2 + 0.0001
  assert f.param_order[0] == ['noise_mean', 'cts'] and f.param_order[1] == ['noise_var', 'cts']
  assert np.allclose(f.cts_hp_bounds[1], [np.log(0.005 * Yvar), np.log(0.2 * Yvar)])
  assert np.allclose(f.cts_hp_bounds[2], [np.log(0.1 * Yvar), np.log(10 * Yvar)])
  xn = np.linalg.norm(X, 'fro') + 1e-4
  assert np.allclose(f.cts_hp_bounds[3], [np.log(0.01 * xn), np.log(10 * xn)])
  assert f.num_hps == 2 + 1 + 3 and f.hp_tune_max_evals == min(1e4, max(500, 6 * 200))
  # gp_core.py:77-82, 455-457: 'default' is direct up to 60 hyper-parameters (served by the PDOO
  # fall-back, oper_utils.py:130-133), with its own evaluation budget
  g = EuclideanGPFitter(list(X), list(Y), options=Namespace(kernel_type='se'))
  assert g.ml_hp_tune_opt_method == 'direct' and g.hp_tune_max_evals == min(1e4, max(500, 6 * 50))
  with pytest.raises(ValueError):
    EuclideanGPFitter(list(X), list(Y), options=Namespace(ml_hp_tune_opt='anneal'))
  with pytest.raises(NotImplementedError):      # euclidean_gp.py:280-282: 'Not implemented Poly kernel yet.'
    EuclideanGPFitter(list(X), list(Y), options=Namespace(kernel_type='poly'))
  with pytest.raises(ValueError):               # euclidean_gp.py:222-223
    EuclideanGPFitter(list(X), list(Y), options=Namespace(kernel_type='expdecay'))


def test_domain_stub():
  dom = EuclideanDomain([[0, 1], [2, 3]])
  assert dom.get_type() == 'euclidean' and dom.get_dim() == 2 and dom.is_a_member([0.5, 2.5])


def test_install_rebinds_reference_names():
  """ dragonfly_amd.install (SURVEY.md section 8b seams S1-S4); needs the reference importable,
      which is only the case in the build container. """
  import os
  import sys
  ref = os.environ.get('DRAGONFLY_REFERENCE', '/root/reference')
  if not os.path.isdir(os.path.join(ref, 'dragonfly')):
    pytest.skip('reference not present')
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
  import make_golden
  make_golden.import_reference()
  import dragonfly.gp.kernel as ref_kernel
  import dragonfly.gp.euclidean_gp as ref_egp
  import dragonfly.opt.gpb_acquisitions as ref_acq
  from dragonfly_amd import install, euclidean_gp, gpb_acquisitions
  orig_se, orig_gp, orig_ucb = ref_kernel.SEKernel, ref_egp.EuclideanGP, ref_acq.asy.ucb
  orig_mfgp = ref_egp.EuclideanMFGP
  patched = install.install()
  try:
    assert ref_egp.EuclideanMFGP is orig_mfgp              # multi-fidelity rebinding is opt-in
    assert ref_kernel.SEKernel is K.SEKernel and ref_kernel.AdditiveKernel is K.AdditiveKernel
    assert ref_egp.EuclideanGP is euclidean_gp.EuclideanGP
    assert ref_acq.asy.ucb.__wrapped__ is gpb_acquisitions.asy_ucb and ref_acq.syn.ts.__wrapped__ is gpb_acquisitions.syn_ts
    assert ref_acq.asy.ucb.reference_callable is orig_ucb      # non-Euclidean domains keep the reference's
    assert gpb_acquisitions.external_maximise_with_method is not None and len(patched) >= 20
    # the reference's kernel factory now builds device kernels
    kern, _, _ = ref_egp.get_euclidean_integral_gp_kernel_with_scale(
        'se', 2.0, {'dim': 3}, np.log([0.3, 0.4, 0.5]), [], False, [[0, 2], [1]])
    assert isinstance(kern, K.AdditiveKernel) and isinstance(kern.kernel_list[0], K.SEKernel)
  finally:
    install.uninstall()
  assert ref_kernel.SEKernel is orig_se and ref_egp.EuclideanGP is orig_gp and ref_acq.asy.ucb is orig_ucb
  assert ref_egp.EuclideanMFGP is orig_mfgp
  from dragonfly_amd import mf_gp
  install.install(multi_fidelity=True)
  try:
    assert ref_egp.EuclideanMFGP is mf_gp.EuclideanMFGP
  finally:
    install.uninstall()
  assert ref_egp.EuclideanMFGP is orig_mfgp


def test_reference_se_kernel_helper_known_answers(monkeypatch):
  """ gp/unittest_kernel.py:153-209: effective norms of the SE kernel (known answers) and the
      std-slack bounds, with the kernel matrices coming from the engine """
  from oracle_engine import patch_engine
  patch_engine(monkeypatch)
  data_1, data_2 = np.array([1, 2]), np.array([[0, 1, 2], [1, 1, 0.5]])
  for data, bws, dim, l2, l1 in ((data_1, [0.1, 1], 2, np.sqrt(104), 12),
                                 (data_2, [0.5, 1, 2], 3, np.array([np.sqrt(2), np.sqrt(5.0625)]), np.array([2, 3.25]))):
    kern = K.SEKernel(dim, 1, bws)
    single = len(data.shape) == 1
    assert np.linalg.norm(kern.get_effective_norm(data, order=2, is_single=single) - l2) < 1e-5
    assert np.linalg.norm(kern.get_effective_norm(data, order=1, is_single=single) - l1) < 1e-5
  def post_std(kern, X_tr, X_te):
    K_tr, K_tetr, K_te = kern.evaluate(X_tr, X_tr), kern.evaluate(X_te, X_tr), kern.evaluate(X_te, X_te)
    return np.sqrt(np.diag(K_te - K_tetr.dot(np.linalg.solve(K_tr, K_tetr.T))))
  rs = np.random.RandomState(11)
  for dim, scale, num in ([2, 1, 10], [3, 2, 0], [10, 6, 13]):
    kern = K.SEKernel(dim, scale, list(rs.random_sample(dim) * 0.3 + 0.5))
    X_1, X_2, X_tr = rs.random_sample((5, dim)), rs.random_sample((5, dim)), rs.random_sample((num, dim))
    std_diff = np.abs(post_std(kern, X_tr, X_1) - post_std(kern, X_tr, X_2))
    std_slack = kern.compute_std_slack(X_1, X_2)
    assert np.all(std_diff <= std_slack)
    assert np.all(std_slack <= kern.hyperparams['scale'] * kern.get_effective_norm(X_1 - X_2, order=2, is_single=False))
  kern = K.SEKernel(2, 1.0, [0.5, 2.0])
  kern.change_smoothness(2.0)
  assert np.array_equal(kern.hyperparams['dim_bandwidths'], [1.0, 4.0])


def test_rccl_id_rendezvous_file_is_private_and_fresh(tmp_path, monkeypatch):
  """ dragonfly_amd/parallel.py: the unique id travels through a 0600 file in a directory only this user
      can write; rank 0 replaces whatever a crashed run left; readers ignore files with foreign modes """
  import os
  import stat
  import threading
  from dragonfly_amd import parallel
  d = tmp_path / 'rdzv'
  d.mkdir(mode=0o700)
  monkeypatch.setenv('DFH_RDZV_DIR', str(d))
  monkeypatch.setenv('TORCHELASTIC_RESTART_COUNT', '3')
  path = parallel._rendezvous_path()                      # pylint: disable=protected-access
  assert os.path.dirname(path) == str(d) and '_3_' in os.path.basename(path)
  with open(path, 'wb') as f:                             # a stale, world-readable leftover of the right size
    f.write(b'x' * 128)
  os.chmod(path, 0o644)
  got = {}
  t = threading.Thread(target=lambda: got.update(r=parallel.exchange_unique_id(1, None, timeout=20.0, nbytes=128)))
  t.start()
  blob, p0 = parallel.exchange_unique_id(0, lambda: bytes(range(128)), nbytes=128)
  t.join()
  assert p0 == path and got['r'][0] == blob == bytes(range(128))          # the reader waited for the fresh, private file
  assert stat.S_IMODE(os.stat(path).st_mode) == 0o600
  os.chmod(str(d), 0o777)
  with pytest.raises(RuntimeError):
    parallel._rendezvous_path()                           # pylint: disable=protected-access


def test_gaplog_margins(monkeypatch):
  """ dragonfly_amd/gaplog.py (round 6): the relative margin of an arg-max / of a comparison, NaNs out of competition,
      the summary's counts -- the instrument behind profiles/r06_argmax_gaps.json. """
  from dragonfly_amd import gaplog
  monkeypatch.setattr(gaplog, 'ENABLED', True)
  gaplog.reset()
  gaplog.top2('a', [1.0, 3.0, np.nan, 2.0, -np.inf])          # (3 - 2) / 3
  gaplog.top2('a', [5.0, 5.0])                                # an exact tie
  gaplog.top2('a', [7.0])                                     # nothing to compare with
  gaplog.pair('b', -2.0, -2.0 * (1 + 1e-13))
  gaplog.pair('b', np.nan, 1.0)
  s = gaplog.summary()
  assert s['a']['count'] == 2 and s['a']['min'] == 0.0 and s['a']['exact_ties'] == 1 and s['a']['below_1e-8'] == 1
  assert s['b']['count'] == 1 and 0.5e-13 < s['b']['min'] < 2e-13 and s['b']['below_1e-12'] == 1 and s['b']['below_1e-10'] == 1
  gaplog.reset()
  monkeypatch.setattr(gaplog, 'ENABLED', False)
  gaplog.top2('a', [1.0, 2.0])
  assert gaplog.summary() == {}
