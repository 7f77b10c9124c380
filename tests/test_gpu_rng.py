"""MI355X: device candidate generation is bit-identical to the draw the reference
makes on the host -- np.random.random((m, d)) from the global MT19937 state mapped to the bounds
(dragonfly/utils/oper_utils.py:62, general_utils.py:25-27) -- and leaves the generator in the state
NumPy would have left it in.  NumPy itself is the reference's arithmetic here, so it is the
checker; oracle/ref_rng.py is checked against it on the CPU side."""
import numpy as np
import pytest

from oracle import ref_rng as R

pytestmark = pytest.mark.gpu

BOX = np.array([[-5.0, 10.0], [0.0, 15.0], [0.1, 0.3], [2.0, 2.5], [-1.0, 1.0]])


@pytest.mark.parametrize('seed,burn,m,d', [(1, 0, 1000, 2), (2, 5, 313, 5), (3, 623, 1, 1), (4, 1248, 4096, 3),
                                           (5, 11, 0, 4), (6, 1, 65536, 32)])
def test_mt19937_candidates_match_numpy_and_continue_the_stream(engine, seed, burn, m, d):
  ref = np.random.RandomState(seed)
  dev = np.random.RandomState(seed)
  if burn:
    ref.random_sample(burn)
    dev.random_sample(burn)
  bounds = None if d > len(BOX) else BOX[:d]
  want = ref.random_sample((m, d))
  if bounds is not None:
    want = R.map_to_bounds(want, bounds)
  got = engine.random_candidates(m, d, bounds=bounds, rng=dev)
  assert got.shape == (m, d)
  if m:
    assert np.array_equal(got.download(), want)
  a, b = ref.get_state(), dev.get_state()
  assert np.array_equal(a[1], b[1]) and a[2] == b[2]
  # ... and later host draws (the TS normals, the next hyper-parameter samples) continue identically
  assert np.array_equal(ref.normal(size=7), dev.normal(size=7))


def test_mt19937_global_state_and_host_output(engine):
  np.random.seed(4242)
  want = R.map_to_bounds(np.random.random((777, 3)), BOX[:3])
  tail = np.random.random(5)
  np.random.seed(4242)
  out = np.empty((777, 3))
  engine.random_candidates(777, 3, bounds=BOX[:3], out=out)      # rng=None: the global state
  assert np.array_equal(out, want)
  assert np.array_equal(np.random.random(5), tail)


def test_mt19937_against_the_restatement_across_many_blocks(engine):
  rs = np.random.RandomState(99)
  rs.random_sample(100)
  st = rs.get_state()
  want, key, pos = R.mt19937_random_sample(st[1], st[2], (3000, 7))
  got = engine.random_candidates(3000, 7, rng=rs).download()
  after = rs.get_state()
  assert np.array_equal(got, want) and np.array_equal(after[1], key) and after[2] == pos


@pytest.mark.parametrize('burn,m,d', [(0, 1000, 2), (3, 101, 4), (1, 1, 1), (2, 1, 2), (4, 50000, 5), (0, 0, 3)])
def test_philox_candidates_match_numpy_generator(engine, burn, m, d):
  ones = (1 << 64) - 1
  mk = lambda: np.random.Generator(np.random.Philox(key=np.array([7, ones], dtype=np.uint64),
                                                    counter=np.array([ones - 2, ones, 3, 0], dtype=np.uint64)))
  ref, dev = mk(), mk()
  if burn:
    ref.random(burn)
    dev.random(burn)
  want = R.map_to_bounds(ref.random((m, d)), BOX[:d])
  got = engine.random_candidates(m, d, bounds=BOX[:d], rng=dev)
  if m:
    assert np.array_equal(got.download(), want)
  sa, sb = ref.bit_generator.state, dev.bit_generator.state
  assert np.array_equal(sa['state']['counter'], sb['state']['counter'])
  assert sa['buffer_pos'] == sb['buffer_pos']
  assert np.array_equal(ref.random(9), dev.random(9))


def test_unsupported_generator_is_rejected(engine):
  with pytest.raises(ValueError):
    engine.random_candidates(10, 2, rng=np.random.Generator(np.random.PCG64(1)))


@pytest.mark.parametrize('mean_kind', ['constant', 'callable'])
def test_acquisitions_pick_the_same_point_with_device_and_host_candidates(engine, mean_kind, monkeypatch):
  from argparse import Namespace
  from dragonfly_amd import gpb_acquisitions as A
  from dragonfly_amd.euclidean_gp import EuclideanGP
  from dragonfly_amd.gp_core import ConstantMean
  from dragonfly_amd.kernel import SEKernel, AdditiveKernel
  from dragonfly_amd.oper_utils import EuclideanDomain
  rs = np.random.RandomState(8)
  d = 4
  X = rs.random_sample((60, d))
  Y = np.sin(3 * X[:, 0]) + X[:, 1] ** 2 - X[:, 2] * X[:, 3]
  mean = ConstantMean(float(np.median(Y))) if mean_kind == 'constant' else (lambda x: 0.1 * np.asarray(x)[:, 0])
  gp = EuclideanGP(X, Y, SEKernel(d, float(Y.var()), 0.4 * np.ones(d)), mean, float(Y.var()) / 20)
  add_kernel = AdditiveKernel(float(Y.var()), [SEKernel(2, 1.0, 0.4 * np.ones(2)) for _ in range(2)],
                              [[0, 2], [3, 1]])
  add_gp = EuclideanGP(X, Y, add_kernel, ConstantMean(float(np.median(Y))), float(Y.var()) / 20)
  domain = EuclideanDomain([[-1, 2], [0, 1], [0.5, 0.75], [0, 3]])
  mk_anc = lambda: Namespace(max_evals=3000, t=60, domain=domain, domain_bounds=domain.bounds,
                             acq_opt_method='rand', curr_max_val=float(Y.max()), handle_parallel='halluc',
                             eval_points_in_progress=[np.array([0.3, 0.3, 0.6, 1.0])], is_mf=False)
  for acq, model in (('ucb', gp), ('ei', gp), ('pi', gp), ('ttei', gp), ('ts', gp), ('add_ucb', add_gp)):
    picks, tails = [], []
    for on_device in (True, False):
      monkeypatch.setattr(A, 'DEVICE_CANDIDATES', on_device)
      np.random.seed(31)
      anc = mk_anc()
      if acq == 'ts':
        anc.eval_points_in_progress = []
      picks.append(np.asarray(getattr(A.asy, acq)(model, anc), dtype=float))
      tails.append(np.random.random(4))
    assert np.array_equal(picks[0], picks[1]), acq
    assert np.array_equal(tails[0], tails[1]), acq


@pytest.mark.parametrize('kind', ['mt19937', 'philox'])
def test_row_shards_tile_the_full_draw_and_leave_the_same_state(engine, kind):
  """ Multi-GPU candidate generation: every rank starts from the same generator state, keeps its
      own rows of the m x d block and ends in the state of the full draw. """
  m, d = 5000, 7
  def fresh():
    if kind == 'mt19937':
      rs = np.random.RandomState(77)
      rs.random_sample(3)
      return rs
    gen = np.random.Generator(np.random.Philox(key=np.array([3, 4], dtype=np.uint64)))
    gen.random(2)
    return gen
  state_of = lambda g: (g.get_state()[1].tolist(), g.get_state()[2]) if kind == 'mt19937' else \
                       (g.bit_generator.state['state']['counter'].tolist(), g.bit_generator.state['buffer_pos'])
  ref = fresh()
  full = engine.random_candidates(m, d, bounds=np.tile(BOX, (2, 1))[:d], rng=ref).download()
  for world in (1, 3, 8):
    from dragonfly_amd.parallel import shard_bounds
    parts = []
    for rank in range(world):
      lo, hi = shard_bounds(m, rank, world, align=64)
      g = fresh()
      part = engine.random_candidates(m, d, bounds=np.tile(BOX, (2, 1))[:d], rng=g, rows=(lo, hi - lo))
      assert part.shape == (hi - lo, d)
      parts.append(part.download() if hi > lo else np.zeros((0, d)))
      assert state_of(g) == state_of(ref)
    assert np.array_equal(np.concatenate(parts), full)
  with pytest.raises(ValueError):
    engine.random_candidates(10, 2, rows=(5, 6))


def test_large_mt19937_shards_jump_over_the_rest_of_the_stream(engine):
  """ Shards with more than 2^23 words before or after them do not walk those words: the state jumps
      (csrc/mtjump.hip).  Rows and the state handed back must still be the full draw's
      (np.random.random((m, d)), oper_utils.py:62); a rank with no rows only moves the state. """
  from dragonfly_amd.parallel import shard_bounds
  m, d = 1 << 19, 32                       # 2^25 words; a quarter is 2^23
  ref = np.random.RandomState(77)
  ref.random_sample(13)
  want = ref.random_sample((m, d))
  want_state = ref.get_state()
  for world, ranks in ((2, (0, 1)), (4, (0, 2, 3))):
    for rank in ranks:
      lo, hi = shard_bounds(m, rank, world, align=64)
      g = np.random.RandomState(77)
      g.random_sample(13)
      part = engine.random_candidates(m, d, rng=g, rows=(lo, hi - lo)).download()
      assert np.array_equal(part, want[lo:hi])
      st = g.get_state()
      assert np.array_equal(st[1], want_state[1]) and st[2] == want_state[2]
  g = np.random.RandomState(77)
  g.random_sample(13)
  engine.random_candidates(m, d, rng=g, rows=(m // 2, 0))
  assert np.array_equal(g.get_state()[1], want_state[1]) and g.get_state()[2] == want_state[2]


def test_c_abi_argument_checks(engine):
  """ bad generator positions / row windows / dimensions are DFH_ERR_BAD_ARG (-> ValueError), not
      memory errors """
  import ctypes as C
  from dragonfly_amd._lib import check
  key = np.zeros(624, dtype=np.uint32)
  out = np.empty(8)
  ptr = lambda a: a.ctypes.data_as(C.c_void_p)
  for pos, m, d, r0, rc in ((625, 4, 2, 0, 4), (-1, 4, 2, 0, 4), (0, 4, 0, 0, 4), (0, 4, 2, 3, 2), (0, 4, 2, -1, 1)):
    p = C.c_int32(pos)
    with pytest.raises(ValueError):
      check(engine.lib.dfh_rand_mt19937_uniform(engine.ctx, ptr(key), C.byref(p), m, d, r0, rc, None, ptr(out)))
  pk = np.zeros(2, dtype=np.uint64); ctr = np.zeros(4, dtype=np.uint64); held = np.zeros(4, dtype=np.uint64)
  bp = C.c_int32(5)
  with pytest.raises(ValueError):
    check(engine.lib.dfh_rand_philox_uniform(engine.ctx, ptr(pk), ptr(ctr), ptr(held), C.byref(bp), 4, 2, 0, 4, None, ptr(out)))


# ---- np.random.normal on the device (dfh_rand_mt19937_normal) ----------------------------------
@pytest.mark.parametrize('m', [1, 2, 3, 10, 1001, 4096, 300000])
def test_device_normals_equal_numpy_word_for_word(engine, m):
  """ draw_gaussian_samples' np.random.normal(size=(m, 1)) (general_utils.py:230): the device
      stream equals NumPy's bit for bit and leaves the generator -- cached second gaussian
      included -- where NumPy leaves it, so later host draws continue identically """
  for seed, warm in ((5, 0), (6, 1), (7, 3)):
    ref = np.random.RandomState(seed)
    dev = np.random.RandomState(seed)
    for rs in (ref, dev):
      rs.random_sample(11)
      if warm:
        rs.normal(size=warm)          # odd counts leave a cached gaussian behind
    want = ref.normal(size=(m, 1)).ravel()
    got = engine.random_normals(m, rng=dev).download()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), \
        (m, seed, int(np.sum(got != want)), float(np.max(np.abs(got - want))))
    sr, sd = ref.get_state(), dev.get_state()
    assert np.array_equal(sr[1], sd[1]) and sr[2:] == sd[2:]
    assert np.array_equal(ref.normal(size=5), dev.normal(size=5))
    assert np.array_equal(ref.random_sample(7), dev.random_sample(7))


def test_device_normals_two_million_match_numpy(engine):
  """ the size of BASELINE config 4's draw: 2 097 152 normals, every one equal to NumPy's """
  m = 2097152
  ref, dev = np.random.RandomState(304), np.random.RandomState(304)
  want = ref.standard_normal(m)
  got = engine.random_normals(m, rng=dev).download()
  assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), int(np.sum(got != want))
  assert np.array_equal(ref.get_state()[1], dev.get_state()[1]) and ref.get_state()[2:] == dev.get_state()[2:]


def test_device_normals_global_state_and_host_output(engine):
  np.random.seed(99)
  want = np.random.normal(size=777)
  tail = np.random.random(3)
  np.random.seed(99)
  out = np.empty(777)
  engine.random_normals(777, out=out)
  assert np.array_equal(out, want) and np.array_equal(np.random.random(3), tail)
