"""CPU: the random-stream restatements of oracle/ref_rng.py against NumPy's own generators (the
reference draws its candidates with np.random.random, dragonfly/utils/oper_utils.py:62) and the
Random123 known-answer vectors of Philox4x64-10."""
import numpy as np

from oracle import ref_rng as R


def test_mt19937_restatement_matches_numpy_legacy_stream():
  for seed, burn, shape in ((1234, 7, (500, 3)), (5, 0, (313, 2)), (77, 623, (1, 1)), (9, 1248, (2000, 5))):
    rs = np.random.RandomState(seed)
    if burn:
      rs.random_sample(burn)
    state = rs.get_state()
    want = rs.random_sample(shape)
    got, key, pos = R.mt19937_random_sample(state[1], state[2], shape)
    after = rs.get_state()
    assert np.array_equal(got, want)
    assert np.array_equal(key, after[1]) and pos == after[2]


def test_philox_known_answers():
  # Random123 kat_vectors, philox4x64 10 rounds
  assert R.philox4x64_10([0] * 4, [0] * 2) == [0x16554d9eca36314c, 0xdb20fe9d672d0fdc,
                                              0xd7e772cee186176b, 0x7e68b68aec7ba23b]
  ones = (1 << 64) - 1
  assert R.philox4x64_10([ones] * 4, [ones] * 2) == [0x87b092c3013fe90b, 0x438c3c67be8d0224,
                                                    0x9cc7d7c69cd777b6, 0xa09caebf594f0ba0]
  assert R.philox4x64_10([0x243f6a8885a308d3, 0x13198a2e03707344, 0xa4093822299f31d0, 0x082efa98ec4e6c89],
                         [0x452821e638d01377, 0xbe5466cf34e90c6c]) == \
      [0xa528f45403e61d95, 0x38c72dbd566e9788, 0xa5a1610e72fd18b5, 0x57bd43b5e52b7fe6]


def test_philox_restatement_matches_numpy_generator():
  ones = (1 << 64) - 1
  for key, counter, burn, shape in (([11, 22], [5, 0, 0, 0], 3, (101, 4)), ([0, 0], [0, 0, 0, 0], 0, (64, 2)),
                                    ([7, ones], [ones, ones, 3, 0], 1, (33, 3)), ([1, 2], [9, 9, 9, 9], 4, (1, 1))):
    bit_gen = np.random.Philox(key=np.array(key, dtype=np.uint64), counter=np.array(counter, dtype=np.uint64))
    gen = np.random.Generator(bit_gen)
    if burn:
      gen.random(burn)
    st = bit_gen.state
    want = gen.random(shape)
    got, ctr, held, held_pos = R.philox_random(st['state']['key'], st['state']['counter'], st['buffer'],
                                               st['buffer_pos'], shape)
    after = bit_gen.state
    assert np.array_equal(got, want)
    assert ctr == [int(c) for c in after['state']['counter']]
    assert held == [int(b) for b in after['buffer']] and held_pos == after['buffer_pos']


def test_map_to_bounds_is_two_rounded_operations():
  pts = np.random.RandomState(3).random_sample((50, 3))
  bounds = np.array([[-5.0, 10.0], [0.0, 15.0], [0.1, 0.3]])
  want = pts * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0]
  assert np.array_equal(R.map_to_bounds(pts, bounds), want)
