"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the two random streams the
candidate generator can follow, so the device streams can be checked word for word.

The reference draws its candidates with the *global* NumPy generator,
`np.random.random((max_evals, dim))` (dragonfly/utils/oper_utils.py:62), and maps them to the box
with `pts * (hi - lo) + lo` (dragonfly/utils/general_utils.py:25-27).  The arithmetic therefore
lives in a third-party dependency, NumPy (unpinned in the reference's requirements.txt; 2.2.6
here), whose published algorithms are restated below:

 * MT19937 (Matsumoto & Nishimura 1998) with NumPy's legacy double construction
   (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53  (numpy/random/src/mt19937, `mt19937_next_double`);
 * Philox4x64-10 (Salmon et al. 2011, Random123) as wrapped by numpy.random.Philox: the 256-bit
   counter is incremented BEFORE each block of four words, doubles are (w >> 11) * 2^-53.

Pinning: tests/test_oracle_rng.py checks both restatements against NumPy's own generators (the
reference's dependency, importable wherever the tests run) on seeded states, and Philox against
the Random123 known-answer vectors.
"""
import numpy as np

_U32 = np.uint32
_MASK64 = (1 << 64) - 1


# ---------------------------------------------------------------------------------------------
# MT19937
# ---------------------------------------------------------------------------------------------
def mt19937_twist(key):
  """ The next 624-word state block from the current one (one call of mt19937_gen). """
  s = [int(w) for w in key]
  for k in range(624):
    y = (s[k] & 0x80000000) | (s[(k + 1) % 624] & 0x7fffffff)
    s[k] = s[(k + 397) % 624] ^ (y >> 1) ^ (0x9908b0df if (y & 1) else 0)
  return np.array(s, dtype=np.uint32)


def mt19937_temper(y):
  y = np.asarray(y, dtype=np.uint64)
  y = y ^ (y >> np.uint64(11))
  y = y ^ ((y << np.uint64(7)) & np.uint64(0x9d2c5680))
  y = y ^ ((y << np.uint64(15)) & np.uint64(0xefc60000))
  y = y ^ (y >> np.uint64(18))
  return (y & np.uint64(0xffffffff)).astype(np.uint32)


def mt19937_words(key, pos, count):
  """ `count` tempered 32-bit outputs starting at position `pos` of state block `key`; returns
      (words, key_after, pos_after) exactly as NumPy's generator would leave its state. """
  key = np.array(key, dtype=np.uint32)
  out = np.empty(count, dtype=np.uint32)
  done = 0
  while done < count:
    if pos == 624:
      key, pos = mt19937_twist(key), 0
    take = min(624 - pos, count - done)
    out[done:done + take] = mt19937_temper(key[pos:pos + take])
    done += take
    pos += take
  return out, key, pos


def mt19937_random_sample(key, pos, shape):
  """ np.random.random(shape) of a legacy generator in state (key, pos). """
  count = int(np.prod(shape))
  w, key, pos = mt19937_words(key, pos, 2 * count)
  a = (w[0::2] >> _U32(5)).astype(np.float64)
  b = (w[1::2] >> _U32(6)).astype(np.float64)
  return ((a * 67108864.0 + b) / 9007199254740992.0).reshape(shape), key, pos


# ---------------------------------------------------------------------------------------------
# Philox4x64-10
# ---------------------------------------------------------------------------------------------
_PH_M0, _PH_M1 = 0xD2E7470EE14C6C93, 0xCA5A826395121157
_PH_W0, _PH_W1 = 0x9E3779B97F4A7C15, 0xBB67AE8584CAA73B


def philox4x64_10(counter, key):
  """ One Philox4x64 block, ten rounds: four 64-bit words from a 256-bit counter and 128-bit key. """
  c = [int(v) & _MASK64 for v in counter]
  k = [int(v) & _MASK64 for v in key]
  for rnd in range(10):
    if rnd:
      k = [(k[0] + _PH_W0) & _MASK64, (k[1] + _PH_W1) & _MASK64]
    p0, p1 = _PH_M0 * c[0], _PH_M1 * c[2]
    c = [(p1 >> 64) ^ c[1] ^ k[0], p1 & _MASK64, (p0 >> 64) ^ c[3] ^ k[1], p0 & _MASK64]
  return c


def philox_counter_add(counter, inc):
  """ 256-bit little-endian-word counter + inc. """
  v = sum(int(c) << (64 * i) for i, c in enumerate(counter)) + int(inc)
  return [(v >> (64 * i)) & _MASK64 for i in range(4)]


def philox_random(key, counter, buffer, buffer_pos, shape):
  """ Generator(Philox).random(shape) from the state (key, counter, buffer, buffer_pos); returns
      (array, counter_after, buffer_after, buffer_pos_after). """
  count = int(np.prod(shape))
  counter = [int(c) for c in counter]
  buffer = [int(b) for b in buffer]
  words = []
  for _ in range(count):
    if buffer_pos >= 4:
      counter = philox_counter_add(counter, 1)
      buffer = philox4x64_10(counter, key)
      buffer_pos = 0
    words.append(buffer[buffer_pos])
    buffer_pos += 1
  w = np.array(words, dtype=np.uint64)
  vals = (w >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
  return vals.reshape(shape), counter, buffer, buffer_pos


def map_to_bounds(pts, bounds):
  """ dragonfly/utils/general_utils.py:25-27: one multiply and one add, each rounded. """
  bounds = np.asarray(bounds, dtype=np.float64)
  return pts * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0]
