"""Univariate slice sampling with speculative, batched density evaluations.

The reference samples hyper-parameters from the posterior (hp_tune_criterion 'post_sampling',
dragonfly/gp/gp_core.py:592-726) one coordinate at a time with the slice sampler of
dragonfly/sampling/slice.py:38-89.  Every step of that sampler -- the stepping-out of the slice
edges and the shrinking towards an accepted point -- is a `while` loop that calls the log density
once per turn, and every call is a GP fit: this is where an ordinary Dragonfly run spends most of
its time.

The loops only *look* sequential.  The points the stepping-out loop visits are q_l - k w and
q_r + k w, known in advance; the points the shrinking loop visits depend on the density only
through "all earlier ones were rejected", and on the random stream, which can be read ahead and
rewound.  So each loop's next few candidates are evaluated in ONE batched density call (one
dfh_gp_lml_batch on the device), the first candidate that ends the loop is found, and exactly the
random numbers the reference would have consumed are consumed.  The chain is the reference's, draw
for draw; what changes is two device calls per step instead of six to eight.

And often ONE: the shrinking loop's first candidates depend on the stepping-out only through the edges it ends with,
and with the width tuned to the slices (slice.py:84-87) the edges mostly stay where they start.  The first batch of an
update therefore carries both loops' candidates -- the shrinking ones drawn under the guess that neither edge moves.
If the guess holds, the update may be over with that one call; if not, their draws go back into the stream, their
densities are dropped, and the update goes on as before (the stepping-out values of that batch are used either way).
"""
import ctypes

import numpy as np
import numpy.random as nr

from . import gaplog


class _GlobalStreamPosition(object):
  """ Un-reading the last few draws of the GLOBAL legacy stream without copying its 2.5 KB state.

      nr.get_state() + nr.set_state() cost 80 us together; the shrinking loop below did both on every turn, a fifth of
      a short optimisation run.  The MT19937 state is 624 words and a position; nr.rand() takes two words per double; as
      long as the draws to be un-read came out of the block of words that is current, un-reading them is `pos -= 2 k`.
      The position is reached through the bit generator's documented ctypes interface (state_address: struct
      { uint32_t key[624]; int pos; }).  Whenever that does not apply -- another bit generator behind np.random, a
      block boundary inside the draws -- the caller takes the state-copy route. """

  def __init__(self):
    self._bind()

  def _bind(self):
    self.pos = None
    self._bitgen = None
    try:
      bitgen = nr.mtrand._rand._bit_generator      # pylint: disable=protected-access
      if type(bitgen).__name__ != 'MT19937':
        return
      addr = bitgen.ctypes.state_address
      addr = addr.value if hasattr(addr, 'value') else int(addr)
      self.pos = ctypes.c_int.from_address(addr + 624 * 4)
      self._bitgen = bitgen                          # (keeps the state's memory alive)
    except Exception:      # pylint: disable=broad-except
      self.pos = None

  def room_for(self, doubles):
    """ True if the next `doubles` calls' worth of words all come from the current block (so they can be un-read) """
    # (np.random.set_bit_generator / a re-seeded legacy RandomState puts ANOTHER generator behind np.random: the cached
    #  position would then belong to the old one and `unread` would rewind the wrong state -- bind again when it changed)
    if nr.mtrand._rand._bit_generator is not self._bitgen:      # pylint: disable=protected-access
      self._bind()
    return self.pos is not None and 0 <= self.pos.value and self.pos.value + 2 * doubles <= 624

  def unread(self, doubles):
    self.pos.value -= 2 * doubles


_STREAM = None


def _stream():
  global _STREAM      # pylint: disable=global-statement
  if _STREAM is None:
    _STREAM = _GlobalStreamPosition()
  return _STREAM


class SpeculativeSlice(object):
  """ slice.py:15-36 for a univariate target given as a batch log-density:
      logp_batch([x_0, ..., x_k]) -> [log p(x_0), ..., log p(x_k)].  w, tune as in the reference;
      `ahead_step` / `ahead_shrink`: how many candidates of each loop go into one batch; `merge_first`: the first batch
      of an update holds both loops' candidates (see the module text). """

  def __init__(self, logp_batch, w=1., tune=True, ahead_step=3, ahead_shrink=4, merge_first=True):
    self.logp_batch = logp_batch
    self.merge_first = bool(merge_first)
    self.merged_done = 0      # updates whose stepping-out AND shrinking were settled by their first batch
    self.w = w
    self.tune = tune
    self.n_tunes = 0.
    self.ahead_step, self.ahead_shrink = int(ahead_step), int(ahead_shrink)
    self.batches = 0          # batched density calls made
    self.evaluated = 0        # densities evaluated in them
    self.consumed = 0         # densities the reference's loops would have asked for
    self._known = (None, None)
    self._level = None        # the current slice's level y (for the margin log of dragonfly_amd.gaplog)

  def _logp(self, xs):
    self.batches += 1
    self.evaluated += len(xs)
    vals = np.asarray(self.logp_batch(list(xs)), dtype=np.float64).ravel()
    if gaplog.ENABLED and self._level is not None:      # every value of a batch is compared with the slice's level
      for v in vals:
        gaplog.pair('slice_compare', self._level, v)
    return vals

  def _step_out(self, y, ql, qr, w, need_l=True, need_r=True):
    """ slice.py:52-63: move each edge outwards by w until the density there is below the level. """
    while need_l or need_r:
      lefts, rights = [], []
      edge = ql
      for _ in range(self.ahead_step if need_l else 0):
        lefts.append(edge)
        edge = edge - w               # the reference's repeated `ql[i] -= w[i]`
      edge = qr
      for _ in range(self.ahead_step if need_r else 0):
        rights.append(edge)
        edge = edge + w
      vals = self._logp(lefts + rights)
      lv, rv = vals[:len(lefts)], vals[len(lefts):]
      if need_l:
        stop = next((k for k, v in enumerate(lv) if not y < v), None)
        self.consumed += (stop + 1) if stop is not None else len(lefts)
        if stop is not None:
          ql, need_l = lefts[stop], False
        else:
          ql = lefts[-1] - w
      if need_r:
        # the reference only starts on the right edge once the left one is settled, but the
        # density calls of the two loops do not interact, so their order does not matter
        stop = next((k for k, v in enumerate(rv) if not y < v), None)
        self.consumed += (stop + 1) if stop is not None else len(rights)
        if stop is not None:
          qr, need_r = rights[stop], False
        else:
          qr = rights[-1] + w
    return ql, qr

  def _shrink_candidates(self, q0, ql, qr):
    """ The next ahead_shrink candidates of the shrinking loop, each drawn from the interval its predecessors'
        rejection would leave: (fast, saved state, candidates, the edges each was drawn from, the edges after all). """
    stream = _stream()
    fast = stream.room_for(self.ahead_shrink)
    state = None if fast else nr.get_state()
    draws = nr.rand(self.ahead_shrink)
    cands, edges = [], []
    l, r = ql, qr
    for u in draws:
      q = (r - l) * u + l
      cands.append(q)
      edges.append((l, r))
      if q > q0:
        r = q
      elif q < q0:
        l = q
    return fast, state, cands, edges, (l, r)

  def _give_back(self, fast, state, keep):
    """ Exactly `keep` of the ahead_shrink draws just read are the reference's: the others go back into the stream. """
    if fast:
      _stream().unread(self.ahead_shrink - keep)
    else:
      nr.set_state(state)
      if keep:
        nr.rand(keep)

  def _shrink(self, y, q0, ql, qr):
    """ slice.py:65-76: draw uniformly from [ql, qr]; a rejected draw becomes the new edge on its
        side of q0.  Returns the accepted point, its log density and the final edges. """
    while True:
      fast, state, cands, edges, after = self._shrink_candidates(q0, ql, qr)
      vals = self._logp(cands)
      hit = next((j for j, v in enumerate(vals) if not v < y), None)
      if hit is not None:
        self._give_back(fast, state, hit + 1)
        self.consumed += hit + 1
        l, r = edges[hit]
        return cands[hit], vals[hit], l, r
      self.consumed += len(cands)
      ql, qr = after                  # all rejected: the stream has advanced by ahead_shrink draws

  def _first_batch(self, y, q0, ql, qr, w):
    """ Both loops' first candidates in one density call (see the module text).  Returns what _shrink returns. """
    lefts, rights = [], []
    edge = ql
    for _ in range(self.ahead_step):
      lefts.append(edge)
      edge = edge - w
    edge = qr
    for _ in range(self.ahead_step):
      rights.append(edge)
      edge = edge + w
    fast, state, cands, edges, after = self._shrink_candidates(q0, ql, qr)      # guess: the edges stay
    vals = self._logp(lefts + rights + cands)
    lv, rv, sv = vals[:len(lefts)], vals[len(lefts):len(lefts) + len(rights)], vals[len(lefts) + len(rights):]
    stop_l = next((k for k, v in enumerate(lv) if not y < v), None)
    stop_r = next((k for k, v in enumerate(rv) if not y < v), None)
    self.consumed += (stop_l + 1) if stop_l is not None else len(lefts)
    self.consumed += (stop_r + 1) if stop_r is not None else len(rights)
    if stop_l == 0 and stop_r == 0:
      hit = next((j for j, v in enumerate(sv) if not v < y), None)
      if hit is not None:
        self._give_back(fast, state, hit + 1)
        self.consumed += hit + 1
        self.merged_done += 1
        l, r = edges[hit]
        return cands[hit], sv[hit], l, r
      self.consumed += len(cands)
      return self._shrink(y, q0, after[0], after[1])
    # an edge moves: the shrinking candidates were drawn from the wrong interval -- their draws go back, all of them
    self._give_back(fast, state, 0)
    need_l, need_r = stop_l is None, stop_r is None
    ql = lefts[stop_l] if stop_l is not None else lefts[-1] - w
    qr = rights[stop_r] if stop_r is not None else rights[-1] + w
    ql, qr = self._step_out(y, ql, qr, w, need_l, need_r)
    return self._shrink(y, q0, ql, qr)

  def _sample(self, q0):
    """ One slice-sampling update of the scalar q0 (slice.py:38-89 with len(q0) == 1). """
    w = float(np.resize(self.w, 1)[0])
    known_q, known_lp = self._known
    if known_q is not None and known_q == q0:
      lp0 = known_lp                  # the density of the point accepted in the previous update
      self.consumed += 1
    else:
      self._level = None
      lp0 = self._logp([q0])[0]
      self.consumed += 1
    y = lp0 - nr.standard_exponential()
    self._level = y
    ql = q0 - nr.uniform(0, w)
    qr = q0 + w
    if self.merge_first:
      q, lp, ql, qr = self._first_batch(y, q0, ql, qr, w)
    else:
      ql, qr = self._step_out(y, ql, qr, w)
      q, lp, ql, qr = self._shrink(y, q0, ql, qr)
    if self.tune:
      self.w = w * (self.n_tunes / (self.n_tunes + 1)) + (qr - ql) / (self.n_tunes + 1)
      self.n_tunes += 1
    self._known = (q, lp)
    return q

  def sample(self, q0, num_samples=1, burn=100):
    """ slice.py:91-110: `burn` discarded updates, then num_samples kept ones; [num_samples x 1]. """
    if num_samples is None:
      num_samples = 1
    q = float(np.ravel(q0)[0])
    for _ in range(burn):
      q = self._sample(q)
    samples = np.zeros([num_samples, 1])
    for i in range(num_samples):
      q = self._sample(q)
      samples[i] = q
    return samples
