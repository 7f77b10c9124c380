"""The speculative slice sampler (dragonfly_amd/slice_sampler.py) against the REAL reference's
sampler (tests/golden/slice_cases.npz from oracle/make_golden.py: dragonfly/sampling/slice.py driven
through distributions/model.py): the same chain sample for sample, the same number of random
numbers consumed -- with the density evaluated a few candidates per batched call."""
import numpy as np
import pytest

from conftest import load_golden
from oracle.test_objectives import SLICE_CASES


@pytest.mark.parametrize('name,logp,start,num,burn,seed', SLICE_CASES, ids=[c[0] for c in SLICE_CASES])
def test_chain_is_the_reference_chain(name, logp, start, num, burn, seed):
  from dragonfly_amd.slice_sampler import SpeculativeSlice
  g = load_golden('slice_cases')
  for ahead_step, ahead_shrink in ((3, 4), (1, 1), (6, 9)):
    sampler = SpeculativeSlice(lambda xs: [logp(x) for x in xs], ahead_step=ahead_step, ahead_shrink=ahead_shrink)
    np.random.seed(seed)
    chain = sampler.sample(start, num, burn)
    assert chain.shape == g[name + '_chain'].shape
    assert np.array_equal(chain, g[name + '_chain'])
    assert np.random.random() == float(g[name + '_next_random'])      # the stream is where the reference leaves it
    assert sampler.consumed == int(g[name + '_calls'])                 # density values the reference asked for
    if (ahead_step, ahead_shrink) == (3, 4):
      assert sampler.batches * 2.5 < sampler.consumed                 # ... fetched in far fewer calls
      assert sampler.evaluated < 3 * sampler.consumed
      # the stepping-out and the shrinking share their first batch: the same chain with them apart, in more calls
      apart = SpeculativeSlice(lambda xs: [logp(x) for x in xs], ahead_step=ahead_step, ahead_shrink=ahead_shrink,
                               merge_first=False)
      np.random.seed(seed)
      assert np.array_equal(apart.sample(start, num, burn), chain) and np.random.random() == float(g[name + '_next_random'])
      assert apart.consumed == sampler.consumed and apart.merged_done == 0
      assert sampler.merged_done > 0 and sampler.batches < apart.batches


def test_unreading_draws_lands_where_the_state_copy_lands():
  """ slice_sampler._GlobalStreamPosition: `pos -= 2 k` on the global MT19937 state == set_state(saved) + re-draw, at
      every position of the 624-word block (the block boundary takes the state-copy route) """
  import numpy.random as nr
  from dragonfly_amd import slice_sampler as S
  stream = S._stream()        # pylint: disable=protected-access
  assert stream.pos is not None, 'np.random is not backed by the legacy MT19937 state'
  saved = nr.get_state()
  try:
    fast_seen = slow_seen = 0
    for seed in (0, 1):
      nr.seed(seed)
      for trial in range(1500):
        ahead, hit = 4, trial % 4
        s0 = nr.get_state()
        fast = stream.room_for(ahead)
        d = nr.rand(ahead)
        if fast:
          stream.unread(ahead - (hit + 1))
          fast_seen += 1
        else:
          nr.set_state(s0)
          nr.rand(hit + 1)
          slow_seen += 1
        a = nr.get_state()
        nr.set_state(s0)
        d2 = nr.rand(hit + 1)
        b = nr.get_state()
        assert np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3:] == b[3:] and np.array_equal(d[:hit + 1], d2)
        nr.rand(trial % 7)
    assert fast_seen > 0 and slow_seen > 0
  finally:
    nr.set_state(saved)


def test_merged_first_batch_is_the_same_chain_over_many_streams():
  """ merge_first against the loops apart: the same chain and the same stream position for every seed, target and
      look-ahead -- including the updates whose speculative draws cross the 624-word block of the MT19937 state (the
      state-copy route of giving draws back) and targets with regions of zero density """
  from dragonfly_amd.slice_sampler import SpeculativeSlice
  targets = [
    lambda x: -0.5 * x * x,
    lambda x: float(np.logaddexp(-0.5 * (x + 3.0) ** 2, -2.0 * (x - 2.0) ** 2)),
    lambda x: -abs(x) ** 0.7 if -4.0 < x < 6.0 else -np.inf,
    lambda x: -50.0 * (x - 0.3) ** 2,                       # much narrower than the initial width: long shrinking
    lambda x: -0.001 * x * x,                               # much wider: long stepping-out
  ]
  merged_total = 0
  for t, logp in enumerate(targets):
    for seed in range(12):
      for ahead in ((3, 4), (1, 2), (5, 7)):
        chains, ends, stats = [], [], []
        for merge in (True, False):
          sampler = SpeculativeSlice(lambda xs, f=logp: [f(x) for x in xs], ahead_step=ahead[0], ahead_shrink=ahead[1],
                                     merge_first=merge)
          np.random.seed(1000 * t + seed)
          np.random.rand(seed * 37 % 311)                   # start at different places of the state's block
          chains.append(sampler.sample(0.1, 60, 25))
          ends.append(np.random.random())
          stats.append((sampler.consumed, sampler.batches, sampler.merged_done))
        assert np.array_equal(chains[0], chains[1]) and ends[0] == ends[1]
        assert stats[0][0] == stats[1][0] and stats[0][1] <= stats[1][1] and stats[1][2] == 0
        merged_total += stats[0][2]
  assert merged_total > 1000
