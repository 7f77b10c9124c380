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
