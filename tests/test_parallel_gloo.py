"""CPU, world_size 2, gloo: the cross-rank arg-max exchange used when candidates shard across GPUs
(dragonfly_amd/parallel.py).  On the GPU box the same code runs over RCCL (backend "nccl")."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from dragonfly_amd import parallel
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%%s' %% os.environ['PORT'],
                            rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
    rank, world = dist.get_rank(), dist.get_world_size()
    rs = np.random.RandomState(11)
    ok = True
    for trial in range(30):
      m = int(rs.randint(1, 40))
      vals = rs.randint(0, 5, size=m).astype(float)
      if trial %% 4 == 0:
        vals[rs.randint(0, m)] = np.nan
      lo, hi = parallel.shard_bounds(m, rank, world, align=(4 if trial %% 2 else 1))
      if hi > lo:
        j = int(np.argmax(vals[lo:hi])); lv, li = vals[lo + j], lo + j
      else:
        lv, li = float('nan'), -1
      v, i = parallel.allgather_argmax(lv, li)
      ok &= (i == int(np.argmax(vals)))
      ok &= (v != v) if np.isnan(vals[i]) else (v == vals[i])
    # sharded device-generated candidates: a stand-in engine / GP with the product's interface
    # (rows of the seeded host draw; a fixed linear acquisition), so that the sharding, the
    # exchange of the winner and of its point are exercised without a GPU
    class Shard(object):
      def __init__(self, a): self.a, self.shape = a, a.shape
      def row(self, i): return self.a[i].copy()
    class Eng(object):
      def random_candidates(self, num, dim, bounds=None, rng=None, out=None, rows=None):
        full = rng.random_sample((num, dim)) * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0]
        return Shard(full[rows[0]:rows[0] + rows[1]])
    class FakeGP(object):
      engine = Eng()
      def acq_argmax(self, acq, cands, params=(0.0, 0.0), mean_const=0.0):
        vals = cands.a.dot(np.arange(1, cands.a.shape[1] + 1.0)) if len(cands.a) else np.zeros(0)
        j = int(np.argmax(vals)); return float(vals[j]), j
    bounds = np.array([[-1.0, 2.0], [0.0, 1.0], [3.0, 5.0]])
    for m in (1, 2, 7, 100):
      rs_all = np.random.RandomState(5)
      full = rs_all.random_sample((m, 3)) * (bounds[:, 1] - bounds[:, 0]) + bounds[:, 0]
      want = int(np.argmax(full.dot(np.arange(1, 4.0))))
      rs_rank = np.random.RandomState(5)
      v, i, pt = parallel.sharded_rand_acq_argmax(FakeGP(), 'ucb', m, bounds, rank=rank, world_size=world, rng=rs_rank)
      ok &= (i == want) and np.array_equal(pt, full[want]) and v == float(full[want].dot(np.arange(1, 4.0)))
      ok &= np.array_equal(rs_rank.random_sample(3), rs_all.random_sample(3))
    print('RANK', rank, 'OK' if ok else 'FAIL')
    dist.destroy_process_group()
''') % ROOT


def test_allgather_argmax_two_ranks(tmp_path):
  pytest.importorskip('torch')
  script = tmp_path / 'worker.py'
  script.write_text(WORKER)
  port = str(29500 + os.getpid() % 2000)
  procs = []
  for rank in range(2):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', PORT=port, MASTER_ADDR='127.0.0.1')
    procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True))
  outs = [p.communicate(timeout=240)[0] for p in procs]
  for rank, (p, out) in enumerate(zip(procs, outs)):
    assert p.returncode == 0, out
    assert 'RANK %d OK' % rank in out, out
