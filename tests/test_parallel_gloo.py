"""CPU, world_size 2: the shard / evaluate / exchange helpers of dragonfly_amd/parallel.py driven
by a stand-in transport (torch.distributed "gloo" behind the communicator interface; test
infrastructure only).  The product's transport is RCCL inside libdfhip.so (parallel.RcclComm /
parallel.MultiEngine, csrc/mgpu.hip); the shard bounds and the reduce are the library's own
host functions (dfh_shard_bounds, dfh_reduce_argmax) in both cases.  Also: the file rendezvous
that carries the RCCL unique id between the processes of a node, with two real processes."""
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
    import torch
    class GlooComm(object):
      """ the communicator interface of parallel.RcclComm over gloo """
      rank, size = rank, world
      def _gather(self, vec):
        t = torch.tensor(np.asarray(vec, dtype=np.float64))
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [o.numpy() for o in out]
      def allgather_argmax(self, v, i):
        vs = self._gather([float(v)])
        t = torch.tensor([int(i)], dtype=torch.int64)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return parallel.reduce_argmax([a[0] for a in vs], [int(o.item()) for o in out])
      def allgather_rows(self, row, is_owner):
        for a in self._gather(np.concatenate([[1.0 if is_owner else 0.0], row])):
          if a[0] == 1.0:
            return a[1:].copy()
        raise RuntimeError('no owner')
    comm = GlooComm()
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
      v, i = comm.allgather_argmax(lv, li)
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
      v, i, pt = parallel.sharded_rand_acq_argmax(FakeGP(), 'ucb', m, bounds, comm=comm, rng=rs_rank)
      ok &= (i == want) and np.array_equal(pt, full[want]) and v == float(full[want].dot(np.arange(1, 4.0)))
      ok &= np.array_equal(rs_rank.random_sample(3), rs_all.random_sample(3))
    # sharded_acq_argmax / sharded_thompson with a stand-in fitted GP (values = a fixed function of the rows)
    class FakeGP2(object):
      def acq_argmax(self, acq, cands, params=(0.0, 0.0), mean_const=0.0):
        vals = np.sin(7 * cands.sum(axis=1)); j = int(np.argmax(vals)); return float(vals[j]), j
      def thompson(self, cands, U, block=4, mean_const=0.0):
        vals = np.sin(7 * cands.sum(axis=1)) + U; j = int(np.argmax(vals)); return float(vals[j]), j
    for m in (1, 3, 9, 64):
      rs2 = np.random.RandomState(m)
      cands, U = rs2.rand(m, 2), rs2.randn(m)
      v, i = parallel.sharded_acq_argmax(FakeGP2(), 'ei', cands, comm=comm)
      ok &= i == int(np.argmax(np.sin(7 * cands.sum(axis=1))))
      v, i = parallel.sharded_thompson(FakeGP2(), cands, U, 4, comm=comm)
      ok &= i == int(np.argmax(np.sin(7 * cands.sum(axis=1)) + U))
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


RDZV_WORKER = textwrap.dedent('''
    import os, sys, hashlib
    sys.path.insert(0, %r)
    from dragonfly_amd import parallel
    rank = int(os.environ['RANK'])
    blob, path = parallel.exchange_unique_id(rank, lambda: os.urandom(128), key=os.environ['KEY'], timeout=60)
    print('RANK', rank, hashlib.sha1(blob).hexdigest())
''') % ROOT


def test_unique_id_file_rendezvous_two_processes(tmp_path):
  """ what carries ncclGetUniqueId's 128 bytes from rank 0 to the other processes (no torch) """
  script = tmp_path / 'rdzv.py'
  script.write_text(RDZV_WORKER)
  key = 'test_%d' % os.getpid()
  procs = []
  for rank in (1, 0):       # the reader starts first and has to wait for the writer
    env = dict(os.environ, RANK=str(rank), KEY=key, DFH_RDZV_DIR=str(tmp_path))
    procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True))
  outs = [p.communicate(timeout=120)[0] for p in procs]
  assert all(p.returncode == 0 for p in procs), outs
  digests = [o.split()[-1] for o in outs]
  assert digests[0] == digests[1] and len(digests[0]) == 40, outs


HOSTX_WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    from dragonfly_amd import parallel
    comm = parallel.HostExchangeComm.from_env(key=os.environ['KEY'])
    rank, world = comm.rank, comm.size
    rs = np.random.RandomState(5)
    ok = True
    for trial in range(25):
      m = int(rs.randint(1, 50))
      vals = rs.randint(0, 4, size=m).astype(float)
      if trial %% 5 == 0:
        vals[rs.randint(0, m)] = np.nan
      lo, hi = parallel.shard_bounds(m, rank, world, align=(8 if trial %% 2 else 1))
      if hi > lo:
        j = int(np.argmax(vals[lo:hi])); lv, li = vals[lo + j], lo + j
      else:
        lv, li = float('nan'), -1
      v, i = comm.allgather_argmax(lv, li)
      ok &= (i == int(np.argmax(vals)))
      ok &= (v != v) if np.isnan(vals[i]) else (v == vals[i])
    ok &= float(comm.allreduce_max([float(rank), -float(rank)])[0]) == world - 1
    row = comm.allgather_rows(np.arange(4.0) + rank, rank == world - 1)
    ok &= row[0] == world - 1
    comm.barrier()
    comm.close()
    print('OK' if ok else 'MISMATCH')
''')


def test_host_exchange_comm_three_processes(tmp_path):
  """ parallel.HostExchangeComm (the test-mode stand-in bench.py uses when several launcher processes share
      one device, where RCCL refuses to form a communicator): same interface and same reduce as RcclComm,
      three real processes, files in a private directory; nothing is left behind """
  script = tmp_path / 'hostx_worker.py'
  script.write_text(HOSTX_WORKER % ROOT)
  rdzv = tmp_path / 'rdzv'
  rdzv.mkdir(mode=0o700)
  procs = []
  for r in range(3):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE='3', KEY='hostx-test', DFH_RDZV_DIR=str(rdzv))
    procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
  outs = [p.communicate(timeout=120) for p in procs]
  for p, (out, err) in zip(procs, outs):
    assert p.returncode == 0, err[-2000:]
    assert out.strip().endswith('OK'), (out, err[-500:])
  assert os.listdir(str(rdzv)) == []
