"""GPU: the multi-GPU entry points of the C-ABI (dfh_mgpu_*, dfh_comm_*; csrc/mgpu.hip) on the one
device a test box has.  The N = 1 path goes through the same code as N = 8 -- per-device context,
replicated fit, shard evaluation, RCCL all-gather of the (value, index) pair, reduce -- and must
equal the single-device calls bit for bit (reference semantics: one array, obj_vals.argmax(),
dragonfly/utils/oper_utils.py:73)."""
import ctypes as C

import numpy as np
import pytest

from dragonfly_amd import _lib, parallel
from dragonfly_amd.engine import KernelSpec

pytestmark = pytest.mark.gpu


def _problem(n=700, d=5, m=3000, seed=3):
  rs = np.random.RandomState(seed)
  X = rs.rand(n, d)
  Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
  spec = KernelSpec('se', d, float(Y.var()), np.full(d, 0.4))
  return X, Y - np.median(Y), float(np.median(Y)), float(Y.var() / 20), spec, rs.rand(m, d), rs.randn(m)


def test_mgpu_one_device_equals_single_device_calls(engine):
  X, yc, mean_c, noise, spec, cands, U = _problem()
  gp = engine.gp_fit(spec, X, yc, noise)
  want_ts = gp.thompson(cands, U, block=512, mean_const=mean_c)
  want_ei = gp.acq_argmax('ei', cands, params=(float(yc.max() + mean_c), 0.0), mean_const=mean_c)
  mg = parallel.MultiEngine(1)
  try:
    lml = mg.fit(spec, X, yc, noise)
    assert lml == [gp.lml]
    # host shards, then shards resident in the rank's HBM
    got = mg.thompson([cands], [U], block=512, mean_const=mean_c, return_local=True)
    assert got[:2] == want_ts and got[2] == [want_ts]
    cd, ud = mg.engines[0].to_device(cands), mg.engines[0].to_device(U)
    assert mg.thompson([cd], [ud], block=512, mean_const=mean_c) == want_ts
    assert mg.acq_argmax('ei', [cd], params=(float(yc.max() + mean_c), 0.0), mean_const=mean_c) == want_ei
    # the communicator as RCCL itself reports it (dfh_comm_info: ncclCommCount / ncclCommUserRank / ncclGetVersion)
    info = mg.comm_info(0)
    assert info['backend'] == 'rccl' and info['ranks_formed'] == 1 and info['rank'] == 0 and info['rccl_version'] > 20000
    # the exchange alone: NaN-first / lowest-index rule comes back through RCCL unchanged
    assert mg.allgather_argmax([2.5], [17]) == (2.5, 17)
    v, i = mg.allgather_argmax([float('nan')], [4])
    assert v != v and i == 4
    # a second fit replaces the first
    mg.fit(spec, X[:300], yc[:300], noise)
    g2 = engine.gp_fit(spec, X[:300], yc[:300], noise)
    assert mg.thompson([cd], [ud], block=512, mean_const=mean_c) == g2.thompson(cands, U, block=512, mean_const=mean_c)
    cd.free()
    ud.free()
  finally:
    mg.close()


def test_mgpu_rank_loop_on_one_device_matches_unsharded(engine):
  """ what N devices compute, computed rank after rank on one device, pushed through the RCCL
      exchange: equal to the unsharded call (shards cut on block boundaries) """
  X, yc, mean_c, noise, spec, cands, U = _problem(m=4096 + 700)
  gp = engine.gp_fit(spec, X, yc, noise)
  B = 256
  want = gp.thompson(cands, U, block=B, mean_const=mean_c)
  mg = parallel.MultiEngine(1)
  try:
    for world in (2, 3, 8):
      vals, idxs = [], []
      for r in range(world):
        lo, hi = parallel.shard_bounds(len(cands), r, world, align=B)
        if hi > lo:
          v, i = gp.thompson(cands[lo:hi], U[lo:hi], block=B, mean_const=mean_c)
          vals.append(v)
          idxs.append(i + lo)
        else:
          vals.append(float('nan'))
          idxs.append(-1)
      assert parallel.reduce_argmax(vals, idxs) == want
      # one pair at a time through the device collective (a 1-rank communicator), then the reduce
      through = [mg.allgather_argmax([v], [i]) for v, i in zip(vals, idxs)]
      live = [(v, i) for v, i in through if i >= 0]
      assert parallel.reduce_argmax([p[0] for p in live], [p[1] for p in live]) == want
  finally:
    mg.close()


def test_mgpu_more_devices_than_visible_fails_loudly():
  with pytest.raises(_lib.DfhipError):
    parallel.MultiEngine(_lib.device_count() + 1)
  h = C.c_void_p()
  rc = _lib.load().dfh_mgpu_create(_lib.device_count() + 1, None, C.byref(h))
  assert rc == _lib.DFH_ERR_BAD_ARG and 'visible' in _lib.last_error()


def test_process_per_gpu_communicator_single_rank(engine, monkeypatch, tmp_path):
  """ dfh_comm_*: ncclGetUniqueId -> file rendezvous -> ncclCommInitRank, world size 1 """
  monkeypatch.setenv('RANK', '0')
  monkeypatch.setenv('WORLD_SIZE', '1')
  monkeypatch.setenv('DFH_RDZV_DIR', str(tmp_path))
  comm = parallel.RcclComm.from_env(engine, key='gpu_test')
  try:
    assert (comm.rank, comm.size) == (0, 1)
    info = comm.info()
    assert (info['ranks_formed'], info['rank']) == (1, 0) and info['rccl_version'] > 20000
    assert comm.allgather_argmax(1.25, 7) == (1.25, 7)
    v, i = comm.allgather_argmax(float('nan'), -1)       # every shard empty
    assert i == -1
    row = np.arange(5.0)
    assert np.array_equal(comm.allgather_rows(row, True), row)
    assert np.array_equal(comm.allreduce_max([3.0, -1.0]), [3.0, -1.0])
    comm.barrier()
    X, yc, mean_c, noise, spec, cands, U = _problem(n=200, m=500)
    gp = engine.gp_fit(spec, X, yc, noise)
    assert parallel.sharded_thompson(gp, cands, U, 128, mean_const=mean_c, comm=comm) == \
        gp.thompson(cands, U, block=128, mean_const=mean_c)
  finally:
    comm.close()
  assert not list(tmp_path.iterdir())       # rank 0 removed the rendezvous file


@pytest.mark.parametrize('nranks', [2, 4, 8])
def test_mgpu_concurrent_contexts_on_one_device(engine, nranks, monkeypatch):
  """ N contexts and N host threads driven CONCURRENTLY through dfh_mgpu_fit / _ts / _acq_argmax on the
      one device of the box (library test switch; the 16-byte pairs are reduced on the host because
      RCCL refuses duplicate devices): what would first show up on an 8-GPU node -- races in the
      per-context pools, per-device function attributes, thread-local error state, the one-launch
      panels' flag hand-offs with several launches in flight -- has to show up here.  Results are
      those of the unsharded single-context calls, bit for bit, every time. """
  monkeypatch.setenv('DFH_MGPU_ALLOW_DUPLICATE_DEVICES', '1')
  n, d, B = 2500, 6, 256
  m = nranks * 1536 - 5 * B + 77                          # ragged: the last shard is short
  X, yc, mean_c, noise, spec, cands, U = _problem(n=n, d=d, m=m, seed=40 + nranks)
  gp = engine.gp_fit(spec, X, yc, noise)
  want_ts = gp.thompson(cands, U, block=B, mean_const=mean_c)
  best = float(yc.max() + mean_c)
  want_ei = gp.acq_argmax('ei', cands, params=(best, 0.0), mean_const=mean_c)
  want_ucb = gp.acq_argmax('ucb', cands, params=(2.0, 0.0), mean_const=mean_c)
  bounds = [parallel.shard_bounds(m, r, nranks, align=B) for r in range(nranks)]
  mg = parallel.MultiEngine(nranks, device_ids=[0] * nranks)
  try:
    cs = [cands[lo:hi] for lo, hi in bounds]
    us = [U[lo:hi] for lo, hi in bounds]
    for rep in range(3):
      lml = mg.fit(spec, X, yc, noise)
      assert lml == [gp.lml] * nranks, rep
      got = mg.thompson(cs, us, block=B, mean_const=mean_c, return_local=True)
      assert got[:2] == want_ts, rep
      # every rank's local winner is what a single context finds on that shard
      if rep == 0:
        for r, (lo, hi) in enumerate(bounds):
          if hi > lo:
            v, i = gp.thompson(cands[lo:hi], U[lo:hi], block=B, mean_const=mean_c)
            assert got[2][r] == (v, i + lo), r
      assert mg.acq_argmax('ei', cs, params=(best, 0.0), mean_const=mean_c) == want_ei, rep
      assert mg.acq_argmax('ucb', cs, params=(2.0, 0.0), mean_const=mean_c) == want_ucb, rep
    # shards resident in each context's memory, a smaller second fit replacing the first
    cd = [e.to_device(c) for e, c in zip(mg.engines, cs)]
    ud = [e.to_device(u) for e, u in zip(mg.engines, us)]
    assert mg.thompson(cd, ud, block=B, mean_const=mean_c) == want_ts
    mg.fit(spec, X[:900], yc[:900], noise)
    g2 = engine.gp_fit(spec, X[:900], yc[:900], noise)
    assert mg.thompson(cd, ud, block=B, mean_const=mean_c) == g2.thompson(cands, U, block=B, mean_const=mean_c)
    for a in cd + ud:
      a.free()
  finally:
    mg.close()
