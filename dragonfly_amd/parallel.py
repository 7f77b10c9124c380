"""Multi-GPU acquisition: candidates shard across the GPUs of a node.

The reference evaluates every candidate in one NumPy array in one process and takes
`obj_vals.argmax()` (dragonfly/utils/oper_utils.py:59-80; no collective exists in Dragonfly).
Candidates are independent given the fitted GP, so each GPU fits the (replicated) GP, evaluates a
contiguous shard of the candidate set with the fused device call, and the ranks exchange only
their local (value, global index) pairs: an RCCL all-gather of 16 bytes per rank over xGMI, then
the same deterministic reduction on every rank -- first NaN wins, else the largest value, ties to
the lowest global index, i.e. exactly np.argmax over the whole set.  RCCL has no MAXLOC and an
all-reduce(max) alone would lose the index, hence gather-then-reduce.

Everything here goes through libdfhip.so (csrc/mgpu.hip: RCCL by dlopen); there is no PyTorch:

  * `MultiEngine`  -- ONE process, N devices (dfh_mgpu_*: a context and a host thread per device,
                      ncclCommInitAll).  `python bench.py --gpus N` without a launcher.
  * `RcclComm`     -- one process PER GPU (dfh_comm_*: ncclGetUniqueId / ncclCommInitRank), the
                      128-byte unique id passed through a file keyed by the launcher's rendezvous
                      (MASTER_PORT, run id, parent pid).  What a `torch.distributed.run` launch
                      of bench.py uses -- the launcher only provides RANK / WORLD_SIZE.
  * `sharded_*`    -- the shard / evaluate / exchange helpers, written against a communicator
                      object (rank, size, allgather_argmax, allgather_rows) so that the CPU tests
                      can drive them with a stand-in transport.
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib
from ._lib import check


# ---- the host-only contract (C-ABI: dfh_shard_bounds, dfh_reduce_argmax) ---------------------
def shard_bounds(m, rank, world_size, align=1):
  """ [lo, hi) of rank's contiguous shard of m candidates; shard edges fall on multiples of
      `align` (the TS block size, so blocked-joint sampling is rank-count invariant). """
  lo, hi = C.c_int64(0), C.c_int64(0)
  check(_lib.load().dfh_shard_bounds(int(m), int(rank), int(world_size), int(align), C.byref(lo), C.byref(hi)))
  return int(lo.value), int(hi.value)


def reduce_argmax(vals, idxs):
  """ The winner among per-rank (value, global index) pairs -- np.argmax ordering: first NaN, else
      the largest value, ties to the lowest index; ranks with an empty shard pass idx < 0 and are
      skipped.  Returns (None, -1) when every shard was empty. """
  v = np.ascontiguousarray(vals, dtype=np.float64).reshape(-1)
  i = np.ascontiguousarray(idxs, dtype=np.int64).reshape(-1)
  bv, bi = C.c_double(0), C.c_int64(-1)
  check(_lib.load().dfh_reduce_argmax(v.ctypes.data_as(_lib.c_double_p), i.ctypes.data_as(_lib.c_int64_p),
                                      len(v), C.byref(bv), C.byref(bi)))
  if bi.value < 0:
    return None, -1
  return float(bv.value), int(bi.value)


# ---- passing the RCCL unique id between the processes of one node ---------------------------
def _rendezvous_dir():
  """ A directory only this user can write: DFH_RDZV_DIR if given, else $XDG_RUNTIME_DIR, else
      /tmp/dfhip-<uid> (created 0700).  A directory that belongs to somebody else or is writable by
      others is refused: the id file is what every rank of the job trusts. """
  base = os.environ.get('DFH_RDZV_DIR') or os.environ.get('XDG_RUNTIME_DIR')
  if not base:
    base = os.path.join('/tmp', 'dfhip-%d' % os.getuid())
  os.makedirs(base, mode=0o700, exist_ok=True)
  st = os.stat(base)
  if st.st_uid != os.getuid() or (st.st_mode & 0o022):
    raise RuntimeError('Rendezvous directory %s is not private to this user (owner %d, mode %o); set DFH_RDZV_DIR.'
                       % (base, st.st_uid, st.st_mode & 0o777))
  return base


def _rendezvous_path(key=None):
  if key is None:
    # the launcher's rendezvous, its restart generation (a restarted worker group must not pick up
    # the id of the group that crashed) and the launcher's pid
    key = '%s_%s_%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.environ.get('TORCHELASTIC_RUN_ID', 'none'),
                           os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), os.getppid())
  name = 'dfhip_rccl_id_%s.bin' % ''.join(ch if ch.isalnum() or ch in '-_' else '_' for ch in str(key))
  return os.path.join(_rendezvous_dir(), name)


def exchange_unique_id(rank, make_id, key=None, timeout=600.0, nbytes=_lib.UNIQUE_ID_BYTES):
  """ Rank 0 creates the id (make_id() -> bytes) and publishes it atomically in a file every rank
      of this node can see; the others wait for the file.  Returns (id bytes, path).
      Rank 0 removes whatever a crashed run left under the name, writes a fresh 0600 file created with
      O_EXCL and renames it into place; the readers refuse links and files they do not own. """
  path = _rendezvous_path(key)
  if rank == 0:
    blob = bytes(make_id())
    assert len(blob) == nbytes
    tmp = '%s.%d.tmp' % (path, os.getpid())
    for stale in (path, tmp):
      try:
        os.remove(stale)
      except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
    with os.fdopen(fd, 'wb') as f:
      f.write(blob)
    os.replace(tmp, path)
    return blob, path
  deadline = time.time() + timeout
  while True:
    try:
      fd = os.open(path, os.O_RDONLY | getattr(os, 'O_NOFOLLOW', 0))
      with os.fdopen(fd, 'rb') as f:
        st = os.fstat(f.fileno())
        blob = f.read() if (st.st_uid == os.getuid() and not (st.st_mode & 0o077)) else b''
      if len(blob) == nbytes:
        return blob, path
    except OSError:
      pass
    if time.time() > deadline:
      raise RuntimeError('Timed out waiting for the RCCL unique id at %s.' % path)
    time.sleep(0.01)


class RcclComm(object):
  """ This process's rank of an RCCL communicator bound to its Engine (one process per GPU). """

  def __init__(self, engine, rank, world_size, unique_id):
    self.engine, self.rank, self.size = engine, int(rank), int(world_size)
    self.lib = engine.lib
    self.handle = None
    h = C.c_void_p()
    buf = C.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES)
    check(self.lib.dfh_comm_create(engine.ctx, self.size, self.rank, buf, C.byref(h)))
    self.handle = h

  @classmethod
  def from_env(cls, engine, key=None):
    """ RANK / WORLD_SIZE from the launcher's environment; the unique id through the file
        rendezvous.  The file is removed once every rank has joined. """
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    lib = engine.lib

    def make_id():
      buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
      check(lib.dfh_comm_unique_id(buf))
      return buf.raw
    uid, path = exchange_unique_id(rank, make_id, key=key)
    comm = cls(engine, rank, world, uid)      # returns when all ranks have arrived
    if rank == 0:
      try:
        os.remove(path)
      except OSError:
        pass
    return comm

  def allgather_argmax(self, local_val, local_idx):
    bv, bi = C.c_double(0), C.c_int64(-1)
    check(self.lib.dfh_comm_allgather_argmax(self.handle, float(local_val), int(local_idx), C.byref(bv), C.byref(bi)))
    return float(bv.value), int(bi.value)

  def allgather_rows(self, row, is_owner):
    """ The row held by the one rank with is_owner set, on every rank. """
    row = np.ascontiguousarray(row, dtype=np.float64).reshape(-1)
    send = np.concatenate([[1.0 if is_owner else 0.0], row])
    recv = np.empty((self.size, len(send)), dtype=np.float64)
    check(self.lib.dfh_comm_allgather_f64(self.handle, send.ctypes.data_as(_lib.c_double_p), len(send),
                                          recv.ctypes.data_as(_lib.c_double_p)))
    for r in range(self.size):
      if recv[r, 0] == 1.0:
        return recv[r, 1:].copy()
    raise RuntimeError('allgather_rows: no rank owns the row.')

  def allreduce_max(self, values):
    a = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
    check(self.lib.dfh_comm_allreduce_max(self.handle, a.ctypes.data_as(_lib.c_double_p), len(a)))
    return a

  def barrier(self):
    check(self.lib.dfh_comm_barrier(self.handle))

  def info(self):
    """ {'backend', 'ranks_formed', 'rank', 'rccl_version'} as the communicator itself reports them
        (ncclCommCount / ncclCommUserRank / ncclGetVersion). """
    nr, rk, ver = C.c_int32(0), C.c_int32(-1), C.c_int32(0)
    check(self.lib.dfh_comm_info(self.handle, C.byref(nr), C.byref(rk), C.byref(ver)))
    return {'backend': 'rccl', 'ranks_formed': int(nr.value), 'rank': int(rk.value), 'rccl_version': int(ver.value)}

  def close(self):
    if self.handle is not None:
      self.lib.dfh_comm_destroy(self.handle)
      self.handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:     # pylint: disable=broad-except
      pass


class HostExchangeComm(object):
  """ TEST MODE (DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1): the communicator interface of RcclComm for
      several processes that share ONE device.  RCCL refuses to form a communicator with two ranks
      on the same GPU, so a one-GPU box cannot run `torch.distributed.run --nproc-per-node N
      bench.py` over RCCL; with this stand-in the launcher route of bench.py -- environment, shard
      arithmetic, global indices, the reduce, the rank-0 JSON -- is exercised there all the same.
      The pairs travel through files in the private rendezvous directory, one file per rank and
      collective, named by a nonce rank 0 publishes; the reduce is the library's (dfh_reduce_argmax).
      Never selected unless the switch is set: a production launch goes through RcclComm. """

  def __init__(self, rank, world_size, key=None, timeout=600.0):
    self.rank, self.size, self.timeout = int(rank), int(world_size), timeout
    nonce, self._id_path = exchange_unique_id(self.rank, lambda: os.urandom(16), key=key, timeout=timeout, nbytes=16)
    self._stem = os.path.join(_rendezvous_dir(), 'dfhip_hostx_%s' % nonce.hex())
    self._seq = 0
    self._mine = []

  @classmethod
  def from_env(cls, key=None):
    return cls(int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), key=key)

  def _allgather(self, row):
    """ every rank's row (float64 vector of equal length), in rank order """
    row = np.ascontiguousarray(row, dtype=np.float64).reshape(-1)
    name = lambda r: '%s_%d_%d.bin' % (self._stem, self._seq, r)
    tmp = name(self.rank) + '.tmp'
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
    with os.fdopen(fd, 'wb') as f:
      f.write(row.tobytes())
    os.replace(tmp, name(self.rank))
    out = np.empty((self.size, len(row)))
    deadline = time.time() + self.timeout
    for r in range(self.size):
      while True:
        try:
          with open(name(r), 'rb') as f:
            blob = f.read()
          if len(blob) == row.nbytes:
            out[r] = np.frombuffer(blob, dtype=np.float64)
            break
        except OSError:
          pass
        if time.time() > deadline:
          raise RuntimeError('HostExchangeComm: rank %d never wrote collective %d.' % (r, self._seq))
        time.sleep(0.001)
    # every rank has WRITTEN this collective, hence finished READING the one before: its files can go
    for path in self._mine:
      try:
        os.remove(path)
      except OSError:
        pass
    self._mine = [name(self.rank)]
    self._seq += 1
    return out

  def allgather_argmax(self, local_val, local_idx):
    pair = np.empty(2)
    pair[0] = float(local_val)
    pair[1:].view(np.int64)[0] = int(local_idx)          # the index travels as its 8 bytes
    rows = self._allgather(pair)
    return reduce_argmax(rows[:, 0], np.ascontiguousarray(rows[:, 1]).view(np.int64))

  def allgather_rows(self, row, is_owner):
    row = np.ascontiguousarray(row, dtype=np.float64).reshape(-1)
    rows = self._allgather(np.concatenate([[1.0 if is_owner else 0.0], row]))
    owners = np.nonzero(rows[:, 0] == 1.0)[0]
    if len(owners) == 0:
      raise RuntimeError('allgather_rows: no rank owns the row.')
    return rows[owners[0], 1:].copy()

  def allreduce_max(self, values):
    return self._allgather(np.atleast_1d(values)).max(axis=0)

  def barrier(self):
    self._allgather([0.0])

  def info(self):
    return {'backend': 'host files (test mode)', 'ranks_formed': 0, 'rank': self.rank, 'rccl_version': 0}

  def close(self):
    """ A rank that closes has finished every read; rank 0 waits until all have and clears the files. """
    if self._seq < 0:
      return
    self._seq = -1
    import glob      # pylint: disable=import-outside-toplevel
    open('%s_done_%d' % (self._stem, self.rank), 'wb').close()
    if self.rank == 0:
      deadline = time.time() + self.timeout
      while len(glob.glob(self._stem + '_done_*')) < self.size and time.time() < deadline:
        time.sleep(0.001)
      for path in glob.glob(self._stem + '_*') + [self._id_path]:
        try:
          os.remove(path)
        except OSError:
          pass


# ---- one process, N devices -----------------------------------------------------------------
def _ptr_array(items):
  arr = (C.c_void_p * len(items))()
  for r, it in enumerate(items):
    p = _engine_ptr(it)
    arr[r] = p.value if isinstance(p, C.c_void_p) else p
  return arr


def _engine_ptr(a):
  from .engine import _ptr      # pylint: disable=import-outside-toplevel
  return _ptr(a)


class MultiEngine(object):
  """ N MI355X in one process (dfh_mgpu): engines[r] is rank r's Engine view (uploads, buffers);
      fit() replicates the GP; thompson() / acq_argmax() run the shards and the RCCL exchange. """

  def __init__(self, n_devices, device_ids=None):
    from .engine import Engine      # pylint: disable=import-outside-toplevel
    self.lib = _lib.load()
    self.handle = None
    n_devices = int(n_devices)
    visible = _lib.device_count()
    # (test mode of the library, DFH_MGPU_ALLOW_DUPLICATE_DEVICES=1: a device id may repeat -- N contexts
    #  and host threads on one device, pairs reduced on the host; tests/test_gpu_mgpu.py)
    shared = (os.environ.get('DFH_MGPU_ALLOW_DUPLICATE_DEVICES', '0') not in ('', '0') and device_ids is not None
              and len(set(int(i) for i in device_ids)) < len(device_ids))
    if n_devices > visible and not shared:
      raise _lib.DfhipError('%d GPUs requested, %d visible.' % (n_devices, visible))
    ids = None if device_ids is None else (C.c_int * n_devices)(*[int(i) for i in device_ids])
    h = C.c_void_p()
    check(self.lib.dfh_mgpu_create(n_devices, ids, C.byref(h)))
    self.handle = h
    self.size = n_devices
    self.engines = [Engine.from_ctx(C.c_void_p(self.lib.dfh_mgpu_ctx(h, r)),
                                    r if device_ids is None else int(device_ids[r]))
                    for r in range(n_devices)]
    self.lml = None
    self.jitter_power = None
    self._keep = None

  def comm_info(self, rank=0):
    """ Rank `rank`'s communicator as RCCL reports it (dfh_comm_info); ranks_formed == 0: the host-exchange test mode. """
    nr, rk, ver = C.c_int32(0), C.c_int32(-1), C.c_int32(0)
    comm = C.c_void_p(self.lib.dfh_mgpu_comm(self.handle, int(rank)))
    if not comm:
      return {'backend': 'none (one device)' if self.size == 1 else 'host exchange (test mode)', 'ranks_formed': 0, 'rank': int(rank), 'rccl_version': 0}
    check(self.lib.dfh_comm_info(comm, C.byref(nr), C.byref(rk), C.byref(ver)))
    return {'backend': 'rccl' if nr.value > 0 else 'host exchange (test mode)', 'ranks_formed': int(nr.value),
            'rank': int(rk.value), 'rccl_version': int(ver.value)}

  def _per_rank(self, a):
    """ One entry per rank: a list is taken as is, anything else is shared by all ranks. """
    if isinstance(a, (list, tuple)):
      if len(a) != self.size:
        raise ValueError('Need one entry per device (%d), got %d.' % (self.size, len(a)))
      return list(a)
    return [a] * self.size

  def fit(self, spec, X, y_centred, noise_var, allow_jitter=True):
    """ dfh_gp_fit on every device.  X / y_centred: a host array (shared) or a list with one
        host array / DeviceArray per rank.  Returns the per-rank lml list (all equal). """
    from .engine import DeviceArray, _f64      # pylint: disable=import-outside-toplevel
    Xs = [x if isinstance(x, DeviceArray) else _f64(x) for x in self._per_rank(X)]
    ys = [y if isinstance(y, DeviceArray) else _f64(y) for y in self._per_rank(y_centred)]
    n, d = Xs[0].shape
    desc = spec.to_desc()
    lml = (C.c_double * self.size)()
    jp = (C.c_int32 * self.size)()
    check(self.lib.dfh_mgpu_fit(self.handle, C.byref(desc), _ptr_array(Xs), n, d, _ptr_array(ys),
                                float(noise_var), 0 if allow_jitter else _lib.FIT_NO_JITTER, lml, jp))
    self.lml = list(lml)
    self.jitter_power = [None if p == _lib.INT32_MIN else int(p) for p in jp]
    return self.lml

  def free_fit(self):
    check(self.lib.dfh_mgpu_free_fit(self.handle))

  def gp_handle(self, rank):
    return C.c_void_p(self.lib.dfh_mgpu_gp(self.handle, int(rank)))

  @staticmethod
  def _shards(shards):
    from .engine import DeviceArray, _f64      # pylint: disable=import-outside-toplevel
    out = [s if isinstance(s, DeviceArray) or s is None else _f64(s) for s in shards]
    sizes = np.ascontiguousarray([0 if s is None else s.shape[0] for s in out], dtype=np.int64)
    return out, sizes

  def thompson(self, cand_shards, U_shards, block=4096, mean_const=0.0, return_local=False):
    """ Blocked-joint Thompson sampling over the concatenation of the shards (cut them with
        shard_bounds(..., align=block)); returns (best value, GLOBAL index[, per-rank winners]). """
    Xs, ms = self._shards(cand_shards)
    Us, _ = self._shards([None if u is None else np.ravel(u) if not hasattr(u, 'ptr') else u for u in U_shards])
    bv, bi = C.c_double(0), C.c_int64(-1)
    lv = (C.c_double * self.size)()
    li = (C.c_int64 * self.size)()
    check(self.lib.dfh_mgpu_ts(self.handle, _ptr_array(Xs), ms.ctypes.data_as(_lib.c_int64_p), int(block),
                               _ptr_array(Us), float(mean_const), C.byref(bv), C.byref(bi), lv, li))
    if return_local:
      return float(bv.value), int(bi.value), list(zip(list(lv), list(li)))
    return float(bv.value), int(bi.value)

  def acq_argmax(self, acq, cand_shards, params=(0.0, 0.0), mean_const=0.0, return_local=False):
    from .engine import ACQ_IDS      # pylint: disable=import-outside-toplevel
    Xs, ms = self._shards(cand_shards)
    p = (C.c_double * 2)(float(params[0]), float(params[1]) if len(params) > 1 else 0.0)
    bv, bi = C.c_double(0), C.c_int64(-1)
    lv = (C.c_double * self.size)()
    li = (C.c_int64 * self.size)()
    check(self.lib.dfh_mgpu_acq_argmax(self.handle, ACQ_IDS[acq], p, _ptr_array(Xs),
                                       ms.ctypes.data_as(_lib.c_int64_p), float(mean_const), C.byref(bv),
                                       C.byref(bi), lv, li))
    if return_local:
      return float(bv.value), int(bi.value), list(zip(list(lv), list(li)))
    return float(bv.value), int(bi.value)

  def allgather_argmax(self, vals, idxs):
    """ The exchange alone (RCCL all-gather + reduce) for per-rank pairs given by the caller. """
    v = np.ascontiguousarray(vals, dtype=np.float64)
    i = np.ascontiguousarray(idxs, dtype=np.int64)
    bv, bi = C.c_double(0), C.c_int64(-1)
    check(self.lib.dfh_mgpu_allgather_argmax(self.handle, v.ctypes.data_as(_lib.c_double_p),
                                             i.ctypes.data_as(_lib.c_int64_p), C.byref(bv), C.byref(bi)))
    return float(bv.value), int(bi.value)

  def sync(self):
    check(self.lib.dfh_mgpu_sync(self.handle))

  def close(self):
    if self.handle is not None:
      for e in self.engines:
        e.ctx = None
      self.lib.dfh_mgpu_destroy(self.handle)
      self.handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:     # pylint: disable=broad-except
      pass


# ---- shard / evaluate / exchange against a communicator ---------------------------------------
def sharded_acq_argmax(fitted_gp, acq, cands, params=(0.0, 0.0), mean_const=0.0, comm=None):
  """ Fused acquisition arg-max of this rank's shard + the cross-rank exchange.
      Returns (best_val, best_global_idx); identical on every rank. """
  rank, world = (0, 1) if comm is None else (comm.rank, comm.size)
  lo, hi = shard_bounds(len(cands), rank, world)
  if hi > lo:
    v, i = fitted_gp.acq_argmax(acq, cands[lo:hi], params=params, mean_const=mean_const)
    i += lo
  else:
    v, i = float('nan'), -1
  if world == 1:
    return v, i
  return comm.allgather_argmax(v, i)


def sharded_thompson(fitted_gp, cands, U, block, mean_const=0.0, comm=None):
  """ Blocked-joint Thompson sampling over the candidate set, sharded on block boundaries. """
  rank, world = (0, 1) if comm is None else (comm.rank, comm.size)
  lo, hi = shard_bounds(len(cands), rank, world, align=block)
  if hi > lo:
    v, i = fitted_gp.thompson(cands[lo:hi], np.asarray(U)[lo:hi], block=block, mean_const=mean_const)
    i += lo
  else:
    v, i = float('nan'), -1
  if world == 1:
    return v, i
  return comm.allgather_argmax(v, i)


def advance_mt19937(rng, n_doubles):
  """ Move a legacy NumPy generator (None / np.random: the global state; or a RandomState) past
      n_doubles draws of random_sample without making them: polynomial jump-ahead on the host
      (dfh_mt19937_advance), a few ms whatever the distance.  The state afterwards is the state the
      draws would have left. """
  legacy = np.random if (rng is None or rng is np.random) else rng
  state = legacy.get_state()
  if state[0] != 'MT19937':
    raise ValueError('The legacy NumPy state is not MT19937.')
  key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
  pos = C.c_int32(int(state[2]))
  check(_lib.load().dfh_mt19937_advance(key.ctypes.data_as(C.c_void_p), C.byref(pos), 2 * int(n_doubles)))
  legacy.set_state((state[0], key, int(pos.value)) + tuple(state[3:]))


def sharded_random_candidates(engine, m, bounds, rank=0, world_size=1, align=1, rng=None):
  """ This rank's shard of the m random candidates of a 'rand' acquisition, generated in HBM
      (Engine.random_candidates, rows=...).  Every rank must call this with the generator in the
      same state -- seed the ranks alike, as the single process of the reference is seeded once --
      and every rank's generator ends in the state of the full draw, so the draws that follow
      (the TS normals) are the same everywhere.  Returns (DeviceArray [hi - lo x d], lo, hi). """
  bounds = np.asarray(bounds, dtype=np.float64)
  lo, hi = shard_bounds(int(m), rank, world_size, align=align)
  shard = engine.random_candidates(int(m), len(bounds), bounds=bounds, rng=rng, rows=(lo, hi - lo))
  return shard, lo, hi


def sharded_rand_acq_argmax(fitted_gp, acq, m, bounds, params=(0.0, 0.0), mean_const=0.0, comm=None, rng=None):
  """ random_maximise of an acquisition over m candidates (oper_utils.py:70-80), candidates
      generated and evaluated shard by shard on the GPUs.  Returns (best value, global index, the
      winning point); identical on every rank: the point travels from its owner in a second tiny
      all-gather. """
  rank, world = (0, 1) if comm is None else (comm.rank, comm.size)
  shard, lo, hi = sharded_random_candidates(fitted_gp.engine, m, bounds, rank, world, rng=rng)
  if hi > lo:
    v, i = fitted_gp.acq_argmax(acq, shard, params=params, mean_const=mean_const)
    i += lo
  else:
    v, i = float('nan'), -1
  if world > 1:
    v, i = comm.allgather_argmax(v, i)
  mine = lo <= i < hi
  point = shard.row(i - lo) if mine else np.zeros(len(bounds))
  if world > 1:
    point = comm.allgather_rows(point, mine)
  return v, i, point
