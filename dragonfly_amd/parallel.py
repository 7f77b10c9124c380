"""Multi-GPU acquisition: candidates shard across the GPUs of a node, one process per GPU.

The reference evaluates every candidate in one NumPy array in one process (SURVEY.md section 8e:
no collective exists in Dragonfly).  Candidates are independent given the fitted GP, so each rank
fits the (replicated) GP on its own GPU, evaluates a contiguous shard of the candidate set with
the fused device call, and the ranks exchange only their local (value, global index) pairs: an
all-gather of 16 bytes per rank over RCCL/xGMI (torch.distributed backend "nccl"; "gloo" in the
CPU tests), followed by the same deterministic reduction on every rank -- first NaN wins, else the
largest value, ties to the lowest global index, i.e. exactly np.argmax over the whole set
(dragonfly/utils/oper_utils.py:73).  RCCL has no MAXLOC and an all-reduce(max) alone would lose
the index, hence gather-then-reduce.
"""
import numpy as np


def shard_bounds(m, rank, world_size, align=1):
  """ [lo, hi) of rank's contiguous shard of m candidates; shard edges fall on multiples of
      `align` (the TS block size, so blocked-joint sampling is rank-count invariant). """
  nblk = (m + align - 1) // align
  per = (nblk + world_size - 1) // world_size
  lo = min(m, rank * per * align)
  hi = min(m, (rank + 1) * per * align)
  return lo, hi


def better(va, ia, vb, ib):
  """ np.argmax ordering between two (value, index) pairs. """
  na, nb = va != va, vb != vb
  if na or nb:
    if na and nb:
      return ia < ib
    return na
  if va > vb:
    return True
  if va < vb:
    return False
  return ia < ib


def reduce_argmax(vals, idxs):
  """ The winner among per-rank (value, global index) pairs; ranks with an empty shard pass
      idx < 0 and are skipped. """
  best_v, best_i = None, -1
  for v, i in zip(vals, idxs):
    i = int(i)
    if i < 0:
      continue
    if best_i < 0 or better(float(v), i, best_v, best_i):
      best_v, best_i = float(v), i
  return best_v, best_i


def allgather_argmax(local_val, local_idx, group=None, device=None):
  """ All-gather (value, global index) over the process group and reduce. Works with any
      torch.distributed backend; `device` is the tensor device ('cuda:<n>' for nccl/RCCL). """
  import torch
  import torch.distributed as dist
  if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
    return float(local_val), int(local_idx)
  world = dist.get_world_size(group)
  if device is None:
    device = 'cuda:%d' % torch.cuda.current_device() if dist.get_backend(group) == 'nccl' else 'cpu'
  # value and index travel as two float64 / int64 tensors (bit-exact; NaNs preserved)
  v = torch.tensor([float(local_val)], dtype=torch.float64, device=device)
  i = torch.tensor([int(local_idx)], dtype=torch.int64, device=device)
  vs = [torch.empty_like(v) for _ in range(world)]
  is_ = [torch.empty_like(i) for _ in range(world)]
  dist.all_gather(vs, v, group=group)
  dist.all_gather(is_, i, group=group)
  return reduce_argmax([float(t.item()) for t in vs], [int(t.item()) for t in is_])


def sharded_acq_argmax(fitted_gp, acq, cands, params=(0.0, 0.0), mean_const=0.0, rank=0,
                       world_size=1, group=None, device=None):
  """ Fused acquisition arg-max of this rank's shard + the cross-rank exchange.
      Returns (best_val, best_global_idx); identical on every rank. """
  lo, hi = shard_bounds(len(cands), rank, world_size)
  if hi > lo:
    v, i = fitted_gp.acq_argmax(acq, cands[lo:hi], params=params, mean_const=mean_const)
    i += lo
  else:
    v, i = float('nan'), -1
  if world_size == 1:
    return v, i
  return allgather_argmax(v, i, group=group, device=device)


def sharded_thompson(fitted_gp, cands, U, block, mean_const=0.0, rank=0, world_size=1, group=None,
                     device=None):
  """ Blocked-joint Thompson sampling over the candidate set, sharded on block boundaries. """
  lo, hi = shard_bounds(len(cands), rank, world_size, align=block)
  if hi > lo:
    v, i = fitted_gp.thompson(cands[lo:hi], np.asarray(U)[lo:hi], block=block, mean_const=mean_const)
    i += lo
  else:
    v, i = float('nan'), -1
  if world_size == 1:
    return v, i
  return allgather_argmax(v, i, group=group, device=device)


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


def sharded_rand_acq_argmax(fitted_gp, acq, m, bounds, params=(0.0, 0.0), mean_const=0.0, rank=0,
                            world_size=1, group=None, device=None, rng=None):
  """ random_maximise of an acquisition over m candidates (oper_utils.py:70-80), candidates
      generated and evaluated shard by shard on the GPUs.  Returns (best value, global index, the
      winning point); identical on every rank: the point travels from its owner in a second tiny
      all-gather. """
  shard, lo, hi = sharded_random_candidates(fitted_gp.engine, m, bounds, rank, world_size, rng=rng)
  if hi > lo:
    v, i = fitted_gp.acq_argmax(acq, shard, params=params, mean_const=mean_const)
    i += lo
  else:
    v, i = float('nan'), -1
  if world_size > 1:
    v, i = allgather_argmax(v, i, group=group, device=device)
  mine = lo <= i < hi
  point = shard.row(i - lo) if mine else np.zeros(len(bounds))
  if world_size > 1:
    point = allgather_rows(point, mine, group=group, device=device)
  return v, i, point


def allgather_rows(row, is_owner, group=None, device=None):
  """ The row held by the one rank with is_owner set, on every rank. """
  import torch
  import torch.distributed as dist
  world = dist.get_world_size(group)
  if device is None:
    device = 'cuda:%d' % torch.cuda.current_device() if dist.get_backend(group) == 'nccl' else 'cpu'
  payload = torch.tensor(np.concatenate([[1.0 if is_owner else 0.0], np.asarray(row, dtype=np.float64)]),
                         dtype=torch.float64, device=device)
  gathered = [torch.empty_like(payload) for _ in range(world)]
  dist.all_gather(gathered, payload, group=group)
  for t in gathered:
    if float(t[0].item()) == 1.0:
      return t[1:].cpu().numpy()
  raise RuntimeError('allgather_rows: no rank owns the row.')
