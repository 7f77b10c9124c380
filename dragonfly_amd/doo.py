"""Optimistic tree search (DOO / parallel DOO) with a batched evaluation frontier.

The reference maximises an acquisition with acq_opt_method 'pdoo' -- and with 'direct' whenever
its Fortran DIRECT is not built, dragonfly/utils/oper_utils.py:130-133 -- by the tree search of
dragonfly/utils/doo.py: the most optimistic leaf box is popped, halved along one coordinate, and
the objective is evaluated at the centre of each half, ONE point per Python callback
(doo.py:112-121).  On the MI355X a single-point posterior costs the same as a few hundred: every
call streams the factor L from HBM once.  This module therefore keeps the search itself -- which
box is popped, how it is split, the budget accounting over the N restarts of PDOO with their
decaying rho, the final choice -- decision for decision the reference's (doo.py:123-240), and
changes only WHERE the objective values come from: a cache that is filled a *frontier* at a time.
When the search needs a centre that is not cached, the centres it is likely to need next (the
halves of the F most optimistic leaves and the next two generations below the box being split) are
evaluated with it in one vectorised call.  Every value is the objective at exactly the point the
reference would query, so the search visits the same boxes and returns the same point; what
changes is the number of device calls.

Re-queries of a known box at another fidelity (doo.py:135-147) are answered from the cache too:
for a deterministic objective they return the same number, and they are still charged to the
budget as the reference charges them.
"""
import heapq
from argparse import Namespace

import numpy as np

from . import gaplog
from .general_utils import map_to_bounds


class _Leaf(object):
  """ A box [lo, hi] of the unit cube with the objective at its centre.  Ordered for heapq so that
      the largest upper bound is popped first, ties resolved by the heap exactly as
      queue.PriorityQueue resolves them in the reference (doo.py:88-95). """
  __slots__ = ('lo', 'hi', 'value', 'fidel', 'bound', 'height', 'split_dim')

  def __init__(self, lo, hi, value, fidel, bound, height, split_dim):
    self.lo, self.hi = lo, hi
    self.value, self.fidel, self.bound = value, fidel, bound
    self.height, self.split_dim = height, split_dim

  def __lt__(self, other):
    return other.bound < self.bound

  def key(self):
    return _box_key(self.lo, self.hi)


def _box_key(lo, hi):
  return lo.tobytes() + hi.tobytes()


def _halving_dim(lo, hi, parent_dim):
  """ The coordinate a box is split along: its longest side (first of equals), but never the one
      its parent was split along twice in a row (doo.py:163-170). """
  dim = int(np.argmax(np.abs(hi - lo)))
  if dim == parent_dim:
    dim = (parent_dim - 1) % len(lo)
  return dim


def _sub_boxes(lo, hi, dim, arity):
  """ The `arity` boxes a box is cut into along `dim` (np.linspace edges, doo.py:173-181). """
  edges = np.linspace(lo[dim], hi[dim], arity + 1)
  boxes = []
  for i in range(arity):
    sub_lo, sub_hi = lo.copy(), hi.copy()
    sub_lo[dim], sub_hi[dim] = edges[i], edges[i + 1]
    boxes.append((sub_lo, sub_hi))
  return boxes


class FrontierEvaluator(object):
  """ Objective values at box centres, cached, fetched a frontier at a time.

      obj_rows  maps an array [m x d] of points (original coordinates) to m values.
      frontier  how many of the most optimistic open leaves have their halves prefetched on a miss
      depth     how many generations below the box being split are prefetched on a miss
      (frontier = depth = 0: one call per requested point, the reference's access pattern). """

  def __init__(self, obj_rows, bounds, frontier=32, depth=2):
    self.obj_rows = obj_rows
    self.bounds = np.asarray(bounds, dtype=np.float64)
    self.frontier, self.depth = int(frontier), int(depth)
    self.cache = {}
    self.calls = 0            # vectorised objective calls made
    self.evaluated = 0        # points evaluated in those calls
    self.requested = 0        # values the search asked for (the reference's callback count)

  @staticmethod
  def centre(lo, hi):
    return (lo + hi) / 2.0

  def value(self, lo, hi, speculate=None):
    """ The objective at the centre of [lo, hi]; `speculate()` yields further boxes worth fetching
        in the same call if this one is not cached. """
    self.requested += 1
    centre = self.centre(lo, hi)
    key = centre.tobytes()
    if key not in self.cache:
      batch, keys = [centre], {key}
      if speculate is not None and (self.frontier > 0 or self.depth > 0):
        for s_lo, s_hi in speculate():
          s_centre = self.centre(s_lo, s_hi)
          s_key = s_centre.tobytes()
          if s_key not in self.cache and s_key not in keys:
            keys.add(s_key)
            batch.append(s_centre)
      pts = map_to_bounds(np.array(batch), self.bounds)
      vals = np.asarray(self.obj_rows(pts), dtype=np.float64).ravel()
      if len(vals) != len(batch):
        raise ValueError('The objective returned %d values for %d points.' % (len(vals), len(batch)))
      self.calls += 1
      self.evaluated += len(batch)
      for c, v in zip(batch, vals):
        self.cache[c.tobytes()] = float(v)
    return self.cache[key]


class OptimisticTreeSearch(object):
  """ DOO and its restarts (PDOO) over the unit cube; see the module docstring. """

  def __init__(self, evaluator, dim, total_budget, nu_max=1.0, rho_max=0.9, arity=2, c_init=0.8,
               tol=1e-3):
    self.ev = evaluator
    self.dim = int(dim)
    self.total_budget = total_budget
    self.nu_max, self.rho_max, self.arity = nu_max, rho_max, int(arity)
    self.C, self.tol = c_init, tol
    self.known = {}           # box key -> [value, fidelity] of its latest query (doo.py:109)
    self.query_points = []    # centres in the order the reference would have queried them
    self.query_vals = []

  # -- one query -------------------------------------------------------------------------------
  def _query(self, lo, hi, height, rho, nu, split_dim, speculate):
    """ doo.py:123-157: the value of a box at the fidelity its diameter asks for, the cost charged
        for it (0 when a value of that fidelity is already known), and its upper bound. """
    diam = nu * (rho ** height)
    z = min(max(1 - diam / self.C, self.tol), 1.0)
    key = _box_key(lo, hi)
    seen = self.known.get(key)
    if seen is not None and abs(seen[1] - z) <= self.tol:
      value, cost = seen[0], 0
    else:
      value = self.ev.value(lo, hi, speculate)
      if len(self.query_vals) <= self.total_budget:
        self.query_points.append(self.ev.centre(lo, hi))
        self.query_vals.append(value)
      if seen is not None:
        if abs(value - seen[0]) > self.C * abs(seen[1] - z):
          self.C = 2.0 * self.C
        seen[0], seen[1] = value, z
      else:
        self.known[key] = [value, z]
      cost = 1.0
    bound = diam + self.C * (1.0 - z) + value
    return _Leaf(lo, hi, value, z, bound, height, split_dim), cost

  # -- speculation -----------------------------------------------------------------------------
  def _below(self, lo, hi, parent_dim, generations):
    """ The boxes of the next `generations` levels under [lo, hi]. """
    if generations <= 0:
      return
    dim = _halving_dim(lo, hi, parent_dim)
    for s_lo, s_hi in _sub_boxes(lo, hi, dim, self.arity):
      yield s_lo, s_hi
      for box in self._below(s_lo, s_hi, dim, generations - 1):
        yield box

  def _speculation(self, siblings, split_dim, heap):
    def _boxes():
      for box in siblings:                                   # needed for certain
        yield box
      for s_lo, s_hi in siblings:                            # likely: DOO descends where it just split
        for box in self._below(s_lo, s_hi, split_dim, self.ev.depth):
          yield box
      if self.ev.frontier > 0 and heap:
        for leaf in heapq.nsmallest(self.ev.frontier, heap):  # the next pops unless a child overtakes
          for box in self._below(leaf.lo, leaf.hi, leaf.split_dim, 1):
            yield box
    return _boxes

  # -- DOO -------------------------------------------------------------------------------------
  def _split(self, leaf, rho, nu, heap):
    """ doo.py:159-187 """
    dim = _halving_dim(leaf.lo, leaf.hi, leaf.split_dim)
    boxes = _sub_boxes(leaf.lo, leaf.hi, dim, self.arity)
    speculate = self._speculation(boxes, dim, heap)
    children, cost = [], 0
    for s_lo, s_hi in boxes:
      child, c = self._query(s_lo, s_hi, leaf.height + 1, rho, nu, dim, speculate)
      children.append(child)
      cost = cost + c
    return children, cost

  def run_doo(self, budget, nu, rho):
    """ doo.py:189-234: returns (value, fidelity, centre in the unit cube, cost, height). """
    heap = []
    lo, hi = np.zeros(self.dim), np.ones(self.dim)
    root, cost = self._query(lo, hi, 0, rho, nu, 0, self._speculation([], 0, heap))
    heapq.heappush(heap, root)
    visited = {}              # box key -> leaf, in first-visit order
    while cost <= budget:
      leaf = heapq.heappop(heap)
      if gaplog.ENABLED and heap:
        gaplog.pair('doo_expand', leaf.bound, heap[0].bound)
      visited[leaf.key()] = leaf
      children, split_cost = self._split(leaf, rho, nu, heap)
      first = children[0]
      if np.array_equal(first.lo, leaf.lo) and np.array_equal(first.hi, leaf.hi):
        break                 # the box can no longer be halved in floating point
      cost = cost + split_cost
      for child in children:
        heapq.heappush(heap, child)
    while heap:
      leaf = heapq.heappop(heap)
      visited[leaf.key()] = leaf
    best, best_score = None, float('-inf')
    for leaf in visited.values():
      score = leaf.value - self.C * (1.0 - leaf.fidel)
      if score > best_score:
        best, best_score = leaf, score
    if best is None:
      return 0, 0, 0, cost, 0
    if gaplog.ENABLED:
      gaplog.top2('doo_best', [leaf.value - self.C * (1.0 - leaf.fidel) for leaf in visited.values()])
    return best.value, best.fidel, (best.lo + best.hi) / 2, cost, best.height

  # -- PDOO ------------------------------------------------------------------------------------
  def run_pdoo(self, mult=0.5):
    """ doo.py:236-256: N DOO runs with smoothness rho_max^(N/(N-i)), the best by value - C(1-z). """
    d_max = int(np.log(self.arity) / np.log(1 / self.rho_max))
    n = self.total_budget / 1.0
    num_runs = int(mult * d_max * np.log(n / np.log(n)))
    budget = self.total_budget / float(num_runs)
    results = []
    for i in range(num_runs):
      rho = (self.rho_max) ** (float(num_runs) / (num_runs - i))
      results.append(self.run_doo(budget, self.nu_max, rho))
    scores = [r[0] - self.C * (1 - r[1]) for r in results]
    gaplog.top2('pdoo_run', scores)
    return results, int(np.argmax(scores))


def pdoo_maximise_batched(obj_rows, bounds, max_evals, frontier=32, depth=2, return_history=False,
                          nu_max=1.0, rho_max=0.9, arity=2, c_init=0.8, tol=1e-3, mult=0.5):
  """ PDOO with the reference's constants (oper_utils.py:257-271) over a vectorised objective
      `obj_rows([m x d]) -> [m]`.  Returns (value, point, history); history (when asked for) lists
      the points in the order the reference would have queried them plus the evaluator's counters:
      `device_calls`, `points_evaluated`, `points_requested`. """
  bounds = np.asarray(bounds, dtype=np.float64)
  evaluator = FrontierEvaluator(obj_rows, bounds, frontier, depth)
  search = OptimisticTreeSearch(evaluator, len(bounds), max_evals * 1.0, nu_max, rho_max, arity, c_init, tol)
  results, index = search.run_pdoo(mult)
  max_pt = map_to_bounds(results[index][2], bounds)
  history = None
  if return_history:
    history = Namespace(query_points=[map_to_bounds(x, bounds) for x in search.query_points],
                        query_vals=list(search.query_vals), device_calls=evaluator.calls,
                        points_evaluated=evaluator.evaluated, points_requested=evaluator.requested)
  return results[index][0], max_pt, history


def pdoo_maximise(obj, bounds, max_evals, return_history=False):
  """ oper_utils.py:257-271 for a single-point objective `obj(x [d]) -> value`: one callback per
      queried point, as in the reference. """
  rows = lambda pts: np.array([float(np.ravel(obj(pt))[0]) for pt in pts])
  return pdoo_maximise_batched(rows, bounds, max_evals, frontier=0, depth=0, return_history=return_history)


def pdoo_minimise(obj, bounds, max_evals, return_history=False):
  """ oper_utils.py:274-282 """
  max_val, min_pt, history = pdoo_maximise(lambda x: -obj(x), bounds, max_evals, return_history)
  return -max_val, min_pt, history
