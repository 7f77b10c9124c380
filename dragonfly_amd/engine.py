"""Thin Python layer over the C-ABI: one Engine per GPU (dfh_ctx) and the fitted-GP handle.

Everything numeric happens in libdfhip.so; this module only marshals NumPy arrays and maps
status codes to the exceptions the reference raises at the same point.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import (ACQ_EI, ACQ_MEAN, ACQ_PI, ACQ_STD, ACQ_TTEI, ACQ_UCB, GET_ALPHA, GET_K, GET_L,
                   INT32_MIN, KERNEL_ADDITIVE, KERNEL_EXPDECAY, KERNEL_MATERN, KERNEL_POLY, KERNEL_PRODUCT, KERNEL_SE,
                   KernelDesc,
                   check)

ACQ_IDS = {'mean': ACQ_MEAN, 'ucb': ACQ_UCB, 'ei': ACQ_EI, 'pi': ACQ_PI, 'ttei': ACQ_TTEI,
           'std': ACQ_STD}


def _f64(a):
  return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
  """ void* of a NumPy array, a DeviceArray, or None. """
  if a is None:
    return None
  if isinstance(a, DeviceArray):
    return a.ptr
  if isinstance(a, C.c_void_p):
    return a
  # (the address as an integer: what a.ctypes.data_as(c_void_p) passes, without building the ctypes view -- 1.1 against
  #  2.5 us, five times per call of a tuning objective that a real run evaluates a hundred thousand times)
  return a.__array_interface__['data'][0]


class DeviceArray(object):
  """ A float64 buffer resident in HBM (dfh_malloc). Passed to the C-ABI by pointer. """

  def __init__(self, engine, shape):
    self.engine = engine
    self.shape = tuple(int(s) for s in np.atleast_1d(shape))
    self.size = int(np.prod(self.shape))
    p = C.c_void_p()
    check(engine.lib.dfh_malloc(engine.ctx, self.size * 8, C.byref(p)))
    self.ptr = p

  def upload(self, host):
    host = _f64(host)
    assert host.size == self.size
    check(self.engine.lib.dfh_memcpy_h2d(self.engine.ctx, self.ptr, _ptr(host), self.size * 8))
    return self

  def download(self):
    out = np.empty(self.shape, dtype=np.float64)
    check(self.engine.lib.dfh_memcpy_d2h(self.engine.ctx, _ptr(out), self.ptr, self.size * 8))
    return out

  def slice(self, start, count):
    """ `count` doubles from element `start`, as a host array. """
    out = np.empty(int(count), dtype=np.float64)
    check(self.engine.lib.dfh_memcpy_d2h(self.engine.ctx, _ptr(out), self.offset(start), int(count) * 8))
    return out

  def row(self, i):
    """ Row i of a 2-D buffer as a host array (the winning candidate of a device-side search). """
    return self.slice(int(i) * self.shape[-1], self.shape[-1])

  def offset(self, n_elems):
    """ A raw pointer n_elems doubles into the buffer (no ownership). """
    return C.c_void_p(self.ptr.value + 8 * int(n_elems))

  def view(self, start, shape):
    """ A DeviceArray over the elements [start, start + prod(shape)) of this buffer: no copy, no
        ownership (it keeps this buffer alive; free() on the view does nothing). """
    v = DeviceArray.__new__(DeviceArray)
    v.engine = self.engine
    v.shape = tuple(int(x) for x in np.atleast_1d(shape))
    v.size = int(np.prod(v.shape))
    assert 0 <= int(start) and int(start) + v.size <= self.size
    v.ptr = self.offset(start)
    v._owner = self
    return v

  def free(self):
    if getattr(self, '_owner', None) is not None:      # a view
      self.ptr = None
      self._owner = None
      return
    if self.ptr is not None and self.engine.ctx is not None:
      self.engine.lib.dfh_free(self.engine.ctx, self.ptr)
    self.ptr = None

  def __del__(self):
    try:
      self.free()
    except Exception:     # pylint: disable=broad-except
      pass


_SINGLE_KINDS = {'se': KERNEL_SE, 'matern': KERNEL_MATERN, 'poly': KERNEL_POLY, 'expdecay': KERNEL_EXPDECAY}


_DESC_DTYPE = np.dtype({'names': [f[0] for f in KernelDesc._fields_],
                        'formats': ['<i4' if f[1] is C.c_int32 else ('<f8' if f[1] is C.c_double else '<u8') for f in KernelDesc._fields_],
                        'offsets': [getattr(KernelDesc, f[0]).offset for f in KernelDesc._fields_],
                        'itemsize': C.sizeof(KernelDesc)})


def _single_kind_descs(specs, d, backing):
  """ struct dfh_kernel_desc[len(specs)] for a list of SE / Matern / polynomial / exponential-decay candidates of
      dimension d, filled column-wise (a tuning batch holds thousands: 6 us of ctypes per candidate otherwise);
      None if some candidate has groups or a bandwidth vector of another length (to_desc reports those). """
  nb = len(specs)
  if nb == 0:
    return None
  try:
    kinds = [_SINGLE_KINDS[sp.kind] for sp in specs]
  except KeyError:
    return None
  if any(sp.bandwidths is None or sp.bandwidths.size != d or sp.dim != d for sp in specs):
    return None
  bw = np.empty((nb, d), dtype=np.float64)
  for i, sp in enumerate(specs):
    bw[i] = sp.bandwidths
  arr = np.zeros(nb, dtype=_DESC_DTYPE)
  arr['kind'] = kinds
  arr['dim'] = d
  arr['scale'] = [sp.scale for sp in specs]
  arr['nu'] = [sp.nu for sp in specs]
  base = bw.__array_interface__['data'][0]
  arr['bw'] = range(base, base + 8 * d * nb, 8 * d) if nb < 64 else base + np.arange(nb, dtype=np.uint64) * np.uint64(8 * d)
  backing += [bw, arr]
  return arr.ctypes.data_as(C.POINTER(KernelDesc))


class KernelSpec(object):
  """ Host-side description of a Euclidean kernel, convertible to struct dfh_kernel_desc.
      kind: 'se' | 'matern' | 'poly' (nu = order, bandwidths = dim_scalings) | 'expdecay' (nu =
      offset, bandwidths = powers) | 'additive' | 'product' (coordinate-wise product, kernel.py:541;
      its factors may be any of the four single kinds, an additive kernel's se / matern / poly).
      A product may hold ADDITIVE FACTORS (an AdditiveKernel among its kernels): list the factor's
      groups like any others and give group_factors[g] = index of the factor group g belongs to
      (non-decreasing), factor_sums[f] = True for an additive factor, factor_scales[f] = its scale. """

  def __init__(self, kind, dim, scale, bandwidths=None, nu=0.0, groups=None, sub_kinds=None,
               sub_scales=None, sub_nus=None, sub_bandwidths=None, group_factors=None, factor_sums=None,
               factor_scales=None):
    self.kind = kind
    self.dim = int(dim)
    self.scale = float(scale)
    self.nu = float(nu) if nu is not None else 0.0
    self.bandwidths = None if bandwidths is None else _f64(np.ravel(bandwidths))
    self.groups = groups
    self.sub_kinds = sub_kinds
    self.sub_scales = sub_scales
    self.sub_nus = sub_nus
    self.sub_bandwidths = sub_bandwidths
    self.group_factors, self.factor_sums, self.factor_scales = group_factors, factor_sums, factor_scales
    self._keep = []

  def signature(self):
    """ Hashable value identifying the kernel exactly (used to decide whether a cached posterior
        was built with the same kernel). """
    def _t(a):
      return None if a is None else tuple(np.ravel(np.asarray(a, dtype=float)).tolist())
    groups = None if self.groups is None else tuple(tuple(int(c) for c in g) for g in self.groups)
    subs = None if self.sub_bandwidths is None else tuple(_t(b) for b in self.sub_bandwidths)
    kinds = None if self.sub_kinds is None else tuple(self.sub_kinds)
    return (self.kind, self.dim, self.scale, self.nu, _t(self.bandwidths), groups, kinds,
            _t(self.sub_scales), _t(self.sub_nus), subs, _t(self.group_factors), _t(self.factor_sums),
            _t(self.factor_scales))

  def to_desc(self):
    """ Builds the ctypes struct.  The arrays it points at are kept alive both on the struct
        (d.backing) and on self -- appended, never dropped, so an earlier descriptor of the same
        spec stays valid (a list of specs may hold one object several times). """
    d = KernelDesc()
    first = len(self._keep)
    if self.kind in _SINGLE_KINDS:
      d.kind = _SINGLE_KINDS[self.kind]
      d.dim = self.dim
      d.scale = self.scale
      d.nu = self.nu
      if self.bandwidths is None or self.bandwidths.size != self.dim:
        raise ValueError('Dimension of dim_bandwidths should be the same as dimension.')
      bw = _f64(self.bandwidths)
      self._keep.append(bw)
      d.bw = bw.ctypes.data_as(_lib.c_double_p)
      d.n_groups = 0
    elif self.kind in ('additive', 'product'):
      d.kind = KERNEL_ADDITIVE if self.kind == 'additive' else KERNEL_PRODUCT
      d.dim = self.dim
      d.scale = self.scale
      ng = len(self.groups)
      off = np.zeros(ng + 1, dtype=np.int32)
      off[1:] = np.cumsum([len(g) for g in self.groups])
      dims = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.int32).ravel()
                                                  for g in self.groups]), dtype=np.int32)
      kinds = np.ascontiguousarray([_SINGLE_KINDS[k] for k in self.sub_kinds], dtype=np.int32)
      scales = _f64(self.sub_scales)
      nus = _f64(self.sub_nus if self.sub_nus is not None else np.zeros(ng))
      bws = _f64(np.concatenate([_f64(np.ravel(b)) for b in self.sub_bandwidths]))
      if bws.size != dims.size:
        raise ValueError('Group bandwidths do not match the group dimensions.')
      self._keep += [off, dims, kinds, scales, nus, bws]
      d.n_groups = ng
      d.group_off = off.ctypes.data_as(_lib.c_int32_p)
      d.group_dims = dims.ctypes.data_as(_lib.c_int32_p)
      d.sub_kind = kinds.ctypes.data_as(_lib.c_int32_p)
      d.sub_scale = scales.ctypes.data_as(_lib.c_double_p)
      d.sub_nu = nus.ctypes.data_as(_lib.c_double_p)
      d.sub_bw = bws.ctypes.data_as(_lib.c_double_p)
      if self.group_factors is not None:
        if self.kind != 'product':
          raise ValueError('Additive factors exist in product kernels only.')
        gf = np.ascontiguousarray(self.group_factors, dtype=np.int32)
        fs = np.ascontiguousarray([1 if s else 0 for s in self.factor_sums], dtype=np.int32)
        fsc = _f64(self.factor_scales)
        if gf.size != ng or fs.size != fsc.size or (ng and int(gf.max()) >= fs.size):
          raise ValueError('group_factors / factor_sums / factor_scales do not fit the groups.')
        self._keep += [gf, fs, fsc]
        d.group_factor = gf.ctypes.data_as(_lib.c_int32_p)
        d.factor_is_sum = fs.ctypes.data_as(_lib.c_int32_p)
        d.factor_scale = fsc.ctypes.data_as(_lib.c_double_p)
    else:
      raise ValueError('Unidentified kernel type %s.' % (self.kind))
    d.backing = self._keep[first:]
    if first > 0:
      self._keep = self._keep[first:]       # older arrays live on with the descriptors that use them
    return d


class Engine(object):
  """ One MI355X: a dfh_ctx (stream + workspaces). """

  def __init__(self, device=None):
    self.lib = _lib.load()
    self.ctx = None
    if device is None:
      device = int(os.environ.get('DFH_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    if _lib.device_count() == 0:
      raise _lib.DfhipError('No HIP device visible: dragonfly_amd needs an MI355X (gfx950); '
                            'there is no CPU fallback.')
    ctx = C.c_void_p()
    check(self.lib.dfh_ctx_create(int(device), C.byref(ctx)))
    self.ctx = ctx
    self.device = int(device)
    self._owns_ctx = True

  @classmethod
  def from_ctx(cls, ctx, device):
    """ A view of a context somebody else owns (parallel.MultiEngine's per-device contexts). """
    new = cls.__new__(cls)
    new.lib, new.ctx, new.device, new._owns_ctx = _lib.load(), ctx, int(device), False
    return new

  def close(self):
    if self.ctx is not None and getattr(self, '_owns_ctx', True):
      self.lib.dfh_ctx_destroy(self.ctx)
    self.ctx = None

  def __del__(self):
    try:
      self.close()
    except Exception:     # pylint: disable=broad-except
      pass

  # -- plumbing --------------------------------------------------------------------------
  def name(self):
    buf = C.create_string_buffer(256)
    check(self.lib.dfh_device_name(self.ctx, buf, 256))
    return buf.value.decode()

  def sync(self):
    check(self.lib.dfh_sync(self.ctx))

  def to_device(self, host):
    host = _f64(host)
    return DeviceArray(self, host.shape).upload(host)

  def empty(self, shape):
    return DeviceArray(self, shape)

  def random_candidates(self, num, dim, bounds=None, rng=None, out=None, rows=None):
    """ `num` uniform points of the box `bounds` ([dim][2]; None = unit cube) generated in HBM:
        the draw np.random.random((num, dim)) mapped to the bounds (oper_utils.py:62,
        general_utils.py:25-27), bit for bit, from the state of `rng` -- None / the np.random module
        (the global legacy state the reference uses), a np.random.RandomState, or a
        np.random.Generator over a Philox bit generator.  The generator is advanced exactly as
        the host draw would have advanced it.  Returns a DeviceArray [num x dim] (or fills `out`,
        a DeviceArray or a host array).  rows=(begin, count) keeps only that slice of the rows (the
        result is [count x dim]) while the generator still advances over all `num` rows: the shard
        of one rank in a multi-GPU run. """
    num, dim = int(num), int(dim)
    if num < 0 or dim < 1:
      raise ValueError('random_candidates needs num >= 0 and dim >= 1.')
    if bounds is not None:
      bounds = _f64(bounds)
      if bounds.shape != (dim, 2):
        raise ValueError('bounds must have shape (dim, 2).')
    row_begin, row_count = (0, num) if rows is None else (int(rows[0]), int(rows[1]))
    if row_begin < 0 or row_count < 0 or row_begin + row_count > num:
      raise ValueError('rows must lie inside [0, num).')
    if out is None:
      out = DeviceArray(self, (max(row_count, 1), dim))
      out.shape, out.size = (row_count, dim), row_count * dim
    if rng is None or rng is np.random or isinstance(rng, np.random.RandomState):
      legacy = np.random if (rng is None or rng is np.random) else rng
      state = legacy.get_state()
      if state[0] != 'MT19937':
        raise ValueError('The legacy NumPy state is not MT19937.')
      key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
      pos = C.c_int32(int(state[2]))
      check(self.lib.dfh_rand_mt19937_uniform(self.ctx, _ptr(key), C.byref(pos), num, dim, row_begin,
                                              row_count, _ptr(bounds), _ptr(out)))
      legacy.set_state((state[0], key, int(pos.value)) + tuple(state[3:]))
      return out
    bit_gen = getattr(rng, 'bit_generator', rng)
    if not isinstance(bit_gen, np.random.Philox):
      raise ValueError('Device candidate generation follows MT19937 (np.random / RandomState) or '
                       'Philox (Generator(Philox(...))) streams; got %s.' % (type(bit_gen).__name__))
    state = bit_gen.state
    if state['has_uint32']:
      raise ValueError('The Philox generator holds half a word from a 32-bit draw.')
    key = np.ascontiguousarray(state['state']['key'], dtype=np.uint64).copy()
    counter = np.ascontiguousarray(state['state']['counter'], dtype=np.uint64).copy()
    held = np.ascontiguousarray(state['buffer'], dtype=np.uint64).copy()
    held_pos = C.c_int32(int(state['buffer_pos']))
    check(self.lib.dfh_rand_philox_uniform(self.ctx, _ptr(key), _ptr(counter), _ptr(held),
                                           C.byref(held_pos), num, dim, row_begin, row_count,
                                           _ptr(bounds), _ptr(out)))
    state['state']['counter'] = counter
    state['buffer'] = held
    state['buffer_pos'] = int(held_pos.value)
    bit_gen.state = state
    return out

  def random_normals(self, num, rng=None, out=None):
    """ np.random.normal(size=num) -- the standard normals of draw_gaussian_samples
        (general_utils.py:230) -- generated in HBM from the legacy MT19937 state of `rng` (None / the
        np.random module: the global state the reference uses; or a RandomState), bit for bit, the
        cached second gaussian included; the generator is advanced exactly as the host draw would
        have advanced it.  Returns a DeviceArray [num] (or fills `out`). """
    num = int(num)
    if num < 0:
      raise ValueError('random_normals needs num >= 0.')
    legacy = np.random if (rng is None or rng is np.random) else rng
    if not (legacy is np.random or isinstance(legacy, np.random.RandomState)):
      raise ValueError('Device normals follow the legacy MT19937 stream (np.random / RandomState).')
    if out is None:
      out = DeviceArray(self, (max(num, 1),))
      out.shape, out.size = (num,), num
    state = legacy.get_state()
    if state[0] != 'MT19937':
      raise ValueError('The legacy NumPy state is not MT19937.')
    key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
    pos, has_gauss, gauss = C.c_int32(int(state[2])), C.c_int32(int(state[3])), C.c_double(float(state[4]))
    check(self.lib.dfh_rand_mt19937_normal(self.ctx, _ptr(key), C.byref(pos), C.byref(has_gauss), C.byref(gauss),
                                           num, _ptr(out)))
    legacy.set_state((state[0], key, int(pos.value), int(has_gauss.value), float(gauss.value)))
    return out

  def mem_info(self):
    """ (free, total) HBM bytes of this engine's device. """
    f, t = C.c_uint64(0), C.c_uint64(0)
    check(self.lib.dfh_mem_info(self.ctx, C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)

  def timer_begin(self):
    check(self.lib.dfh_timer_begin(self.ctx))

  def timer_end(self):
    ms = C.c_double(0)
    check(self.lib.dfh_timer_end(self.ctx, C.byref(ms)))
    return ms.value

  def timings(self, enable=True):
    """ Returns the section timings (ms) accumulated since the last call and resets them. """
    arr = (C.c_double * 8)()
    check(self.lib.dfh_ctx_timings(self.ctx, 1 if enable else 0, arr))
    return dict(zip(_lib.T_NAMES, list(arr)))

  def counters(self):
    """ {'chol_fallbacks': factorisations repeated on the hand-off-free schedule, 'chol_cooldown': how many
        of the next ones skip the hand-off schedules} (dfh_ctx_counters) """
    arr = (C.c_int64 * 4)()
    check(self.lib.dfh_ctx_counters(self.ctx, arr))
    return {'chol_fallbacks': int(arr[0]), 'chol_cooldown': int(arr[1])}

  def gemm_profile(self, enable=True, fetch=True):
    """ Per-variant {launches, ms (sum of launch durations), flop, busy_ms (union of the launch
        intervals), bytes (algorithmic)} of the GEMM kernel since the last call (HIP events). """
    arr = (C.c_double * 40)()
    check(self.lib.dfh_ctx_gemm_profile(self.ctx, 1 if enable else 0, arr if fetch else None))
    return [dict(launches=int(arr[5 * v]), ms=arr[5 * v + 1], flop=arr[5 * v + 2], busy_ms=arr[5 * v + 3],
                 bytes=arr[5 * v + 4]) for v in range(8)]

  # -- building blocks ---------------------------------------------------------------------
  def kernel_matrix(self, spec, X1, X2=None, diag_add=0.0, out=None):
    X1h = X1 if isinstance(X1, DeviceArray) else _f64(X1)
    n1 = X1h.shape[0]
    if X2 is None:
      X2h, n2 = None, n1
    else:
      X2h = X2 if isinstance(X2, DeviceArray) else _f64(X2)
      n2 = X2h.shape[0]
    if out is None:
      out = np.zeros((n1, n2), dtype=np.float64)
    if n1 == 0 or n2 == 0:
      return out
    desc = spec.to_desc()
    check(self.lib.dfh_kernel_matrix(self.ctx, C.byref(desc), _ptr(X1h), n1, _ptr(X2h), n2,
                                     float(diag_add), _ptr(out)))
    return out

  def dist_squared(self, X1, X2):
    X1 = _f64(X1)
    X2 = _f64(X2)
    n1, d1 = X1.shape
    n2, d2 = X2.shape
    if d1 != d2:
      raise ValueError('Second dimension of X1 and X2 should be equal.')
    out = np.zeros((n1, n2), dtype=np.float64)
    if n1 == 0 or n2 == 0:
      return out
    check(self.lib.dfh_dist_squared(self.ctx, _ptr(X1), n1, _ptr(X2), n2, d1, _ptr(out)))
    return out

  def gemm(self, A, B, C_in=None, alpha=1.0, beta=0.0, transb=False, lower_only=False,
           shape=None, out=None):
    """ out = beta*C + alpha*A*op(B). Host arrays or (with shape=(M,N,K)) DeviceArrays. """
    if shape is None:
      A = _f64(A)
      B = _f64(B)
      M, K = A.shape
      N = B.shape[1] if transb else B.shape[0]
      if out is None:
        out = np.zeros((M, N)) if C_in is None else _f64(C_in).copy()
      lda, ldb, ldc = K, (N if transb else K), N
    else:
      M, N, K = shape
      lda, ldb, ldc = K, (N if transb else K), N
      assert out is not None
    check(self.lib.dfh_gemm(self.ctx, 1 if transb else 0, M, N, K, float(alpha), _ptr(A), lda,
                            _ptr(B), ldb, float(beta), _ptr(out), ldc, 1 if lower_only else 0))
    return out

  def cholesky(self, M):
    """ numpy.linalg.cholesky semantics (raises LinAlgError when not positive definite). """
    if isinstance(M, DeviceArray):
      n = M.shape[0]
      piv = C.c_int64(0)
      check(self.lib.dfh_cholesky(self.ctx, M.ptr, n, C.byref(piv)))
      return M
    L = _f64(M).copy()
    n = L.shape[0]
    piv = C.c_int64(0)
    check(self.lib.dfh_cholesky(self.ctx, _ptr(L), n, C.byref(piv)))
    return L

  def stable_cholesky(self, M, return_power=False):
    M = _f64(M)
    n = M.shape[0]
    L = np.empty_like(M)
    jp = C.c_int32(INT32_MIN)
    check(self.lib.dfh_stable_cholesky(self.ctx, _ptr(M), n, _ptr(L), C.byref(jp)))
    if return_power:
      return L, (None if jp.value == INT32_MIN else jp.value)
    return L

  def solve_triangular(self, L_lower, b, upper=False):
    """ Solves L x = b (upper=False) or L^T x = b (upper=True) for lower-triangular L. """
    L_lower = _f64(L_lower)
    b = _f64(b)
    n = L_lower.shape[0]
    nrhs = 1 if b.ndim == 1 else b.shape[1]
    x = np.empty_like(b)
    check(self.lib.dfh_solve_triangular(self.ctx, _ptr(L_lower), n, 1 if upper else 0, _ptr(b),
                                        nrhs, _ptr(x)))
    return x

  # -- GP ----------------------------------------------------------------------------------
  def gp_lml_batch(self, specs, X, y, mean_consts, noise_vars, allow_jitter=True, return_powers=False):
    """ Log marginal likelihoods of len(specs) hyper-parameter candidates on the same (X, y): the
        tuning objective of GPFitter (gp_core.py:551-564) for a list of candidates, fitted in
        lock-step groups on the device.  y holds raw labels; mean_consts[c] is subtracted. """
    nb = len(specs)
    Xh = X if isinstance(X, DeviceArray) else _f64(X)
    yh = y if isinstance(y, DeviceArray) else _f64(y)
    n, d = Xh.shape
    backing = []
    descs = _single_kind_descs(specs, d, backing)      # one vectorised fill when every candidate is a plain kernel
    if descs is None:
      descs = (_lib.KernelDesc * max(nb, 1))()
      for i, sp in enumerate(specs):
        one = sp.to_desc()
        backing.append(one.backing)       # the arrays the copied struct points at
        descs[i] = one
    mc = _f64(np.zeros(nb) if mean_consts is None else mean_consts).reshape(-1)
    nv = _f64(noise_vars).reshape(-1)
    if len(mc) != nb or len(nv) != nb:
      raise ValueError('gp_lml_batch: need one mean constant and one noise variance per candidate.')
    lml = np.empty(nb, dtype=np.float64)
    jps = np.empty(nb, dtype=np.int32)
    flags = 0 if allow_jitter else _lib.FIT_NO_JITTER
    if isinstance(Xh, DeviceArray):
      flags |= _lib.LML_X_IS_DEVICE
    if not isinstance(yh, DeviceArray):
      flags |= _lib.LML_Y_IS_HOST
    check(self.lib.dfh_gp_lml_batch(self.ctx, descs, nb, _ptr(Xh), n, d, _ptr(yh), _ptr(mc), _ptr(nv),
                                    flags, _ptr(lml), _ptr(jps)))
    if return_powers:
      return lml, [None if p == INT32_MIN else int(p) for p in jps]
    return lml

  def gp_fit_gram(self, K, y_centred, noise_var, allow_jitter=True, handle_non_psd_kernels='guaranteed_psd'):
    """ Posterior from a Gram matrix the caller evaluated: a FittedGP without a kernel -- use
        predict_gram / predict_covar_gram with it.  handle_non_psd_kernels as in the reference's
        _get_cholesky_decomp (gp_core.py:827-847): 'guaranteed_psd' | 'project_first' |
        'try_before_project'. """
    return FittedGP.from_gram(self, K, y_centred, noise_var, allow_jitter, handle_non_psd_kernels)

  def project_psd(self, M, epsilon=0.0):
    """ project_symmetric_to_psd_cone (general_utils.py:150-163) on the device. """
    M = _f64(M)
    n = M.shape[0]
    if M.shape != (n, n):
      raise ValueError('project_psd: the matrix must be square.')
    out = np.empty_like(M)
    check(self.lib.dfh_project_psd(self.ctx, _ptr(M), n, float(epsilon), _ptr(out)))
    return out

  def gp_fit(self, spec, X, y_centred, noise_var, allow_jitter=True):
    """ Returns a FittedGP (posterior resident in HBM). """
    return FittedGP(self, spec, X, y_centred, noise_var, allow_jitter)


class FittedGP(object):
  """ Handle of a dfh_gp: K, L, alpha live on the device. """

  def __init__(self, engine, spec, X, y_centred, noise_var, allow_jitter=True):
    self.engine = engine
    self.handle = None
    self.spec = spec
    Xh = X if isinstance(X, DeviceArray) else _f64(X)
    yh = y_centred if isinstance(y_centred, DeviceArray) else _f64(y_centred)
    n, d = Xh.shape
    self.n, self.d = int(n), int(d)
    desc = spec.to_desc()
    h = C.c_void_p()
    lml = C.c_double(0)
    jp = C.c_int32(INT32_MIN)
    check(engine.lib.dfh_gp_fit(engine.ctx, C.byref(desc), _ptr(Xh), n, d, _ptr(yh),
                                float(noise_var), 0 if allow_jitter else _lib.FIT_NO_JITTER,
                                C.byref(h), C.byref(lml), C.byref(jp)))
    self.handle = h
    self.lml = lml.value
    self.jitter_power = None if jp.value == INT32_MIN else jp.value

  @classmethod
  def from_gram(cls, engine, K, y_centred, noise_var, allow_jitter=True, handle_non_psd_kernels='guaranteed_psd'):
    Kh = K if isinstance(K, DeviceArray) else _f64(K)
    n = Kh.shape[0]
    if Kh.shape != (n, n):
      raise ValueError('from_gram: the Gram matrix must be square.')
    yh = _f64(y_centred).reshape(-1)
    if yh.shape[0] != n:
      raise ValueError('from_gram: need %d centred labels, got %d.' % (n, yh.shape[0]))
    h = C.c_void_p()
    lml = C.c_double(0)
    jp = C.c_int32(INT32_MIN)
    if handle_non_psd_kernels not in ('guaranteed_psd', 'project_first', 'try_before_project'):
      raise ValueError('Unknown option for handle_non_psd_kernels: %s' % (handle_non_psd_kernels))
    flags = (0 if allow_jitter else _lib.FIT_NO_JITTER) | \
        {'guaranteed_psd': 0, 'project_first': _lib.FIT_PROJECT_FIRST,
         'try_before_project': _lib.FIT_TRY_BEFORE_PROJECT}[handle_non_psd_kernels]
    check(engine.lib.dfh_gp_fit_gram(engine.ctx, _ptr(Kh), n, _ptr(yh), float(noise_var), flags, C.byref(h),
                                     C.byref(lml), C.byref(jp)))
    new = cls.__new__(cls)
    new.engine, new.spec, new.handle = engine, None, h
    new.n, new.d = int(n), 0
    new.lml = lml.value
    new.jitter_power = None if jp.value == INT32_MIN else jp.value
    return new

  def predict_gram(self, K_cross, k_ss=None, mean_const=0.0, mean_vals=None):
    """ (mu, sd) from the caller's cross matrix K(X*, X) [m x n] and prior variances k(x*, x*) [m]
        (k_ss None: mean only). """
    Kc = _f64(K_cross)
    m = Kc.shape[0]
    if Kc.ndim != 2 or Kc.shape[1] != self.n:
      raise ValueError('predict_gram: the cross matrix must be m x %d.' % (self.n))
    mu = np.empty(m, dtype=np.float64)
    sd = np.empty(m, dtype=np.float64) if k_ss is not None else None
    ks = None if k_ss is None else _f64(k_ss).reshape(-1)
    mv = None if mean_vals is None else _f64(mean_vals).reshape(-1)
    check(self.engine.lib.dfh_gp_predict_gram(self.handle, _ptr(Kc), m, _ptr(ks), float(mean_const),
                                              _ptr(mv), _ptr(mu), _ptr(sd)))
    return mu, sd

  def predict_covar_gram(self, K_cross, K_tete):
    """ (K_cross alpha, K_tete - V^T V) for the caller's kernel matrices. """
    Kc, Kt = _f64(K_cross), _f64(K_tete)
    m = Kc.shape[0]
    if Kc.shape != (m, self.n) or Kt.shape != (m, m):
      raise ValueError('predict_covar_gram: need an m x %d cross matrix and an m x m test matrix.' % (self.n))
    mu = np.empty(m, dtype=np.float64)
    cov = np.empty((m, m), dtype=np.float64)
    check(self.engine.lib.dfh_gp_predict_covar_gram(self.handle, _ptr(Kc), m, _ptr(Kt), _ptr(mu), _ptr(cov)))
    return mu, cov

  def append(self, X_new, y_centred_all, allow_jitter=True):
    """ Posterior extended by the rows of X_new (dfh_gp_append): a NEW FittedGP, this one stays
        valid.  y_centred_all: all n+q centred labels, old observations first. """
    Xn = _f64(X_new)
    if Xn.ndim == 1:
      Xn = Xn.reshape(1, -1)
    q = Xn.shape[0]
    if Xn.shape[1] != self.d:
      raise ValueError('append: new points have %d columns, the GP has %d.' % (Xn.shape[1], self.d))
    yh = _f64(y_centred_all).reshape(-1)
    if yh.shape[0] != self.n + q:
      raise ValueError('append: need %d centred labels, got %d.' % (self.n + q, yh.shape[0]))
    h = C.c_void_p()
    lml = C.c_double(0)
    jp = C.c_int32(INT32_MIN)
    check(self.engine.lib.dfh_gp_append(self.handle, _ptr(Xn), q, _ptr(yh),
                                        0 if allow_jitter else _lib.FIT_NO_JITTER,
                                        C.byref(h), C.byref(lml), C.byref(jp)))
    new = FittedGP.__new__(FittedGP)
    new.engine, new.spec, new.handle = self.engine, self.spec, h
    new.n, new.d = self.n + q, self.d
    new.lml = lml.value
    new.jitter_power = None if jp.value == INT32_MIN else jp.value
    return new

  def free(self):
    if self.handle is not None:
      self.engine.lib.dfh_gp_free(self.handle)
      self.handle = None

  def __del__(self):
    try:
      self.free()
    except Exception:     # pylint: disable=broad-except
      pass

  def _get(self, what, shape):
    out = np.empty(shape, dtype=np.float64)
    check(self.engine.lib.dfh_gp_get(self.handle, what, _ptr(out)))
    return out

  def refine_steps(self):
    """ Iterative-refinement steps the triangular solves take per 512-block of the factor
        (all zero unless a diagonal block is ill-conditioned; dfh_gp_refine_steps). """
    nblk = (self.n + 511) // 512
    out = (C.c_int32 * nblk)()
    check(self.engine.lib.dfh_gp_refine_steps(self.handle, out))
    return list(out)

  def get_L(self):
    return self._get(GET_L, (self.n, self.n))

  def get_alpha(self):
    return self._get(GET_ALPHA, (self.n,))

  def get_K(self):
    return self._get(GET_K, (self.n, self.n))

  @staticmethod
  def _rows(Xs):
    if isinstance(Xs, DeviceArray):
      return Xs, Xs.shape[0]
    Xs = _f64(Xs)
    return Xs, Xs.shape[0]

  def predict(self, Xs, want_std=True, X_halluc=None):
    """ (K(Xs,X) alpha, posterior std) -- the caller adds the mean function. """
    Xs, m = self._rows(Xs)
    mu = np.empty(m)
    sd = np.empty(m) if want_std else None
    Xh, q = (None, 0) if X_halluc is None or len(X_halluc) == 0 else self._rows(X_halluc)
    check(self.engine.lib.dfh_gp_predict(self.handle, _ptr(Xs), m, _ptr(Xh), q, _ptr(mu), _ptr(sd)))
    return mu, sd

  def predict_covar(self, Xs, X_halluc=None):
    Xs, m = self._rows(Xs)
    mu = np.empty(m)
    cov = np.empty((m, m))
    Xh, q = (None, 0) if X_halluc is None or len(X_halluc) == 0 else self._rows(X_halluc)
    check(self.engine.lib.dfh_gp_predict_covar(self.handle, _ptr(Xs), m, _ptr(Xh), q, _ptr(mu),
                                               _ptr(cov)))
    return mu, cov

  def acq_argmax(self, acq, Xs, params=(0.0, 0.0), mean_const=0.0, mean_vals=None, X_halluc=None,
                 return_vals=False):
    """ Fused posterior + acquisition + arg-max. Returns (best_val, best_idx[, vals]). """
    Xs, m = self._rows(Xs)
    p = (C.c_double * 2)(float(params[0]), float(params[1]) if len(params) > 1 else 0.0)
    vals = np.empty(m) if return_vals else None
    mv = None if mean_vals is None else (mean_vals if isinstance(mean_vals, DeviceArray)
                                         else _f64(mean_vals))
    Xh, q = (None, 0) if X_halluc is None or len(X_halluc) == 0 else self._rows(X_halluc)
    bv = C.c_double(0)
    bi = C.c_int64(-1)
    check(self.engine.lib.dfh_gp_acq_argmax(self.handle, ACQ_IDS[acq], p, _ptr(Xs), m, _ptr(Xh), q,
                                            float(mean_const), _ptr(mv), _ptr(vals), C.byref(bv),
                                            C.byref(bi)))
    if return_vals:
      return bv.value, bi.value, vals
    return bv.value, bi.value

  def thompson(self, Xs, U, block=4096, mean_const=0.0, mean_vals=None, return_samples=False):
    """ Blocked-joint Thompson sample over the candidates. Returns (best_val, best_idx[, samples,
        jitter_powers]). """
    Xs, m = self._rows(Xs)
    Uh = U if isinstance(U, DeviceArray) else _f64(np.ravel(U))
    block = int(min(block, m))
    nblk = (m + block - 1) // block
    samples = np.empty(m) if return_samples else None
    jps = (C.c_int32 * nblk)()
    mv = None if mean_vals is None else (mean_vals if isinstance(mean_vals, DeviceArray)
                                         else _f64(mean_vals))
    bv = C.c_double(0)
    bi = C.c_int64(-1)
    check(self.engine.lib.dfh_gp_ts(self.handle, _ptr(Xs), m, block, _ptr(Uh), float(mean_const),
                                    _ptr(mv), _ptr(samples), C.byref(bv), C.byref(bi), jps))
    if return_samples:
      return bv.value, bi.value, samples, [None if j == INT32_MIN else j for j in jps]
    return bv.value, bi.value

  def add_ucb_group(self, group, beta, Xg, return_vals=False):
    Xg, m = self._rows(Xg)
    vals = np.empty(m) if return_vals else None
    bv = C.c_double(0)
    bi = C.c_int64(-1)
    check(self.engine.lib.dfh_gp_add_ucb_group(self.handle, int(group), float(beta), _ptr(Xg), m,
                                               _ptr(vals), C.byref(bv), C.byref(bi)))
    if return_vals:
      return bv.value, bi.value, vals
    return bv.value, bi.value

  def add_ucb_all(self, betas, cands_per_group, return_vals=False, sizes=None):
    """ add-UCB for every group of the additive kernel in one device call (one posterior solve for
        all groups).  cands_per_group[g]: [m_g x |group g|] -- or ONE DeviceArray holding the groups'
        candidate blocks back to back, with sizes[g] = m_g.  Returns (best_vals, best_idx[, vals]). """
    if isinstance(cands_per_group, DeviceArray):
      flat = cands_per_group
      ms = np.ascontiguousarray(sizes, dtype=np.int64)
    else:
      blocks = [_f64(c) for c in cands_per_group]
      ms = np.ascontiguousarray([b.shape[0] for b in blocks], dtype=np.int64)
      flat = np.ascontiguousarray(np.concatenate([b.ravel() for b in blocks]), dtype=np.float64)
    G = len(ms)
    be = _f64(np.asarray(betas, dtype=float).reshape(-1))
    if len(be) != G:
      raise ValueError('add_ucb_all: need one beta per group.')
    bv = np.empty(G, dtype=np.float64)
    bi = np.empty(G, dtype=np.int64)
    vals = np.empty(int(ms.sum()), dtype=np.float64) if return_vals else None
    check(self.engine.lib.dfh_gp_add_ucb_all(self.handle, _ptr(be), _ptr(flat), _ptr(ms), _ptr(vals),
                                             _ptr(bv), _ptr(bi)))
    if return_vals:
      return bv, bi, np.split(vals, np.cumsum(ms)[:-1])
    return bv, bi


_default_engine = None


def get_engine():
  """ The process-wide engine (one process per GPU; device = LOCAL_RANK / DFH_DEVICE). """
  global _default_engine
  if _default_engine is None:
    _default_engine = Engine()
  return _default_engine
