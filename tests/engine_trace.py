"""TEST INFRASTRUCTURE: record and replay what crosses the drop-in boundary during a run of the reference's optimiser.

The reference optimiser (dragonfly/opt/gp_bandit.py:405-421, 490, 647-673; the fitters it builds, gp/euclidean_gp.py:
325-339) reaches the engine through one object: dragonfly_amd.engine.Engine and the FittedGP handles it returns.
oracle/make_golden.py runs the UNMODIFIED reference with dragonfly_amd.install() in the build container, where the
engine is the NumPy stand-in of tests/oracle_engine.py (tests/test_install_end_to_end.py proves that such a run
recommends the reference's own points, bit for bit), and records every call on that object IN ORDER -- method,
arguments, results -- into tests/golden/engine_trace_*.npz.  The MI355X test (tests/test_gpu_engine_traces.py, which
needs no Dragonfly checkout and therefore runs on the driver's box) replays each trace against libdfhip.so: the same
calls with the same arguments must give the same results -- values within 1e-10, arg-max indices, jitter powers and
array shapes exactly.  Together: reference == reference-on-stand-in (CPU, here) and stand-in == device, call by call
(GPU), for all 25 configurations of tests/test_gpu_install_end_to_end.py."""
import json

import numpy as np


class _Log(object):
  def __init__(self):
    self.events, self.arrays, self.next_id = [], [], 1
    self._seen = {}

  def array(self, a):
    """ index of the array in the trace's table; equal arrays (the training inputs go into every tuning batch) are stored once """
    a = np.ascontiguousarray(a)
    key = (a.dtype.str, a.shape, a.tobytes())
    if key not in self._seen:
      self.arrays.append(a)
      self._seen[key] = len(self.arrays) - 1
    return self._seen[key]


def _ser(log, v, handles):
  """ JSON-able description of a value; arrays go to log.arrays, engine objects become handle ids """
  from dragonfly_amd.engine import KernelSpec
  if v is None or isinstance(v, (bool, str)):
    return v
  if isinstance(v, (int, np.integer)):
    return {'i': int(v)}
  if isinstance(v, (float, np.floating)):
    return {'f': float(v).hex()}
  if id(v) in handles:
    return {'h': handles[id(v)]}
  if isinstance(v, np.ndarray):
    return {'a': log.array(v)}
  if isinstance(v, KernelSpec):
    fields = {}
    for name in ('kind', 'dim', 'scale', 'nu', 'bandwidths', 'groups', 'sub_kinds', 'sub_scales', 'sub_nus',
                 'sub_bandwidths', 'group_factors', 'factor_sums', 'factor_scales'):
      fields[name] = _ser(log, getattr(v, name, None), handles)
    return {'spec': fields}
  if isinstance(v, (list, tuple)):
    return {'l' if isinstance(v, list) else 't': [_ser(log, x, handles) for x in v]}
  raise TypeError('engine_trace: cannot record a %s' % type(v).__name__)


class _Recorder(object):
  """ Proxy of the stand-in engine or of one of its fitted GPs: forwards everything, logs the calls """

  def __init__(self, log, target, handles, hid):
    self.__dict__.update(_log=log, _target=target, _handles=handles, _hid=hid)

  def __getattr__(self, name):
    attr = getattr(self._target, name)
    if not callable(attr):
      return attr
    log, handles = self._log, self._handles

    def call(*args, **kwargs):
      ev = {'h': self._hid, 'm': name, 'args': [_ser(log, a, handles) for a in args],
            'kwargs': {k: _ser(log, v, handles) for k, v in kwargs.items()}}
      log.events.append(ev)
      real_args = [a._target if isinstance(a, _Recorder) else a for a in args]
      real_kwargs = {k: (v._target if isinstance(v, _Recorder) else v) for k, v in kwargs.items()}
      out = attr(*real_args, **real_kwargs)
      return self._wrap(out, ev)
    return call

  def __setattr__(self, name, value):
    setattr(self._target, name, value)

  def _wrap(self, out, ev):
    from oracle_engine import OracleFittedGP
    log, handles = self._log, self._handles
    if isinstance(out, OracleFittedGP):
      hid = log.next_id
      log.next_id += 1
      proxy = _Recorder(log, out, handles, hid)
      handles[id(proxy)] = hid
      handles[id(out)] = hid
      # (kept alive for the length of the recording: ids must stay unique)
      log.arrays_keepalive = getattr(log, 'arrays_keepalive', []) + [proxy, out]
      ev['out'] = {'new_gp': hid, 'lml': float(out.lml).hex(), 'jitter_power': out.jitter_power, 'n': int(out.n)}
      return proxy
    ev['out'] = _ser(log, out, handles)
    return out


def recording_engine():
  """ (engine proxy, log): install it with oracle_engine.patch_engine's monkeypatching """
  from oracle_engine import OracleEngine
  log = _Log()
  handles = {}
  return _Recorder(log, OracleEngine(), handles, 0), log


def save(path, log, meta):
  """ One JSON blob (events + the table of arrays: dtype, shape, offset) and one flat buffer per dtype: a trace holds
      tens of thousands of short vectors (bandwidths of tuning candidates), which as npz members of their own would
      cost more in zip headers than in data. """
  flat, table = {}, []
  for a in log.arrays:
    kind = 'f8' if a.dtype.kind == 'f' else ('i8' if a.dtype.kind in 'iub' else None)
    if kind is None:
      raise TypeError('engine_trace: array of dtype %s' % a.dtype)
    buf = flat.setdefault(kind, [])
    table.append([kind, list(a.shape), int(sum(len(b) for b in buf)), a.dtype.str])
    buf.append(np.ravel(a).astype(np.float64 if kind == 'f8' else np.int64))
  blob = np.frombuffer(json.dumps({'events': log.events, 'meta': meta, 'arrays': table}, separators=(',', ':')).encode('utf-8'),
                       dtype=np.uint8)
  np.savez_compressed(path, trace_json=blob, **{'flat_' + k: (np.concatenate(v) if v else np.zeros(0)) for k, v in flat.items()})


def load(path):
  with np.load(path) as g:
    rec = json.loads(bytes(g['trace_json']).decode('utf-8'))
    flat = {k[5:]: g[k] for k in g.files if k.startswith('flat_')}
  arrays = []
  for kind, shape, off, dtype in rec['arrays']:
    count = int(np.prod(shape)) if len(shape) else 1
    arrays.append(flat[kind][off:off + count].reshape(shape).astype(np.dtype(dtype)))
  return rec, arrays


# ---- replay ----------------------------------------------------------------------------------------------------
def _de(v, arrays, objs):
  from dragonfly_amd.engine import KernelSpec
  if v is None or isinstance(v, (bool, str)):
    return v
  if 'i' in v:
    return v['i']
  if 'f' in v:
    return float.fromhex(v['f'])
  if 'h' in v:
    return objs[v['h']]
  if 'a' in v:
    return arrays[v['a']]
  if 'l' in v:
    return [_de(x, arrays, objs) for x in v['l']]
  if 't' in v:
    return tuple(_de(x, arrays, objs) for x in v['t'])
  if 'spec' in v:
    f = {k: _de(x, arrays, objs) for k, x in v['spec'].items()}
    return KernelSpec(f['kind'], f['dim'], f['scale'], f['bandwidths'], nu=f['nu'], groups=f['groups'], sub_kinds=f['sub_kinds'],
                      sub_scales=f['sub_scales'], sub_nus=f['sub_nus'], sub_bandwidths=f['sub_bandwidths'],
                      group_factors=f['group_factors'], factor_sums=f['factor_sums'], factor_scales=f['factor_scales'])
  raise ValueError('engine_trace: unknown record %r' % (v,))


def _rel(a, b):
  a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
  if a.shape != b.shape:
    return np.inf
  if a.size == 0:
    return 0.0
  if not np.array_equal(np.isfinite(a), np.isfinite(b)):
    return np.inf
  ok = np.isfinite(b)
  if not ok.any():
    return 0.0
  den = float(np.max(np.abs(b[ok])))
  return float(np.max(np.abs(a[ok] - b[ok])) / (den if den > 0 else 1.0))


def _compare(got, want, arrays, tol, where, worst):
  """ want: the recorded description; got: the live value """
  if want is None or isinstance(want, (bool, str)):
    assert got == want or (want is None and got is None), (where, got, want)
  elif 'i' in want:
    assert int(got) == want['i'], (where, got, want['i'])
  elif 'f' in want:
    r = _rel([got], [float.fromhex(want['f'])])
    worst[0] = max(worst[0], r)
    assert r <= tol, (where, got, float.fromhex(want['f']), r)
  elif 'a' in want:
    r = _rel(got, arrays[want['a']])
    worst[0] = max(worst[0], r)
    assert r <= tol, (where, 'array', r)
  elif 'l' in want or 't' in want:
    items = want.get('l', want.get('t'))
    assert len(got) == len(items), (where, len(got), len(items))
    for k, (g, w) in enumerate(zip(got, items)):
      _compare(g, w, arrays, tol, where + '[%d]' % k, worst)
  else:
    raise ValueError('engine_trace: cannot compare %r' % (want,))


def _thompson_by_truth(gp, fit, args, kwargs, where):
  """ The device's joint draw against the extended-precision draw (oracle/ld_truth.c), bound = twice the NumPy
      stand-in's distance from it (tests/truth_bounds.py): from the kernel for single SE / Matern kernels
      (kernel_draw_bound), from the stand-in's own posterior mean and covariance for every other kernel (draw_bound).
      Returns the device's relative error. """
  from oracle_engine import OracleFittedGP
  from truth_bounds import draw_bound, kernel_draw_bound
  from conftest import relerr
  spec, X, yc, noise = fit
  Xs, U = np.asarray(args[0], dtype=float), np.ravel(np.asarray(args[1], dtype=float))
  block = int(kwargs.get('block', 4096))
  og = OracleFittedGP(None, spec, X, yc, noise).oracle
  _, _, samples, powers = gp.thompson(Xs, U, block=block, return_samples=True)
  dev = np.asarray(samples, dtype=float)          # (drawn without the prior-mean shift: zero-mean GP on both sides)
  ref = og.draw_samples_blocked(Xs, U, block)
  worst = 0.0
  for b0 in range(0, len(Xs), block):
    sl = slice(b0, min(len(Xs), b0 + block))
    mu_b, cov_b = og.eval(Xs[sl], 'covar')
    if spec.kind in ('se', 'matern'):
      nu = float(spec.nu) if spec.kind == 'matern' else 0.0
      bound = kernel_draw_bound(spec.kind, nu, spec.bandwidths, spec.scale, X, yc, noise, Xs[sl], 0.0, U[sl], ref[sl], cov_b)
      err = relerr(dev[sl], ref[sl])
    else:
      # No kernel-level truth for this kernel.  The draw through a numerically singular covariance amplifies the
      # 1e-13 by which two correct covariances differ, so the statement is split: the device's posterior mean and
      # covariance of the block agree with the stand-in's to 1e-10 (inputs), its ladder settles on the same jitter,
      # and GIVEN ITS OWN covariance its draw is as close to the extended-precision draw as the stand-in's is given
      # its own (twice that distance: tests/truth_bounds.py).
      from oracle import ref_longdouble as T
      from oracle import ref_numpy as O
      mu_d, cov_d = gp.predict_covar(Xs[sl])
      assert relerr(mu_d, mu_b) <= 1e-10 and relerr(cov_d, cov_b) <= 1e-10, (where, 'block posterior', relerr(mu_d, mu_b), relerr(cov_d, cov_b))
      _, pw = O.stable_cholesky(np.asarray(cov_b), return_power=True)
      assert powers is None or powers[b0 // block] == pw, (where, 'jitter power', powers, pw)
      jit = 0.0 if pw is None else (10.0 ** pw) * float(np.diag(cov_b).max())
      bound = draw_bound(mu_b, cov_b, U[sl], ref[sl])
      err = relerr(dev[sl], T.gaussian_draw(mu_d, cov_d, U[sl], jit))
    assert err <= bound, (where, 'joint draw', err, bound)
    worst = max(worst, err)
  return worst


def replay(path, engine, tol=1e-10):
  """ Replays the trace on `engine`; returns (calls replayed, largest relative difference seen). """
  rec, arrays = load(path)
  objs = {0: engine}
  worst = [0.0]
  made = []
  fits = {}                                          # handle id -> (spec, X, y_centred, noise) of fits with a kernel description
  try:
    for k, ev in enumerate(rec['events']):
      target = objs[ev['h']]
      args = [_de(a, arrays, objs) for a in ev['args']]
      kwargs = {name: _de(v, arrays, objs) for name, v in ev['kwargs'].items()}
      where = 'event %d: %s.%s' % (k, 'engine' if ev['h'] == 0 else 'gp%d' % ev['h'], ev['m'])
      out = getattr(target, ev['m'])(*args, **kwargs)
      want = ev.get('out')
      if isinstance(want, dict) and 'new_gp' in want:
        objs[want['new_gp']] = out
        made.append(out)
        if ev['m'] == 'gp_fit':
          fits[want['new_gp']] = (args[0], np.asarray(args[1], dtype=float), np.asarray(args[2], dtype=float), float(args[3]))
        elif ev['m'] == 'append' and ev['h'] in fits:      # FittedGP.append(X_new, y_centred_all)
          spec0, X0, _, noise0 = fits[ev['h']]
          fits[want['new_gp']] = (spec0, np.vstack([X0, np.asarray(args[0], dtype=float)]), np.asarray(args[1], dtype=float), noise0)
        assert out.jitter_power == want['jitter_power'], (where, out.jitter_power, want['jitter_power'])
        r = _rel([out.lml], [float.fromhex(want['lml'])])
        worst[0] = max(worst[0], r)
        assert r <= tol and int(out.n) == want['n'], (where, out.lml, float.fromhex(want['lml']), r)
      elif ev['m'] == 'to_device':
        # (the stand-in hands the host array back and the mirrors pass THAT on, so later calls carry the data itself;
        #  here the upload is exercised and released)
        if hasattr(out, 'free'):
          out.free()
      elif ev['m'] == 'free':
        pass
      elif ev['m'] == 'stable_cholesky' and kwargs.get('return_power') and isinstance(want, dict) and 't' in want:
        # (general_utils.py:166-204 on a matrix the caller built -- the multi-fidelity draws' covariances, numerically
        #  singular: two correct factors of such a matrix differ by cond x eps in their ENTRIES.  What is held: the
        #  ladder's power, exactly, and the factor's backward error |L L^T - (M + jitter I)| / |M| against the
        #  stand-in's own, within a factor of two -- or 1e-10 forward agreement when that holds anyway.)
        L_dev, pw_dev = out
        L_ref = arrays[want['t'][0]['a']]
        pw_ref = want['t'][1] if want['t'][1] is None else want['t'][1]['i']
        assert pw_dev == pw_ref, (where, 'jitter power', pw_dev, pw_ref)
        fwd = _rel(L_dev, L_ref)
        if fwd > tol:
          M = np.asarray(args[0], dtype=float)
          jit = 0.0 if pw_ref is None else (10.0 ** pw_ref) * float(np.diag(M).max())
          Mj = M + jit * np.eye(len(M))
          res = lambda L: float(np.max(np.abs(np.tril(L).dot(np.tril(L).T) - Mj)) / np.max(np.abs(M)))
          assert res(L_dev) <= max(1e-13, 2.0 * res(L_ref)), (where, 'backward error', res(L_dev), res(L_ref), fwd)
        else:
          worst[0] = max(worst[0], fwd)
      elif ev['m'] == 'thompson':
        # (value, index[, samples, jitter powers per block]): the index is the decision and must be the reference's;
        # the stand-in does not report the powers (None).  A joint draw through a numerically singular covariance
        # (more candidates than training points: the ladder's jitter decides) cannot agree to 1e-10 -- the value is
        # then held to the bound of tests/truth_bounds.py: twice the stand-in's own distance from the same draw in
        # extended precision (single SE / Matern kernels, which have a truth).
        items = want['t'][:3]
        assert int(out[1]) == items[1]['i'], (where, out[1], items[1]['i'])
        try:
          _compare(tuple(out[:len(items)]), {'t': items}, arrays, tol, where, worst)
        except AssertionError:
          if ev['h'] not in fits:
            raise
          worst[0] = max(worst[0], _thompson_by_truth(target, fits[ev['h']], args, kwargs, where))
      elif ev['m'] in ('acq_argmax', 'add_ucb_group') and isinstance(want, dict) and 't' in want and \
          isinstance(want['t'][1], dict) and 'i' in want['t'][1] and int(out[1]) != want['t'][1]['i']:
        # The index is the decision and must be the reference's -- unless the two candidates TIE: tree-search frontiers
        # hold symmetric cells whose acquisition values agree to the last bits (profiles/r06_argmax_gaps.json: 1 % of the
        # expansions of a run lie below 1e-12), and which of two equal values np.argmax meets first is then decided by
        # rounding.  Accepted only when the live values at the two indices agree to 1e-12 of the largest value; counted.
        live = out if kwargs.get('return_vals') else getattr(target, ev['m'])(*args, **dict(kwargs, return_vals=True))
        vals = np.asarray(live[2], dtype=float)
        i_dev, i_ref = int(live[1]), want['t'][1]['i']
        scale = float(np.max(np.abs(vals[np.isfinite(vals)])))
        gap = abs(vals[i_dev] - vals[i_ref]) / (scale if scale > 0 else 1.0)
        assert gap <= 1e-12, (where, 'index differs and it is not a tie', i_dev, i_ref, vals[i_dev], vals[i_ref], gap)
        rec['meta'].setdefault('ties', []).append((where, gap))
        _compare(out[0], want['t'][0], arrays, tol, where, worst)
        if len(want['t']) > 2 and len(out) > 2:          # (the values themselves, when the call asked for them)
          _compare(out[2], want['t'][2], arrays, tol, where, worst)
      else:
        _compare(out, want, arrays, tol, where, worst)
  finally:
    for gp in made:
      try:
        gp.free()
      except Exception:      # pylint: disable=broad-except
        pass
  return len(rec['events']), worst[0], rec['meta']
