"""Untimed extras of bench.py (`configs` on the JSON line): the workloads Dragonfly spends its time in
besides the headline step, each with the CPU oracle timed beside it and an equality / parity flag.

  C1            BASELINE config 1 through the reference-interface mirrors: Branin, n = 200, SE kernel,
                fit + UCB over 1000 random candidates (latency; the recommended point equals the oracle's)
  hp_tuning     the tuning objective (log marginal likelihood) for 500 and 10 000 hyper-parameter
                candidates -- the reference's budget min(1e4, max(500, 50 #hps)), gp/gp_core.py:456-457
                -- at n in {50, 200, 1000, 4096} through dfh_gp_lml_batch
  append        posterior update with one new observation (q = 1) at n in {1000, 4096, 16384}
                against a full refit (the reference refits: gp_core.py:129-131)
  pdoo          acquisition maximisation by the reference's PDOO tree search, budget 2000, on config 2's GP
  C4_full_1gpu  BASELINE config 4 whole on ONE GPU: the C3 fit + blocked-joint Thompson sampling over all
                2 097 152 candidates (the strong-scaling reference point of `--scaling strong`)

Nothing here is inside bench.py's timed region; the oracle (oracle/ref_numpy.py) is the checker and
the CPU leg only.
"""
import os
import time
from argparse import Namespace

import numpy as np

import bench_configs as BC


def _median_ms(fn, sync, reps=5, warm=1):
  for _ in range(warm):
    fn()
  sync()
  ts = []
  for _ in range(reps):
    t0 = time.perf_counter()
    fn()
    sync()
    ts.append((time.perf_counter() - t0) * 1e3)
  return sorted(ts)[len(ts) // 2]


def _rel(a, b):
  a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
  den = float(np.max(np.abs(b))) if b.size else 1.0
  return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def branin(X):
  """ Branin on [0,1]^2 (the reference scales [-5,10] x [0,15]; euclidean_synthetic_functions) -- maximised """
  x1 = 15.0 * X[:, 0] - 5.0
  x2 = 15.0 * X[:, 1]
  a, b, c, r, s, t = 1.0, 5.1 / (4 * np.pi ** 2), 5.0 / np.pi, 6.0, 10.0, 1.0 / (8 * np.pi)
  return -(a * (x2 - b * x1 ** 2 + c * x1 - r) ** 2 + s * (1 - t) * np.cos(x1) + s)


def config1(eng):
  """ Through the mirrors (dragonfly_amd.gp_core.GP + gpb_acquisitions.asy.ucb): what
      gp_bandit.py:522-560 does per evaluation once the hyper-parameters are chosen. """
  from dragonfly_amd import gpb_acquisitions as A
  from dragonfly_amd import kernel as K
  from dragonfly_amd.gp_core import GP
  from dragonfly_amd.oper_utils import EuclideanDomain
  from oracle import ref_numpy as O
  n, d, m = 200, 2, 1000
  rs = np.random.RandomState(101)
  X = rs.random_sample((n, d))
  Y = branin(X) + 0.01 * rs.randn(n)
  scale, bw, noise, mean_c = float(Y.var()), np.array([0.25, 0.35]), float(Y.var() / 100), float(np.median(Y))
  bounds = np.array([[0.0, 1.0]] * d)
  anc = Namespace(max_evals=m, t=n, domain=EuclideanDomain(bounds), curr_max_val=float(Y.max()),
                  eval_points_in_progress=[], acq_opt_method='rand', handle_parallel='halluc', is_mf=False,
                  domain_bounds=bounds)
  mean_func = lambda x, _c=mean_c: np.array([_c] * len(x))
  Xl, Yl = list(X), list(Y)
  box = {}

  def device():
    np.random.seed(1101)
    gp = GP(Xl, Yl, K.SEKernel(d, scale, bw), mean_func, noise)
    box['pt'] = A.asy.ucb(gp, anc)
    box['lml'] = gp.compute_log_marginal_likelihood()

  def oracle():
    np.random.seed(1101)
    og = O.GPOracle(X, Y, O.KernelSpec('se', d, scale, bw), mean_c, noise)
    cands = np.random.random((m, d))                        # oper_utils.py:61-62 on the unit cube
    mu, sd = og.eval(cands, 'std')
    vals = O.acq_values('ucb', mu, sd, O.ucb_beta_th(d, n))
    box['opt'] = cands[O.argmax_first(vals)[1]]
    box['olml'] = og.lml()

  ms_dev = _median_ms(device, eng.sync, reps=9, warm=2)
  ms_cpu = _median_ms(oracle, lambda: None, reps=5, warm=1)
  return {'workload': 'Branin d=2, n=200, SE: GP fit + UCB arg-max over 1000 random candidates, through the mirrors '
                      '(gp_core.GP, gpb_acquisitions.asy.ucb)',
          'ms': round(ms_dev, 3), 'oracle_ms': round(ms_cpu, 3),
          'point_equal': bool(np.array_equal(box['pt'], box['opt'])),
          'lml_rel': abs(box['lml'] - box['olml']) / abs(box['olml'])}


def hp_tuning(eng):
  from dragonfly_amd.engine import KernelSpec
  from oracle import ref_numpy as O
  out = {}
  for (n, d) in ((50, 3), (200, 6), (1000, 6), (4096, 6)):
    rs = np.random.RandomState(n)
    X = rs.rand(n, d)
    Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
    yv = float(Y.var())
    nb = 10000
    scales = yv * np.exp(rs.randn(nb))
    bws = np.exp(rs.uniform(np.log(0.3), np.log(3.0), size=(nb, d)))
    means = list(rs.randn(nb) * 0.1)
    noises = list(yv * np.exp(rs.uniform(np.log(0.005), np.log(0.2), size=nb)))
    specs = [KernelSpec('se', d, scales[c], bws[c]) for c in range(nb)]
    Xd = eng.to_device(X)
    eng.gp_lml_batch(specs[:64], Xd, Y, means[:64], noises[:64])
    eng.sync()
    row = {}
    lml = None
    for budget in (500, 10000):
      if n >= 4096 and budget > 500:
        # 10 000 fits at n = 4096 are ~10 s of device time: measured on 2000 and scaled (equal work per fit)
        t0 = time.perf_counter()
        part = eng.gp_lml_batch(specs[:2000], Xd, Y, means[:2000], noises[:2000])
        eng.sync()
        row['ms_%d' % budget] = round((time.perf_counter() - t0) * 1e3 * budget / 2000.0, 2)
        row['ms_%d_note' % budget] = 'measured on 2000 candidates, scaled x5'
        lml = part
        continue
      t0 = time.perf_counter()
      lml = eng.gp_lml_batch(specs[:budget], Xd, Y, means[:budget], noises[:budget])
      eng.sync()
      row['ms_%d' % budget] = round((time.perf_counter() - t0) * 1e3, 2)
    # a small group on its own (a tree-search frontier, a slice sampler's loop): 64 and 8 candidates per call
    for small in (64, 8):
      _median_ms(lambda: eng.gp_lml_batch(specs[:small], Xd, Y, means[:small], noises[:small]), eng.sync, reps=2, warm=1)
      row['ms_batch_of_%d' % small] = round(_median_ms(lambda: eng.gp_lml_batch(specs[:small], Xd, Y, means[:small], noises[:small]),
                                                       eng.sync, reps=5, warm=0), 4)
    if n == 50:
      # the slice sampler's call (dragonfly/gp/gp_core.py:551-574 under sampling/slice.py): ONE candidate per call
      _median_ms(lambda: eng.gp_lml_batch(specs[:1], Xd, Y, means[:1], noises[:1]), eng.sync, reps=20, warm=5)
      t0 = time.perf_counter()
      for c in range(1000):
        eng.gp_lml_batch(specs[c:c + 1], Xd, Y, means[c:c + 1], noises[c:c + 1])
      row['ms_one_candidate'] = round((time.perf_counter() - t0), 5)      # s per 1000 calls == ms per call
    kc = 24 if n <= 1000 else 6
    t0 = time.perf_counter()
    ref = [O.GPOracle(X, Y, O.KernelSpec('se', d, scales[c], bws[c]), means[c], noises[c]).lml() for c in range(kc)]
    row['oracle_ms_per_eval'] = round((time.perf_counter() - t0) * 1e3 / kc, 3)
    row['ms_per_eval'] = round(row['ms_10000'] / 10000.0, 4)
    row['lml_rel_max'] = max(abs(lml[c] - ref[c]) / abs(ref[c]) for c in range(kc))
    row['lml_parity'] = bool(row['lml_rel_max'] <= 1e-10)
    out['n%d' % n] = row
    Xd.free()
  # n = 2000: 512 candidates in one call (the one-workgroup-per-candidate objective at its largest size class)
  try:
    n, d, nb = 2000, 6, 512
    rs = np.random.RandomState(n)
    X = rs.rand(n, d)
    Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
    yv = float(Y.var())
    specs = [KernelSpec('se', d, yv * np.exp(rs.randn()), np.exp(rs.uniform(np.log(0.3), np.log(3.0), size=d))) for _ in range(nb)]
    means = list(rs.randn(nb) * 0.1)
    noises = list(yv * np.exp(rs.uniform(np.log(0.005), np.log(0.2), size=nb)))
    Xd = eng.to_device(X)
    eng.gp_lml_batch(specs[:8], Xd, Y, means[:8], noises[:8])
    eng.sync()
    t0 = time.perf_counter()
    lml = eng.gp_lml_batch(specs, Xd, Y, means, noises)
    eng.sync()
    us = (time.perf_counter() - t0) * 1e6 / nb
    ref = [O.GPOracle(X, Y, O.KernelSpec('se', d, specs[c].scale, specs[c].bandwidths), means[c], noises[c]).lml() for c in range(3)]
    out['n2000'] = {'us_each_of_512': round(us, 2), 'lml_rel_max': max(abs(lml[c] - ref[c]) / abs(ref[c]) for c in range(3))}
    Xd.free()
  except Exception as e:      # pylint: disable=broad-except
    out['n2000'] = {'error': repr(e)}
  out['what'] = ('log marginal likelihood of 500 / 10000 hyper-parameter candidates per call sequence (SE-ARD d = 3 / 6), '
                 'dfh_gp_lml_batch; ms_batch_of_64 / _8: one call with that many candidates (wall, host side included); '
                 'oracle = one NumPy fit per candidate (sample of 24 / 6)')
  return out


def chol_sizes(eng):
  """ The fit's sections at the sizes between the configurations (SE-ARD, d = 8): kernel matrix, factorisation, the two
      solves + lml -- general_utils.py:178, gp_core.py:161-163, 222-227. """
  from dragonfly_amd.engine import KernelSpec
  out = {}
  for n in (4096, 8192):
    rs = np.random.RandomState(n)
    d = 8
    X = rs.rand(n, d)
    Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
    spec = KernelSpec('se', d, float(Y.var()), 0.2 * np.sqrt(d) * np.ones(d))
    noise = float(Y.var() / 20)
    Xd, yd = eng.to_device(X), eng.to_device(Y - float(np.median(Y)))

    def fit():
      eng.gp_fit(spec, Xd, yd, noise).free()
    for _ in range(2):
      fit()
    eng.sync()
    best = None
    for _ in range(5):
      eng.timings(True)
      fit()
      eng.sync()
      s = eng.timings(False)
      if best is None or s['chol'] < best['chol']:
        best = s
    out['n%d' % n] = {'ms': round(best['chol'], 4), 'solve_ms': round(best['solve'], 4), 'kernmat_ms': round(best['kernmat'], 4),
                      'frac_of_fp64_mfma_peak': round(float(n) ** 3 / 3 / (best['chol'] * 1e-3) / 1e12 / 78.6, 4)}
    Xd.free()
    yd.free()
  out['what'] = 'fit sections (section timers on: host-synchronised), best of 5, SE-ARD d = 8'
  return out


def append(eng):
  from dragonfly_amd.engine import KernelSpec
  from oracle import ref_numpy as O
  out = {}
  for (n, d) in ((1000, 6), (4096, 6), (16384, 32)):
    rs = np.random.RandomState(n)
    X = rs.rand(n + 1, d)
    Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n + 1)
    bw = 0.2 * np.sqrt(d) * np.ones(d)
    spec = KernelSpec('se', d, float(Y.var()), bw)
    noise = float(Y.var() / 20)
    Xd = eng.to_device(X[:n])
    base = eng.gp_fit(spec, Xd, Y[:n], noise)

    def one():
      ext = base.append(X[n:n + 1], Y[:n + 1])
      ext.free()
    ms = _median_ms(one, eng.sync, reps=5, warm=1)
    Xd1 = eng.to_device(X)

    def refit():
      g = eng.gp_fit(spec, Xd1, Y, noise)
      g.free()
    ms_full = _median_ms(refit, eng.sync, reps=3, warm=1)
    ext = base.append(X[n:n + 1], Y[:n + 1])
    full = eng.gp_fit(spec, Xd1, Y, noise)
    row = {'append_ms': round(ms, 3), 'device_refit_ms': round(ms_full, 3),
           'alpha_rel_vs_refit': _rel(ext.get_alpha(), full.get_alpha()),
           'lml_rel_vs_refit': abs(ext.lml - full.lml) / abs(full.lml)}
    if n <= 4096:
      t0 = time.perf_counter()
      og = O.GPOracle(X, Y, O.KernelSpec('se', d, float(Y.var()), bw), 0.0, noise)
      row['oracle_refit_ms'] = round((time.perf_counter() - t0) * 1e3, 2)
      row['alpha_rel_vs_oracle'] = _rel(ext.get_alpha(), og.alpha)
      row['parity'] = bool(row['alpha_rel_vs_oracle'] <= 1e-10 and abs(ext.lml - og.lml()) <= 1e-10 * abs(og.lml()))
    else:
      row['parity'] = bool(row['alpha_rel_vs_refit'] <= 1e-10 and row['lml_rel_vs_refit'] <= 1e-10)
    ext.free(); full.free(); base.free(); Xd.free(); Xd1.free()
    out['n%d' % n] = row
  out['what'] = 'one new observation (q = 1): dfh_gp_append against a full refit (device, and the NumPy oracle up to n = 4096)'
  return out


def pdoo(eng):
  """ The reference maximises acquisitions with PDOO by default for cts domains when the budget is
      small (gp_bandit.py:71-76); budget 2000 on config 2's GP, UCB.  One point per device call is the
      reference's access pattern; the batched frontier visits the same nodes (identical result). """
  from dragonfly_amd.doo import pdoo_maximise_batched
  from dragonfly_amd.engine import KernelSpec
  from oracle import ref_numpy as O
  c = BC.config2()
  spec = KernelSpec('matern', c['d'], c['scale'], c['bw'], nu=c['nu'])
  gp = eng.gp_fit(spec, c['X'], c['Y'] - c['mean_c'], c['noise'])
  beta = O.ucb_beta_th(c['d'], c['n'])

  def ucb(pts):
    mu, sd = gp.predict(pts)
    return mu + beta * sd
  bounds = [[0.0, 1.0]] * c['d']
  res = {}
  for frontier, depth in ((0, 0), (32, 2)):
    pdoo_maximise_batched(ucb, bounds, 200, frontier=frontier, depth=depth)      # warm-up
    t0 = time.perf_counter()
    v, p, h = pdoo_maximise_batched(ucb, bounds, 2000, frontier=frontier, depth=depth, return_history=True)
    res[frontier] = ((time.perf_counter() - t0) * 1e3, v, p, h)
  # the oracle: the same search with the NumPy posterior as its objective (first 300 evaluations timed, scaled)
  og = O.GPOracle(c['X'], c['Y'], O.KernelSpec('matern', c['d'], c['scale'], c['bw'], nu=c['nu']), c['mean_c'], c['noise'])

  def oucb(pts):
    mu, sd = og.eval(np.atleast_2d(pts), 'std')
    return mu - c['mean_c'] + beta * sd
  t0 = time.perf_counter()
  vo, po, ho = pdoo_maximise_batched(oucb, bounds, 300, frontier=0, depth=0, return_history=True)
  t_or = (time.perf_counter() - t0) * 1e3
  v300, p300, _ = pdoo_maximise_batched(ucb, bounds, 300, frontier=0, depth=0)
  gp.free()
  return {'workload': 'PDOO, budget 2000, UCB on config 2 (n=4096, d=6, Matern-2.5)',
          'ms_one_point_per_call': round(res[0][0], 2), 'device_calls_one_point': int(res[0][3].device_calls),
          'ms_frontier32': round(res[32][0], 2), 'device_calls_frontier32': int(res[32][3].device_calls),
          'choice_equal_between_schedules': bool(res[0][1] == res[32][1] and np.array_equal(res[0][2], res[32][2])),
          'oracle_ms_budget300': round(t_or, 2),
          'oracle_ms_budget2000_scaled': round(t_or * 2000.0 / 300.0, 2),
          'choice_equal_vs_oracle_budget300': bool(np.array_equal(p300, po)),
          'value_rel_vs_oracle_budget300': abs(v300 - vo) / abs(vo)}


def config4_full_one_gpu(eng, prob, spec, steps=2):
  """ The whole of BASELINE config 4 on one device: fit + TS over 2 097 152 candidates, generated
      shard by shard into one resident buffer.  Its arg-max equals the reduce over the eight
      weak-scaling shards evaluated one after the other on the same device. """
  from dragonfly_amd import parallel
  m, d = BC.CANDS_TOTAL_8, BC.DIM
  cd, ud = eng.empty((m, d)), eng.empty((m,))
  for r in range(8):
    cands, U = BC.config4_shard(r)
    cd.view(r * BC.CANDS_PER_GPU * d, (BC.CANDS_PER_GPU, d)).upload(cands)
    ud.view(r * BC.CANDS_PER_GPU, (BC.CANDS_PER_GPU,)).upload(U)
  Xd, yd = eng.to_device(prob['X']), eng.to_device(prob['Y'] - prob['mean_c'])
  box = {}

  def step():
    gp = eng.gp_fit(spec, Xd, yd, prob['noise'])
    box['r'] = gp.thompson(cd, ud, block=BC.TS_BLOCK, mean_const=prob['mean_c'])
    gp.free()
  step()
  eng.sync()
  t0 = time.perf_counter()
  for _ in range(steps):
    step()
  eng.sync()
  ms = (time.perf_counter() - t0) * 1e3 / steps
  v_full, i_full = box['r'][0], int(box['r'][1])
  # the eight shards one after the other, reduced like the ranks' all-gather (first maximum wins)
  gp = eng.gp_fit(spec, Xd, yd, prob['noise'])
  vals, idxs = [], []
  for r in range(8):
    v, i = gp.thompson(cd.view(r * BC.CANDS_PER_GPU * d, (BC.CANDS_PER_GPU, d)), ud.view(r * BC.CANDS_PER_GPU, (BC.CANDS_PER_GPU,)),
                       block=BC.TS_BLOCK, mean_const=prob['mean_c'])[:2]
    vals.append(v); idxs.append(int(i) + r * BC.CANDS_PER_GPU)
  gp.free()
  v_red, i_red = parallel.reduce_argmax(vals, idxs)
  for a in (cd, ud, Xd, yd):
    a.free()
  return {'workload': 'C3 fit + C4 whole: blocked-joint Thompson sampling, block 4096, over all 2097152 candidates on ONE GPU',
          'n_gpus': 1, 'candidates_total': m, 'steps': steps, 'ms_per_step': round(ms, 2),
          'candidates_per_s': round(m / (ms * 1e-3), 1),
          'ts_best': v_full, 'ts_argmax': i_full,
          'argmax_equals_reduce_over_8_shards': bool(i_full == int(i_red) and v_full == v_red)}


def hallucinated_batch(eng, workers=8, m_parity=4096):
  """ A synchronous batch at size (opt/gpb_acquisitions.py:90-115: every earlier recommendation is a
      hallucinated in-progress point of the next, gp/gp_core.py:192-220): config 2's GP (n = 4096, Matern-2.5),
      EI with the hallucinated std, q = 0 .. workers-1 extra rows.  The reference re-factors the (n+q) x (n+q)
      matrix per call; the device appends q rows to the factor.  Device over all 65536 candidates (timed per q);
      the same batch over the first m_parity candidates against the oracle (choice and EI values). """
  from dragonfly_amd.engine import KernelSpec
  from oracle import ref_numpy as O
  c = BC.config2()
  spec = KernelSpec('matern', c['d'], c['scale'], c['bw'], nu=c['nu'])
  Xd, yd = eng.to_device(c['X']), eng.to_device(c['Y'] - c['mean_c'])
  gp = eng.gp_fit(spec, Xd, yd, c['noise'])

  def device_batch(cands_host, cands_dev):
    picks, vals, ms = [], [], []
    for _ in range(workers):
      eng.sync()
      t0 = time.perf_counter()
      bv, bi, ev = gp.acq_argmax('ei', cands_dev, params=(c['best'], 0.0), mean_const=c['mean_c'],
                                 X_halluc=np.array([cands_host[i] for i in picks]) if picks else None, return_vals=True)
      eng.sync()
      ms.append((time.perf_counter() - t0) * 1e3)
      picks.append(int(bi)); vals.append(ev)
    return picks, vals, ms
  cd = eng.to_device(c['cands'])
  device_batch(c['cands'], cd)                              # warm-up
  picks_all, _, ms_all = device_batch(c['cands'], cd)
  sub = np.ascontiguousarray(c['cands'][:m_parity])
  sd = eng.to_device(sub)
  picks_d, vals_d, _ = device_batch(sub, sd)
  og = O.GPOracle(c['X'], c['Y'], O.KernelSpec('matern', c['d'], c['scale'], c['bw'], nu=c['nu']), c['mean_c'], c['noise'])
  picks_o, rels, ms_o = [], [], []
  for w in range(workers):
    t0 = time.perf_counter()
    mus, sds = [], []
    for i0 in range(0, m_parity, 4096):
      chunk = sub[i0:i0 + 4096]
      if picks_o:
        mu, s_h = og.eval_with_hallucinated_observations(chunk, sub[picks_o], 'std')
      else:
        mu, s_h = og.eval(chunk, 'std')
      mus.append(mu); sds.append(s_h)
    ev = O.acq_values('ei', np.concatenate(mus), np.concatenate(sds), c['best'])
    ms_o.append((time.perf_counter() - t0) * 1e3)
    rels.append(_rel(vals_d[w], ev))
    picks_o.append(O.argmax_first(ev)[1])
  for a in (cd, sd, Xd, yd):
    a.free()
  gp.free()
  return {'workload': 'synchronous batch of %d by EI with hallucinated in-progress points on config 2\'s GP (n=4096, d=6, '
                      'Matern-2.5): q = 0..%d extra rows' % (workers, workers - 1),
          'device_ms_per_q_over_65536_candidates': [round(v, 3) for v in ms_all], 'device_batch_ms': round(sum(ms_all), 3),
          'device_picks_over_65536': picks_all,
          'oracle_ms_per_q_over_%d_candidates' % m_parity: [round(v, 1) for v in ms_o],
          'picks_equal_vs_oracle_over_%d' % m_parity: bool(picks_d == picks_o), 'ei_rel_per_q': rels}


def bo_wallclock():
  """ Only where a Dragonfly checkout is beside the GPU (DRAGONFLY_REFERENCE): dragonfly.maximise_function on Hartmann6
      with default options, 60 evaluations, the reference as it is and with dragonfly_amd.install(), each in a process of
      its own (tools/bo_wallclock.py; install() rebinds module globals).  profiles/r05_bo_wallclock.json holds the
      builder's runs at 60 / 200 / 1000 evaluations. """
  import json
  import subprocess
  import sys
  ref = os.environ.get('DRAGONFLY_REFERENCE', '')
  if not ref or not os.path.isdir(os.path.join(ref, 'dragonfly')):
    return {'skipped': 'no Dragonfly checkout on this box (DRAGONFLY_REFERENCE); see profiles/r05_bo_wallclock.json'}
  tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'bo_wallclock.py')
  out = {}
  for mode in ('install', 'ref'):
    res = subprocess.run([sys.executable, tool, '60', mode], capture_output=True, text=True, timeout=900)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    out[mode] = json.loads(lines[-1]) if lines else {'error': res.stderr[-400:]}
  if 'wall_s' in out['ref'] and 'wall_s' in out['install']:
    out['speedup_wall'] = round(out['ref']['wall_s'] / out['install']['wall_s'], 2)
  return out


def run_all(eng, prob, spec, include_c4_full=True):
  out = {}
  for name, fn in (('C1', lambda: config1(eng)), ('hp_tuning', lambda: hp_tuning(eng)), ('append', lambda: append(eng)),
                   ('chol_sizes', lambda: chol_sizes(eng)),
                   ('pdoo', lambda: pdoo(eng)), ('hallucinated_batch', lambda: hallucinated_batch(eng)),
                   ('bo_wallclock', bo_wallclock)):
    try:
      out[name] = fn()
    except Exception as e:      # pylint: disable=broad-except
      out[name] = {'error': repr(e)}
  if include_c4_full:
    try:
      out['C4_full_1gpu'] = config4_full_one_gpu(eng, prob, spec)
    except Exception as e:      # pylint: disable=broad-except
      out['C4_full_1gpu'] = {'error': repr(e)}
  return out
