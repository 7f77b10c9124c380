"""Bring-up checks on a real MI355X: each stage runs in its own process (a faulting kernel must
not hide the later stages).  Usage:  python tests/gpu_check.py [stage ...]   (no args = all)
Writes a log per stage under gpurun_out/check/.  Test infrastructure (it lives under tests/ because
some stages compare with oracle/, which only tests may import); not collected by pytest.
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def relerr(a, b):
  a = np.asarray(a, dtype=float)
  b = np.asarray(b, dtype=float)
  den = np.max(np.abs(b))
  return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def stage_gemm():
  from dragonfly_amd.engine import Engine
  eng = Engine()
  print(eng.name())
  rs = np.random.RandomState(0)
  ok = True
  for (M, N, K) in [(16, 16, 4), (128, 128, 16), (128, 128, 64), (256, 384, 48), (130, 70, 33),
                    (64, 64, 64), (1, 5, 3), (300, 129, 17), (512, 512, 512), (2048, 2048, 64),
                    (1900, 1700, 70), (2048, 1536, 48)]:
    A = rs.randn(M, K)
    B = rs.randn(N, K)
    C0 = rs.randn(M, N)
    ref = 0.5 * C0 + 1.5 * A.dot(B.T)
    got = eng.gemm(A, B, C0, alpha=1.5, beta=0.5)
    e = relerr(got, ref)
    Bn = rs.randn(K, N)
    ref2 = -A.dot(Bn)
    got2 = eng.gemm(A, Bn, None, alpha=-1.0, beta=0.0, transb=True)
    e2 = relerr(got2, ref2)
    print('gemm NT %s err %.2e | NN err %.2e' % ((M, N, K), e, e2))
    ok &= e < 1e-13 and e2 < 1e-13
  # lower-only: tiles above the diagonal must be untouched
  for M in (384, 2304):
   A = rs.randn(M, 40)
   C0 = rs.randn(M, M)
   got = eng.gemm(A, A, C0, alpha=-1.0, beta=1.0, lower_only=True)
   ref = C0 - A.dot(A.T)
   low = np.tril_indices(M)
   e = relerr(got[low], ref[low])
   print('gemm lower M=%d err %.2e' % (M, e))
   ok &= e < 1e-13
  M = 384
  A = rs.randn(M, 40)
  C0 = rs.randn(M, M)
  got = eng.gemm(A, A, C0, alpha=-1.0, beta=1.0, lower_only=True)
  ref = C0 - A.dot(A.T)
  low = np.tril_indices(M)
  e = relerr(got[low], ref[low])
  untouched = np.array_equal(got[:128, 128:], C0[:128, 128:]) and np.array_equal(got[128:256, 256:], C0[128:256, 256:])
  print('gemm lower err %.2e untouched-upper-tiles %s' % (e, untouched))
  ok &= e < 1e-13 and untouched
  print('STAGE gemm', 'PASS' if ok else 'FAIL')


def _specs():
  from dragonfly_amd.engine import KernelSpec
  from oracle import ref_numpy as O
  rs = np.random.RandomState(5)
  out = []
  for d in (1, 2, 6, 32, 45):
    bw = 0.3 + rs.rand(d)
    out.append(('se-d%d' % d, KernelSpec('se', d, 2.3, bw), O.KernelSpec('se', d, 2.3, bw)))
    for nu in (0.5, 1.5, 2.5):
      out.append(('matern%.1f-d%d' % (nu, d), KernelSpec('matern', d, 1.7, bw, nu=nu),
                  O.KernelSpec('matern', d, 1.7, bw, nu=nu)))
  d = 23
  perm = rs.permutation(d)
  groups = [list(perm[0:5]), list(perm[5:8]), list(perm[8:16]), list(perm[16:23])]
  bws = [0.4 + rs.rand(len(g)) for g in groups]
  kinds = ['se', 'matern', 'se', 'matern']
  nus = [0.0, 2.5, 0.0, 1.5]
  scales = [1.0, 1.0, 0.7, 1.3]
  spec = KernelSpec('additive', d, 3.1, groups=groups, sub_kinds=kinds, sub_scales=scales,
                    sub_nus=nus, sub_bandwidths=bws)
  subs = [O.KernelSpec(k, len(g), s, b, nu=(n if k == 'matern' else None))
          for k, g, s, b, n in zip(kinds, groups, scales, bws, nus)]
  out.append(('additive-d23', spec, O.KernelSpec('additive', d, 3.1, groups=groups, subs=subs)))
  return out


def stage_kernmat():
  from dragonfly_amd.engine import Engine
  from oracle import ref_numpy as O
  eng = Engine()
  rs = np.random.RandomState(1)
  ok = True
  for name, spec, ospec in _specs():
    d = spec.dim
    for (n1, n2) in [(7, 5), (128, 128), (200, 333)]:
      X1 = rs.rand(n1, d)
      X2 = rs.rand(n2, d)
      K = eng.kernel_matrix(spec, X1, X2)
      Kr = ospec(X1, X2)
      e = relerr(K, Kr)
      Ks = eng.kernel_matrix(spec, X1, None, diag_add=0.25)
      Ksr = ospec(X1, X1) + 0.25 * np.eye(n1)
      es = relerr(Ks, Ksr)
      sym = np.array_equal(Ks, Ks.T)
      tol = 1e-6 if 'matern0.5' in name else 1e-12     # sqrt near 0 amplifies rounding of dist_sq
      good = e < 1e-12 and es < tol and sym
      ok &= good
      if not good or (n1, n2) == (200, 333):
        print('%-16s (%d,%d) cross err %.2e sym err %.2e bitwise-symmetric %s' % (name, n1, n2, e, es, sym))
  X1 = rs.rand(50, 3)
  X2 = rs.rand(40, 3)
  e = relerr(eng.dist_squared(X1, X2), O.dist_squared(X1, X2))
  print('dist_squared err %.2e' % e)
  ok &= e < 1e-13
  print('STAGE kernmat', 'PASS' if ok else 'FAIL')


def _spd(n, rs, cond_noise=0.05):
  from oracle import ref_numpy as O
  d = 4
  X = rs.rand(n, d)
  K = O.se_kernel(X, X, 1.0, np.full(d, 0.4)) + cond_noise * np.eye(n)
  return K


def stage_chol():
  from dragonfly_amd.engine import Engine
  eng = Engine()
  rs = np.random.RandomState(2)
  ok = True
  for n in (1, 5, 64, 65, 100, 128, 200, 512, 513, 700, 1024, 1500, 2048):
    M = _spd(n, rs)
    L = eng.cholesky(M)
    Lr = np.linalg.cholesky(M)
    e = relerr(L, Lr)
    rec = relerr(L.dot(L.T), M)
    up = float(np.abs(np.triu(L, 1)).max()) if n > 1 else 0.0
    print('chol n=%d err-vs-lapack %.2e recon %.2e upper-max %.1e' % (n, e, rec, up))
    ok &= e < 1e-11 and rec < 1e-13 and up == 0.0
  # not PD
  M = _spd(300, rs)
  M[150, 150] = -1.0
  try:
    eng.cholesky(M)
    print('non-PD: no exception  FAIL')
    ok = False
  except np.linalg.LinAlgError as ex:
    print('non-PD raised LinAlgError:', ex)
  # stable cholesky ladder: rank-deficient PSD matrix
  A = rs.randn(200, 20)
  M = A.dot(A.T)
  from oracle import ref_numpy as O
  try:
    Lr, pr = O.stable_cholesky(M, return_power=True)
  except Exception as ex:   # pylint: disable=broad-except
    Lr, pr = None, repr(ex)
  L, p = eng.stable_cholesky(M, return_power=True)
  print('stable_cholesky power device %s oracle %s recon %.2e' % (p, pr, relerr(L.dot(L.T), M)))
  # triangular solves
  for n, nrhs in ((300, 1), (700, 1), (1100, 37), (600, 600)):
    M = _spd(n, rs)
    Lr = np.linalg.cholesky(M)
    b = rs.randn(n) if nrhs == 1 else rs.randn(n, nrhs)
    from scipy.linalg import solve_triangular
    for upper in (False, True):
      x = eng.solve_triangular(Lr, b, upper=upper)
      xr = solve_triangular(Lr.T if upper else Lr, b, lower=not upper)
      e = relerr(x, xr)
      print('solve_triangular n=%d nrhs=%d upper=%s err %.2e' % (n, nrhs, upper, e))
      ok &= e < 1e-10
  print('STAGE chol', 'PASS' if ok else 'FAIL')


def stage_gp():
  from dragonfly_amd.engine import Engine
  from oracle import ref_numpy as O
  eng = Engine()
  rs = np.random.RandomState(3)
  ok = True
  for name, spec, ospec in _specs():
    if name not in ('se-d6', 'matern2.5-d6', 'se-d32', 'additive-d23', 'matern1.5-d2'):
      continue
    d = spec.dim
    for n in (60, 700, 1500):
      X = rs.rand(n, d)
      Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
      mean_c = float(np.median(Y))
      noise = float(Y.var() / 20)
      og = O.GPOracle(X, Y, ospec, mean_c, noise)
      gp = eng.gp_fit(spec, X, Y - mean_c, noise)
      e_alpha = relerr(gp.get_alpha(), og.alpha)
      e_L = relerr(gp.get_L(), og.L)
      e_lml = abs(gp.lml - og.lml()) / abs(og.lml())
      Xs = rs.rand(777, d)
      mu, sd = gp.predict(Xs)
      mur, sdr = og.eval(Xs, 'std')
      e_mu = relerr(mu + mean_c, mur)
      e_sd = relerr(sd, sdr)
      best = float(Y.max())
      line = '%-14s n=%d alpha %.1e L %.1e lml %.1e mu %.1e sd %.1e' % (name, n, e_alpha, e_L, e_lml, e_mu, e_sd)
      good = max(e_alpha, e_L, e_lml, e_mu, e_sd) < 1e-10
      for acq, params in (('ucb', (2.0, 0.0)), ('ei', (best, 0.0)), ('pi', (best, 0.0)), ('ttei', (best, 0.3))):
        bv, bi, vals = gp.acq_argmax(acq, Xs, params=params, mean_const=mean_c, return_vals=True)
        vr = O.acq_values(acq, mur, sdr, *params)
        rv, ri = O.argmax_first(vr)
        ev = relerr(vals, vr)
        line += ' %s %.1e%s' % (acq, ev, '' if bi == ri else ' ARGMAX-MISMATCH(%d,%d)' % (bi, ri))
        good &= ev < 1e-9 and bi == ri
      # hallucinated
      Xh = rs.rand(3, d)
      mu2, sd2 = gp.predict(Xs, X_halluc=Xh)
      _, sd2r = og.eval_with_hallucinated_observations(Xs, Xh, 'std')
      e_h = relerr(sd2, sd2r)
      line += ' halluc-sd %.1e' % e_h
      good &= e_h < 1e-10 and np.array_equal(mu2, mu)
      # TS (one block and several blocks)
      U = rs.randn(len(Xs))
      for blk in (len(Xs), 256):
        bv, bi, samp, jps = gp.thompson(Xs, U, block=blk, mean_const=mean_c, return_samples=True)
        sr = og.draw_samples_blocked(Xs, U, blk)
        e_ts = relerr(samp, sr)
        line += ' ts[%d] %.1e%s' % (blk, e_ts, '' if bi == int(np.argmax(sr)) else ' TS-ARGMAX-MISMATCH')
        good &= e_ts < 1e-7
      print(line, '' if good else '  <-- FAIL')
      ok &= good
      gp.free()
  print('STAGE gp', 'PASS' if ok else 'FAIL')


def stage_perf():
  from dragonfly_amd.engine import Engine, KernelSpec
  eng = Engine()
  rs = np.random.RandomState(4)
  # GEMM throughput
  for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (16384, 512, 16384), (15872, 15872, 512)]:
    A = eng.to_device(rs.rand(M, K) - 0.5)
    B = eng.to_device(rs.rand(N, K) - 0.5)
    Cd = eng.empty((M, N))
    eng.gemm(A, B, shape=(M, N, K), out=Cd)
    eng.sync()
    reps = 3
    eng.timer_begin()
    for _ in range(reps):
      eng.gemm(A, B, shape=(M, N, K), out=Cd)
    ms = eng.timer_end() / reps
    print('gemm NT %dx%dx%d: %.3f ms  %.1f TF/s' % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
    if M == N:
      eng.timer_begin()
      for _ in range(reps):
        eng.gemm(A, A, alpha=-1.0, beta=1.0, shape=(M, M, K), out=Cd, lower_only=True)
      ms = eng.timer_end() / reps
      print('  syrk-lower %dx%dx%d: %.3f ms  %.1f TF/s (useful flops M*M*K)' % (M, M, K, ms, 1.0 * M * M * K / ms / 1e9))
    A.free(); B.free(); Cd.free()
  # kernel matrix
  n, d = 16384, 32
  X = eng.to_device(rs.rand(n, d))
  Kd = eng.empty((n, n))
  for kind, nu in (('se', 0.0), ('matern', 2.5)):
    spec = KernelSpec(kind, d, 1.3, 0.2 * np.sqrt(d) * (0.5 + np.arange(d) / 32.0), nu=nu)
    eng.kernel_matrix(spec, X, None, out=Kd)
    eng.timings(True)
    for _ in range(3):
      eng.kernel_matrix(spec, X, None, out=Kd)
    t = eng.timings(True)
    ms = t['kernmat'] / 3
    gb = 8.0 * (n * n + 2 * n * d) / 1e9
    print('kernmat %s n=%d d=%d: %.3f ms  %.2f TB/s (algorithmic %.3f GB)' % (kind, n, d, ms, gb / ms, gb))
  # cholesky
  for n in (4096, 8192, 16384):
    spec = KernelSpec('se', d, 1.0, 0.2 * np.sqrt(d) * (0.5 + np.arange(d) / 32.0))
    Xn = eng.to_device(rs.rand(n, d))
    Kn = eng.empty((n, n))
    for rep in range(2):
      eng.kernel_matrix(spec, Xn, None, diag_add=0.05, out=Kn)
      eng.sync()
      t0 = time.time()
      eng.timings(True)
      eng.cholesky(Kn)
      eng.sync()
      wall = (time.time() - t0) * 1e3
      t = eng.timings(True)
    print('cholesky n=%d: %.2f ms (wall %.2f)  %.1f TF/s' % (n, t['chol'], wall, n ** 3 / 3.0 / t['chol'] / 1e9))
    Xn.free(); Kn.free()
  X.free(); Kd.free()
  print('STAGE perf DONE')


def stage_fit16k():
  """ C3-sized end-to-end: fit + posterior, with section timings. """
  from dragonfly_amd.engine import Engine, KernelSpec
  eng = Engine()
  n, d, m = 16384, 32, 65536
  rs = np.random.RandomState(103)
  X = rs.random_sample((n, d))
  w = (np.arange(d) + 1.0) / d
  Y = (X ** 2).dot(w) + 0.01 * rs.randn(n)
  bw = 0.2 * np.sqrt(d) * (0.5 + np.arange(d) / 32.0)
  spec = KernelSpec('se', d, float(Y.var()), bw)
  mean_c = float(np.median(Y))
  noise = float(Y.var() / 20)
  Xc = np.random.RandomState(203).random_sample((m, d))
  Xd = eng.to_device(X)
  yd = eng.to_device(Y - mean_c)
  Xcd = eng.to_device(Xc)
  for rep in range(2):
    eng.timings(True)
    t0 = time.time()
    gp = eng.gp_fit(spec, Xd, yd, noise)
    t1 = time.time()
    bv, bi = gp.acq_argmax('ucb', Xcd, params=(3.0, 0.0), mean_const=mean_c)
    t2 = time.time()
    t = eng.timings(True)
    print('rep %d fit %.1f ms acq(m=%d) %.1f ms lml %.6f best %.6f idx %d  sections %s' % (
        rep, (t1 - t0) * 1e3, m, (t2 - t1) * 1e3, gp.lml, bv, bi,
        {k: round(v, 2) for k, v in t.items()}))
    if rep == 0:
      gp.free()
  eng.timings(False)
  t0 = time.time()
  gp2 = eng.gp_fit(spec, Xd, yd, noise)
  t1 = time.time()
  bv, bi = gp2.acq_argmax('ucb', Xcd, params=(3.0, 0.0), mean_const=mean_c)
  t2 = time.time()
  print('untimed-sections: fit %.1f ms acq %.1f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
  print('STAGE fit16k DONE')


def stage_hptune():
  """ hyper-parameter tuning inner loop: batched lml vs one fit per candidate vs the CPU oracle """
  from dragonfly_amd.engine import Engine, KernelSpec
  from oracle import ref_numpy as O
  eng = Engine()
  for (n, d, nb) in [(50, 3, 512), (200, 6, 512), (1000, 6, 256), (4096, 6, 64), (16384, 32, 12)]:
    rs = np.random.RandomState(n)
    X = rs.rand(n, d)
    Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
    yv = float(Y.var())
    specs = [KernelSpec('se', d, yv * np.exp(rs.randn()), np.exp(rs.uniform(np.log(0.3), np.log(3.0), size=d)))
             for _ in range(nb)]
    means = list(rs.randn(nb) * 0.1)
    noises = list(yv * np.exp(rs.uniform(np.log(0.005), np.log(0.2), size=nb)))
    Xd = eng.to_device(X)
    eng.gp_lml_batch(specs[:8], Xd, Y, means[:8], noises[:8])
    eng.sync()
    t0 = time.time()
    lml = eng.gp_lml_batch(specs, Xd, Y, means, noises)
    tb = time.time() - t0
    k = min(nb, 64)
    t0 = time.time()
    one = []
    for c in range(k):
      g = eng.gp_fit(specs[c], Xd, Y - means[c], noises[c])
      one.append(g.lml)
      g.free()
    ts = (time.time() - t0) / k
    kc = min(nb, 4 if n > 4000 else 16)
    if n > 8192:
      kc = 0
    t0 = time.time()
    ref = [O.GPOracle(X, Y, O.KernelSpec('se', d, specs[c].scale, specs[c].bandwidths), means[c], noises[c]).lml()
           for c in range(kc)]
    tc = (time.time() - t0) / max(kc, 1)
    err1 = max(abs(lml[c] - one[c]) / abs(one[c]) for c in range(k))
    err2 = max([abs(lml[c] - ref[c]) / abs(ref[c]) for c in range(kc)] or [0.0])
    print('hp-tune n=%5d d=%2d nb=%3d: batched %.3f ms/cand | single fits %.3f ms/cand | oracle %.2f ms/cand '
          '| err vs single %.1e vs oracle %.1e' % (n, d, nb, tb * 1e3 / nb, ts * 1e3, tc * 1e3, err1, err2))
  print('STAGE hptune PASS')


def stage_append():
  """ incremental posterior update vs full refit """
  from dragonfly_amd.engine import Engine, KernelSpec
  eng = Engine()
  for (n, d) in [(1000, 6), (4096, 6), (16384, 32)]:
    rs = np.random.RandomState(n)
    X = rs.rand(n + 64, d)
    Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n + 64)
    spec = KernelSpec('se', d, float(Y.var()), 0.2 * np.sqrt(d) * np.ones(d))
    noise = float(Y.var() / 20)
    base = eng.gp_fit(spec, X[:n], Y[:n], noise)
    for q in (1, 8, 64):
      ext = base.append(X[n:n + q], Y[:n + q]); ext.free()
      t0 = time.time()
      for _ in range(3):
        ext = base.append(X[n:n + q], Y[:n + q]); ext.free()
      ta = (time.time() - t0) / 3
      full = eng.gp_fit(spec, X[:n + q], Y[:n + q], noise); full.free()
      t0 = time.time()
      full = eng.gp_fit(spec, X[:n + q], Y[:n + q], noise)
      tf = time.time() - t0
      ext = base.append(X[n:n + q], Y[:n + q])
      err = np.abs(ext.get_alpha() - full.get_alpha()).max() / np.abs(full.get_alpha()).max()
      print('append n=%5d q=%2d: append %.3f ms | full refit %.3f ms | alpha diff %.1e lml diff %.1e'
            % (n, q, ta * 1e3, tf * 1e3, err, abs(ext.lml - full.lml) / abs(full.lml)))
      ext.free(); full.free()
    base.free()
  print('STAGE append PASS')


def stage_configs():
  """ wall-clock of BASELINE configs 2, 3 and 5 (fit + candidate stage), inputs resident in HBM """
  from dragonfly_amd.engine import Engine, KernelSpec
  from oracle import ref_numpy as O
  eng = Engine()
  def timed(fn, reps=3):
    fn(); eng.sync()
    t0 = time.time()
    for _ in range(reps):
      fn()
    eng.sync()
    return (time.time() - t0) / reps * 1e3
  # C2: Hartmann6-like, n=4096, d=6, Matern-2.5, EI over 65536 candidates
  rs = np.random.RandomState(102)
  n, d, m = 4096, 6, 65536
  X = rs.random_sample((n, d)); Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
  spec = KernelSpec('matern', d, float(Y.var()), 0.5 * np.ones(d), nu=2.5)
  Xd, yd, cd = eng.to_device(X), eng.to_device(Y - np.median(Y)), eng.to_device(np.random.RandomState(202).random_sample((m, d)))
  noise = float(Y.var() / 20)
  box = {}
  def c2():
    gp = eng.gp_fit(spec, Xd, yd, noise); box['r'] = gp.acq_argmax('ei', cd, params=(float(Y.max()), 0.0)); gp.free()
  t_fit = timed(lambda: eng.gp_fit(spec, Xd, yd, noise).free())
  t_all = timed(c2)
  print('C2 n=4096 d=6 matern2.5 EI m=65536: fit %.2f ms, fit+acq %.2f ms (TRSM 1.1e12 flop -> %.1f TF/s over the acq stage)'
        % (t_fit, t_all, 4096.0 ** 2 * 65536 / ((t_all - t_fit) * 1e-3) / 1e12))
  # C3: n=16384, d=32, SE-ARD, posterior (mu, sigma) at 65536 candidates
  rs = np.random.RandomState(103)
  n, d, m = 16384, 32, 65536
  X = rs.random_sample((n, d)); Y = (X ** 2).dot((np.arange(d) + 1.0) / d) + 0.01 * rs.randn(n)
  spec = KernelSpec('se', d, float(Y.var()), 0.2 * np.sqrt(d) * (0.5 + np.arange(d) / 32.0))
  Xd, yd, cd = eng.to_device(X), eng.to_device(Y - np.median(Y)), eng.to_device(np.random.RandomState(203).random_sample((m, d)))
  noise = float(Y.var() / 20)
  def c3():
    gp = eng.gp_fit(spec, Xd, yd, noise); box['r'] = gp.acq_argmax('ucb', cd, params=(2.0, 0.0)); gp.free()
  t_fit = timed(lambda: eng.gp_fit(spec, Xd, yd, noise).free(), reps=2)
  t_all = timed(c3, reps=2)
  print('C3 n=16384 d=32 SE-ARD posterior m=65536: fit %.2f ms, fit+posterior+argmax %.2f ms (TRSM 1.76e13 flop -> %.1f TF/s)'
        % (t_fit, t_all, 16384.0 ** 2 * 65536 / ((t_all - t_fit) * 1e-3) / 1e12))
  # C5: additive d=100, 20 groups of 5, n=4096, add-UCB with 3276 candidates per group
  rs = np.random.RandomState(105)
  n, d, G = 4096, 100, 20
  X = rs.random_sample((n, d)); Y = (X ** 2).sum(axis=1) / 10 + 0.05 * rs.randn(n)
  perm = list(np.random.RandomState(405).permutation(d))
  groups = [perm[i:i + 5] for i in range(0, d, 5)]
  spec = KernelSpec('additive', d, float(Y.var()), groups=groups, sub_kinds=['se'] * G, sub_scales=[1.0] * G,
                    sub_nus=[0.0] * G, sub_bandwidths=[0.2 * np.sqrt(5) * np.ones(5)] * G)
  Xd, yd = eng.to_device(X), eng.to_device(Y - np.median(Y))
  noise = float(Y.var() / 20)
  ch = [np.random.RandomState(205 + g).random_sample((65536 // G, 5)) for g in range(G)]
  cg = [eng.to_device(c) for c in ch]
  def c5_groups():
    gp = eng.gp_fit(spec, Xd, yd, noise)
    for g in range(G):
      box['r'] = gp.add_ucb_group(g, 2.0, cg[g])
    gp.free()
  def c5_all():
    gp = eng.gp_fit(spec, Xd, yd, noise)
    box['r'] = gp.add_ucb_all([2.0] * G, ch)
    gp.free()
  t_fit = timed(lambda: eng.gp_fit(spec, Xd, yd, noise).free())
  t_grp = timed(c5_groups)
  t_all = timed(c5_all)
  print('C5 additive d=100 (20x5) n=4096 add-UCB 20 x 3276 candidates: fit %.2f ms, fit+acq %.2f ms in one call '
        '(%.2f ms with one call per group)' % (t_fit, t_all, t_grp))
  print('STAGE configs DONE')


def stage_rng():
  """ candidate generation: device MT19937 / Philox streams vs the host draw + upload """
  from dragonfly_amd.engine import Engine
  eng = Engine()
  for m, d in ((65536, 6), (262144, 32), (2097152, 32)):
    out = eng.empty((m, d))
    bounds = np.stack([np.zeros(d), np.ones(d) * 2.0], axis=1)
    rs = np.random.RandomState(1)
    t0 = time.time(); host = rs.random_sample((m, d)) * 2.0 + 0.0; t_gen = time.time() - t0
    t0 = time.time(); out.upload(host); eng.sync(); t_up = time.time() - t0
    rs = np.random.RandomState(1)
    eng.random_candidates(16, d, bounds=bounds, rng=np.random.RandomState(2), out=out)   # warm-up
    t0 = time.time(); eng.random_candidates(m, d, bounds=bounds, rng=rs, out=out); eng.sync(); t_mt = time.time() - t0
    same = np.array_equal(out.download(), host)
    gen = np.random.Generator(np.random.Philox(key=5))
    ref = np.random.Generator(np.random.Philox(key=5)).random((m, d)) * 2.0 + 0.0
    eng.random_candidates(16, d, bounds=bounds, rng=np.random.Generator(np.random.Philox(key=6)), out=out)
    t0 = time.time(); eng.random_candidates(m, d, bounds=bounds, rng=gen, out=out); eng.sync(); t_ph = time.time() - t0
    same_ph = np.array_equal(out.download(), ref)
    print('m=%d d=%d (%.0f MB): host draw %.1f ms + upload %.1f ms | device MT19937 %.2f ms (identical: %s) | '
          'device Philox %.2f ms = %.0f GB/s (identical: %s)'
          % (m, d, m * d * 8 / 1e6, t_gen * 1e3, t_up * 1e3, t_mt * 1e3, same, t_ph * 1e3,
             m * d * 8 / t_ph / 1e9, same_ph), flush=True)
    out.free()
  print('STAGE rng DONE')


def stage_pdoo():
  """ PDOO acquisition maximisation: one point per device call (the reference's pattern) vs a frontier per call """
  from dragonfly_amd.engine import Engine, KernelSpec
  from dragonfly_amd.doo import pdoo_maximise_batched
  eng = Engine()
  for n, d, budget in ((4096, 6, 2000), (16384, 32, 2000)):
    rs = np.random.RandomState(7)
    X = rs.random_sample((n, d)); Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
    spec = KernelSpec('se', d, float(Y.var()), 0.3 * np.sqrt(d) * np.ones(d))
    gp = eng.gp_fit(spec, X, Y - np.median(Y), float(Y.var() / 20))
    def ucb(pts):
      mu, sd = gp.predict(pts)
      return mu + 2.0 * sd
    bounds = [[0.0, 1.0]] * d
    res = {}
    for frontier, depth in ((0, 0), (32, 2), (128, 3)):
      t0 = time.time()
      v, p, h = pdoo_maximise_batched(ucb, bounds, budget, frontier=frontier, depth=depth, return_history=True)
      res[frontier] = (time.time() - t0, v, p, h)
    same = all(res[f][1] == res[0][1] and np.array_equal(res[f][2], res[0][2]) for f in res)
    print('n=%d d=%d budget=%d (%d callbacks in the reference): ' % (n, d, budget, res[0][3].points_requested)
          + ' | '.join('frontier %d: %.1f ms, %d calls, %d points' % (f, res[f][0] * 1e3, res[f][3].device_calls,
                                                                     res[f][3].points_evaluated) for f in res)
          + ' | identical: %s' % same, flush=True)
    gp.free()
  print('STAGE pdoo DONE')


def stage_slice():
  """ slice sampling of one hyper-parameter (log bandwidth) with the log marginal likelihood as density:
      one fit per density call (the reference's pattern) vs speculative batches """
  from dragonfly_amd.engine import Engine, KernelSpec
  from dragonfly_amd.slice_sampler import SpeculativeSlice
  eng = Engine()
  d = 4
  for n in (50, 200, 1000, 4096):
    rs = np.random.RandomState(3)
    X = rs.random_sample((n, d)); Y = np.sin(3 * X.sum(axis=1)) + 0.05 * rs.randn(n)
    Xd = eng.to_device(X)
    yc = Y - np.median(Y)
    spec_of = lambda lb: KernelSpec('se', d, float(Y.var()), float(np.exp(lb)) * np.ones(d))
    def one_by_one(xs):
      out = []
      for lb in xs:
        gp = eng.gp_fit(spec_of(lb), Xd, yc, 0.01); out.append(gp.lml); gp.free()
      return out
    batched = lambda xs: eng.gp_lml_batch([spec_of(lb) for lb in xs], Xd, Y, [float(np.median(Y))] * len(xs), [0.01] * len(xs))
    iters = 60 if n <= 1000 else 20
    res = {}
    for label, fn, ahead in (('one fit per call', one_by_one, (1, 1)), ('batched', batched, (3, 4))):
      sampler = SpeculativeSlice(fn, ahead_step=ahead[0], ahead_shrink=ahead[1])
      np.random.seed(5)
      sampler.sample(-1.0, 3, 2)                      # warm-up
      sampler = SpeculativeSlice(fn, ahead_step=ahead[0], ahead_shrink=ahead[1])
      np.random.seed(5)
      t0 = time.time(); chain = sampler.sample(-1.0, iters, 0); dt = time.time() - t0
      res[label] = (dt / iters * 1e3, chain, sampler)
    a, b = res['one fit per call'], res['batched']
    print('n=%d: %.2f density values per update | one fit per value %.2f ms/update | speculative batches %.2f ms/update '
          '(%.1f calls, %.1f values per update) | max chain difference %.1e'
          % (n, a[2].consumed / float(iters), a[0], b[0], b[2].batches / float(iters), b[2].evaluated / float(iters),
             float(np.max(np.abs(a[1] - b[1])))), flush=True)
  print('STAGE slice DONE')


STAGES = ['gemm', 'kernmat', 'chol', 'gp', 'perf', 'fit16k', 'hptune', 'append', 'configs', 'rng', 'pdoo', 'slice']

if __name__ == '__main__':
  if len(sys.argv) == 3 and sys.argv[1] == '--run':
    globals()['stage_' + sys.argv[2]]()
    sys.exit(0)
  stages = sys.argv[1:] or STAGES
  outdir = os.path.join(ROOT, 'gpurun_out', 'check')
  os.makedirs(outdir, exist_ok=True)
  for st in stages:
    t0 = time.time()
    with open(os.path.join(outdir, st + '.log'), 'w') as f:
      try:
        rc = subprocess.call([sys.executable, os.path.abspath(__file__), '--run', st], stdout=f,
                             stderr=subprocess.STDOUT, timeout=900)
      except subprocess.TimeoutExpired:
        rc = 'TIMEOUT'
    print('== stage %s rc=%s (%.1fs)' % (st, rc, time.time() - t0), flush=True)
    with open(os.path.join(outdir, st + '.log')) as f:
      txt = f.read()
    print(txt[-6000:], flush=True)
