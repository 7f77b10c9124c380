"""bench.py -- GP-fit + acquisition-batch time at n = 16384, d = 32 on N MI355X GPUs.

    python bench.py [--gpus N --steps K --warmup W]
        ONE process drives N devices through libdfhip.so's in-library fan-out (dfh_mgpu_*: a
        context + host thread per device, RCCL communicator clique) -- no launcher, no PyTorch.
        Fails loudly if fewer than N GPUs are visible.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        one process per GPU (the launcher only provides RANK / LOCAL_RANK / WORLD_SIZE); the ranks
        form an RCCL communicator through dfh_comm_* (unique id passed through a file), again
        without PyTorch in the processes.  WORLD_SIZE must equal --gpus.

One "step" = one pass of Dragonfly's GP hot path over one batch of synthetic input, everything
already resident in HBM when the timed region starts:
    fit   : kernel matrix K(X,X) (SE-ARD) -> K + noise I -> blocked Cholesky -> alpha -> lml
            (GP.build_posterior + compute_log_marginal_likelihood, BASELINE config 3)
    batch : blocked-joint Thompson sampling (block 4096) over this rank's 262144 candidates and the
            arg-max of the draw (asy_ts, BASELINE config 4: 2 097 152 candidates over 8 GPUs);
            rank r holds rows [r*262144, (r+1)*262144) of the ONE seed-204 candidate set
Weak scaling: per-GPU candidates are fixed; every rank fits the (replicated) GP -- the n = 16384
fit does not shard profitably -- and the only exchange is the RCCL
all-gather of one (value, index) pair per rank.  `value` is the step time in ms (max over ranks).

Also on the JSON line: `roofline` for the dominant kernel (the fp64 MFMA GEMM behind Cholesky
SYRK/TRSM, posterior TRSM and TS SYRK) from per-launch HIP events recorded on the launch streams;
`cpu_baseline`: the NumPy oracle (a port of the reference's CPU path) timed on this host's cores --
the full n = 16384 fit and three full Thompson blocks; `parity_vs_oracle`: the device results of
the same inputs against that oracle run; `configs`: BASELINE configs 2 and 5, untimed extras.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import bench_configs as BC      # noqa: E402  pylint: disable=wrong-import-position
import bench_extras as BX       # noqa: E402  pylint: disable=wrong-import-position

N_TRAIN, DIM, TS_BLOCK, CANDS_PER_GPU = BC.N_TRAIN, BC.DIM, BC.TS_BLOCK, BC.CANDS_PER_GPU
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense fp64 matrix peak: 256 CU x 128 flop/clk x 2.4 GHz
HBM_PEAK_TBS = 8.0
# vector fp64: 256 CU x 64 lanes x 2.4 GHz instructions/s (an FMA counted once; it shares the pipe with the MFMAs)
FP64_VALU_PEAK_TOPS = 39.3
VALU_OPS_PER_ELEMENT = {'se': 28.0, 'matern': 45.0}     # epilogue of one kernel-matrix element


def fp64_pipe_frac(kind, packed_width, elements, parts_per_element, ms):
  """ Kernel-matrix kernels whose epilogue (exp / Matern polynomial per element and part) outweighs their
      HBM traffic are bound by the fp64 pipe, which MFMA and VALU share: the time the pipe needs at its
      peaks, -2 X1 X2^T on the matrix cores + the epilogue on the vector unit, over the measured time. """
  t_mfma = 2.0 * packed_width * elements / (FP64_MFMA_PEAK_TFLOPS * 1e12)
  t_valu = VALU_OPS_PER_ELEMENT[kind] * parts_per_element * elements / (FP64_VALU_PEAK_TOPS * 1e12)
  return {'bound': 'fp64 pipe (valu_f64 epilogue + MFMA share it)', 'min_ms_mfma': round(t_mfma * 1e3, 4),
          'min_ms_valu': round(t_valu * 1e3, 4), 'frac_of_fp64_pipe_peak': round((t_mfma + t_valu) / (ms * 1e-3), 4),
          'frac_of_valu_peak': round(t_valu / (ms * 1e-3), 4)}
# kernel-matrix build of the fit: SURVEY 8d's bytes, and the bytes the lower-triangle-only build moves (tiles of 64 x 64)
KM_BYTES_8D = 8 * (N_TRAIN ** 2 + 2 * N_TRAIN * DIM)
_KM_T = (N_TRAIN + 63) // 64
KM_BYTES_MOVED = (8 * (64 * 64 * (_KM_T * (_KM_T + 1) // 2) + 2 * N_TRAIN * DIM)
                  if os.environ.get('DFH_KM_LOWER_ONLY', '1') != '0' else KM_BYTES_8D)
def km_bytes_moved(n, d):
  """ bytes the fit's kernel-matrix build moves: the 64 x 64 tiles on and below the diagonal (n >= 2048) or the whole matrix """
  if n >= 2048 and os.environ.get('DFH_KM_LOWER_ONLY', '1') != '0':
    t = (n + 63) // 64
    return 8.0 * (64 * 64 * (t * (t + 1) // 2) + 2 * n * d)
  return 8.0 * (n * n + 2 * n * d)


PMC_TRAFFIC_FILES = ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json')


def rel(a, b):
  a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
  den = float(np.max(np.abs(b))) if b.size else 1.0
  return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


ELEM_MASK = 1e-3


def rel_elem(a, b):
  """ element-wise relative error (hard part 1): max |a_i - b_i| / |b_i| over the entries
      with |b_i| >= ELEM_MASK * max|b| (entries near zero have no meaningful relative error) """
  a, b = np.asarray(a, dtype=float).ravel(), np.asarray(b, dtype=float).ravel()
  if b.size == 0:
    return 0.0
  keep = (np.abs(b) >= ELEM_MASK * np.max(np.abs(b))) & (b != 0)
  return float(np.max(np.abs(a[keep] - b[keep]) / np.abs(b[keep]))) if keep.any() else 0.0


def both(a, b):
  """ [norm-wise, element-wise] """
  return [rel(a, b), rel_elem(a, b)]


# ---- the two ways to run N GPUs --------------------------------------------------------------
class InProcess(object):
  """ one process, N devices: parallel.MultiEngine (dfh_mgpu_*) """
  mode = 'one process, in-library fan-out (dfh_mgpu_*: host thread + context per device, ncclCommInitAll)'

  def __init__(self, n_gpus, prob, spec, cpg):
    from dragonfly_amd import parallel
    self.world, self.rank, self.cpg = n_gpus, 0, cpg
    from dragonfly_amd import _lib
    ids = None
    if os.environ.get('DFH_MGPU_ALLOW_DUPLICATE_DEVICES', '0') not in ('', '0') and _lib.device_count() < n_gpus:
      # TEST MODE, a box with fewer GPUs than ranks: contexts share devices, the pairs are reduced on the host
      # (RCCL refuses duplicate devices) -- a dry run of this route's plumbing, not a measurement
      ids = [r % _lib.device_count() for r in range(n_gpus)]
      self.mode = 'DRY RUN: %d contexts sharing %d device(s), host reduce instead of RCCL' % (n_gpus, _lib.device_count())
    self.mg = parallel.MultiEngine(n_gpus, device_ids=ids)        # raises if fewer GPUs are visible (outside the test mode)
    self.eng0 = self.mg.engines[0]
    self.spec, self.prob = spec, prob
    self.Xd = [e.to_device(prob['X']) for e in self.mg.engines]
    self.yd = [e.to_device(prob['Y'] - prob['mean_c']) for e in self.mg.engines]
    self.cd, self.ud = [], []
    for r, e in enumerate(self.mg.engines):
      cands, U = BC.config4_rows(r * cpg, (r + 1) * cpg)
      if r == 0:
        self.cands0, self.U0 = cands, U
      self.cd.append(e.to_device(cands))
      self.ud.append(e.to_device(U))
    self.result = {}

  def step(self):
    lml = self.mg.fit(self.spec, self.Xd, self.yd, self.prob['noise'])
    v, i = self.mg.thompson(self.cd, self.ud, block=TS_BLOCK, mean_const=self.prob['mean_c'])
    self.result.update(lml=lml[0], best=v, idx=i)

  def sync(self):
    self.mg.sync()

  def max_over_ranks(self, x):
    return x

  def comm_info(self):
    return self.mg.comm_info(0)

  def close(self):
    self.mg.free_fit()
    self.mg.close()


class PerProcess(object):
  """ one process per GPU (launcher): Engine + parallel.RcclComm (dfh_comm_*) """
  mode = 'one process per GPU (launcher env), dfh_comm_*: ncclGetUniqueId by file rendezvous, ncclCommInitRank'

  def __init__(self, n_gpus, prob, spec, cpg):
    from dragonfly_amd import parallel
    from dragonfly_amd.engine import Engine
    self.world, self.cpg = n_gpus, cpg
    self.rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', str(self.rank)))
    from dragonfly_amd import _lib
    shared = (os.environ.get('DFH_MGPU_ALLOW_DUPLICATE_DEVICES', '0') not in ('', '0')
              and n_gpus > 1 and _lib.device_count() < n_gpus)
    if shared:
      # TEST MODE, a box with fewer GPUs than ranks: the ranks share devices and exchange through the host
      # (RCCL refuses two ranks on one device) -- a dry run of this route's plumbing, not a measurement
      self.eng0 = Engine(local_rank % _lib.device_count())
      self.comm = parallel.HostExchangeComm.from_env()
      self.mode = 'DRY RUN: %d processes sharing %d device(s), host-file exchange instead of RCCL' % (n_gpus, _lib.device_count())
    else:
      self.eng0 = Engine(local_rank)
      self.comm = parallel.RcclComm.from_env(self.eng0)
    self.spec, self.prob = spec, prob
    self.Xd = self.eng0.to_device(prob['X'])
    self.yd = self.eng0.to_device(prob['Y'] - prob['mean_c'])
    self.cands0, self.U0 = BC.config4_rows(self.rank * cpg, (self.rank + 1) * cpg)
    self.cd, self.ud = self.eng0.to_device(self.cands0), self.eng0.to_device(self.U0)
    self.result = {}

  def step(self):
    gp = self.eng0.gp_fit(self.spec, self.Xd, self.yd, self.prob['noise'])
    v, i = gp.thompson(self.cd, self.ud, block=TS_BLOCK, mean_const=self.prob['mean_c'])
    v, i = self.comm.allgather_argmax(v, i + self.rank * self.cpg)
    self.result.update(lml=gp.lml, best=v, idx=i)
    gp.free()

  def sync(self):
    self.eng0.sync()
    self.comm.barrier()

  def max_over_ranks(self, x):
    return float(self.comm.allreduce_max([x])[0])

  def comm_info(self):
    return self.comm.info()

  def close(self):
    self.comm.barrier()
    self.comm.close()


# ---- CPU baseline + parity ---------------------------------------------------------------------
def _blas_threads():
  try:
    import threadpoolctl
    return max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] + [1])
  except Exception:    # pylint: disable=broad-except
    return os.cpu_count()


def oracle_stages(O, kern, X, Y, mean_c, noise, cand_blocks, U_blocks, posterior_chunks=()):
  """ The oracle (oracle/ref_numpy.py: the reference's NumPy/SciPy path restated) stage by stage:
      the fit, then per Thompson block cross matrices / triangular solve / covariance / block
      factorisation + draw.  Returns (timings, results). """
  t, out = {}, {}
  n = len(X)
  t0 = time.perf_counter()
  K = kern(X, X)                                                   # gp_core.py:157
  t['kernel'] = time.perf_counter() - t0
  t0 = time.perf_counter()
  L, power = O.stable_cholesky(K + noise * np.eye(n), return_power=True)     # gp_core.py:843
  del K
  t['chol'] = time.perf_counter() - t0
  t0 = time.perf_counter()
  yc = Y - mean_c
  alpha = O.solve_upper_triangular(L.T, O.solve_lower_triangular(L, yc))     # gp_core.py:161-163
  lml = -0.5 * yc.T.dot(alpha) - (np.log(np.diag(L))).sum() - 0.5 * n * np.log(2 * np.pi)   # :224-226
  t['solve'] = time.perf_counter() - t0
  out.update(alpha=alpha, lml=float(lml), jitter_power=power, blocks=[])
  per_block = []
  for Xc, U in zip(cand_blocks, U_blocks):
    tb = {}
    t0 = time.perf_counter()
    K_tetr = kern(Xc, X)                                            # gp_core.py:172-174
    mu = mean_c + K_tetr.dot(alpha)
    K_tete = kern(Xc, Xc)
    tb['cross'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    V = O.solve_lower_triangular(L, K_tetr.T)                       # gp_core.py:180
    tb['trsm'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    cov = K_tete - V.T.dot(V)                                       # gp_core.py:181
    tb['syrk'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    Lc, pw = O.stable_cholesky(cov, return_power=True)              # general_utils.py:229
    s = (Lc.dot(U.reshape(-1, 1)).T + mu).ravel()                   # general_utils.py:231
    arg = int(s.argmax())
    tb['blockchol'] = time.perf_counter() - t0
    per_block.append(tb)
    out['blocks'].append(dict(mu=mu, sd=np.sqrt(np.diag(cov)), draw=s, argmax=arg, jitter_power=pw))
  t['blocks'] = per_block
  # posterior mean / std only (GP.eval(X, 'std'), gp_core.py:165-190) on further candidate chunks
  out['posterior'] = []
  t0 = time.perf_counter()
  for Xc in posterior_chunks:
    K_tetr = kern(Xc, X)
    V = O.solve_lower_triangular(L, K_tetr.T)
    cov = kern(Xc, Xc) - V.T.dot(V)
    out['posterior'].append(dict(mu=mean_c + K_tetr.dot(alpha), sd=np.sqrt(np.diag(cov))))
  t['posterior'] = time.perf_counter() - t0
  return t, out


class _ReferenceFunctions(object):
  """ Dragonfly's own functions behind the names oracle_stages() calls (used when the reference is
      importable: DRAGONFLY_REFERENCE points at a checkout -- never the case on the driver's box). """

  def __init__(self, general_utils):
    self.gu = general_utils
    self.solve_lower_triangular = general_utils.solve_lower_triangular      # general_utils.py:214-216
    self.solve_upper_triangular = general_utils.solve_upper_triangular      # general_utils.py:218-220

  def stable_cholesky(self, M, return_power=False):
    L = self.gu.stable_cholesky(M)                                          # general_utils.py:166-204
    return (L, None) if return_power else L       # the reference does not say which power it used


def cpu_functions(prob):
  """ (functions, kernel callable, kind): the reference's own kernel object and linear-algebra helpers
      when DRAGONFLY_REFERENCE names a checkout that imports ('reference'), else the oracle's
      restatement of them ('port'). """
  ref_root = os.environ.get('DRAGONFLY_REFERENCE', '')
  if ref_root and os.path.isdir(os.path.join(ref_root, 'dragonfly')):
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import make_golden                                    # pylint: disable=import-outside-toplevel,import-error
    make_golden.import_reference()                        # NumPy-2 shim + import; never edits the reference
    from dragonfly.gp import kernel as ref_kernel         # pylint: disable=import-outside-toplevel,import-error
    from dragonfly.utils import general_utils as ref_gu   # pylint: disable=import-outside-toplevel,import-error
    kern = ref_kernel.SEKernel(DIM, prob['scale'], prob['bw'])              # gp/kernel.py:143-177
    return _ReferenceFunctions(ref_gu), kern, 'reference', 'dragonfly (SEKernel, stable_cholesky, solve_*_triangular) from %s' % ref_root
  from oracle import ref_numpy as O                       # pylint: disable=import-outside-toplevel
  return O, O.KernelSpec('se', DIM, prob['scale'], prob['bw']), 'port', 'oracle/ref_numpy.py'


def cpu_baseline_and_parity(prob, cands0, U0, eng, spec, cpg=CANDS_PER_GPU):
  """ cpu_baseline: the oracle on this host's cores -- the FULL n = 16384 fit once and three full
      Thompson blocks of 4096 candidates (median block time x 64 blocks; the blocks are
      independent and identical in work).  parity_vs_oracle: the device on the same inputs. """
  X, Y, mean_c, noise = prob['X'], prob['Y'], prob['mean_c'], prob['noise']
  O, kern, kind, source = cpu_functions(prob)
  nb = 3
  cb = [cands0[b * TS_BLOCK:(b + 1) * TS_BLOCK] for b in range(nb)]
  ub = [U0[b * TS_BLOCK:(b + 1) * TS_BLOCK] for b in range(nb)]
  kern(X[:256], X[:256]); np.linalg.cholesky(np.eye(64))          # warm-up (first BLAS call)
  # config 3's candidate stage (posterior over RandomState(203).random_sample((65536, 32))): the oracle on the
  # first and the last 4096 of them (its O(m^2) covariance forces chunks anyway), the device on all 65536
  Xs3 = BC.config3_candidates(65536)
  pc = [Xs3[:TS_BLOCK], Xs3[-TS_BLOCK:]]
  t_all0 = time.perf_counter()
  t, ref = oracle_stages(O, kern, X, Y, mean_c, noise, cb, ub, posterior_chunks=pc)
  measured_s = time.perf_counter() - t_all0 - t['posterior']
  fit_s = t['kernel'] + t['chol'] + t['solve']
  block_s = sorted(sum(tb.values()) for tb in t['blocks'])
  n_blocks = cpg // TS_BLOCK
  full_ms = (fit_s + n_blocks * block_s[len(block_s) // 2]) * 1e3
  # one BLAS thread: a smaller sample (the full step would take ~an hour), scaled by algorithmic work
  single = None
  try:
    import threadpoolctl
    n1, b1 = 4096, 1024
    with threadpoolctl.threadpool_limits(limits=1):
      t1, _ = oracle_stages(O, kern, X[:n1], Y[:n1], mean_c, noise, [cands0[:b1]], [U0[:b1]])
    rn, rb = N_TRAIN / float(n1), TS_BLOCK / float(b1)
    tb = t1['blocks'][0]
    fit1 = t1['kernel'] * rn ** 2 + t1['chol'] * rn ** 3 + t1['solve'] * rn ** 2
    blk1 = tb['cross'] * rn * rb + tb['trsm'] * rn ** 2 * rb + tb['syrk'] * rn * rb ** 2 + tb['blockchol'] * rb ** 3
    single = {'value': round((fit1 + n_blocks * blk1) * 1e3, 1), 'unit': 'ms', 'cores': 1,
              'sample': 'fit at n=%d + one TS block of %d with one BLAS thread (threadpoolctl), each stage '
                        'scaled to n=%d, block=%d by its algorithmic work (n^2 kernel, n^3 chol, n^2 b trsm, '
                        'n b^2 syrk, b^3 block chol)' % (n1, b1, N_TRAIN, TS_BLOCK),
              'measured_sample_s': round(t1['kernel'] + t1['chol'] + t1['solve'] + sum(tb.values()), 3)}
  except Exception as e:    # pylint: disable=broad-except
    single = {'error': repr(e)}
  cpu = {
    'value': round(full_ms, 1), 'unit': 'ms', 'cores': int(_blas_threads()), 'kind': kind,
    'sample': (source + ' (NumPy %s): the full fit at n=%d (kernel matrix, Cholesky, alpha, lml) '
               'measured once + %d full Thompson blocks of %d candidates at n=%d; value = fit + %d x median '
               'block time (blocks are independent, equal work); %.1f s of CPU work measured'
               % (np.__version__, N_TRAIN, nb, TS_BLOCK, N_TRAIN, n_blocks, measured_s)),
    'measured_s': {'kernel': round(t['kernel'], 3), 'chol': round(t['chol'], 3), 'solve': round(t['solve'], 3),
                   'blocks': [{k: round(v, 3) for k, v in tb.items()} for tb in t['blocks']]},
    'host_cpus': os.cpu_count(), 'single_thread': single,
  }
  # ---- the device on the same inputs ----
  gp = eng.gp_fit(spec, X, Y - mean_c, noise)
  m3 = nb * TS_BLOCK
  mu, sd = gp.predict(cands0[:m3])
  mu = mu + mean_c
  _, _, samp, jps = gp.thompson(cands0[:m3], U0[:m3], block=TS_BLOCK, mean_const=mean_c, return_samples=True)
  cat = lambda key: np.concatenate([b[key] for b in ref['blocks']])
  alpha_d = gp.get_alpha()
  par = {
    'what': 'device vs oracle on the bench inputs: n=%d fit; mu / sd / joint TS draw on the first %d candidates '
            '(blocks of %d); *_rel: norm-wise max|a-b|/max|b|; *_rel_elem: element-wise max|a_i-b_i|/|b_i| over the '
            'entries with |b_i| >= %g max|b|' % (N_TRAIN, m3, TS_BLOCK, ELEM_MASK),
    'lml_rel': abs(gp.lml - ref['lml']) / abs(ref['lml']),
    'alpha_rel': rel(alpha_d, ref['alpha']), 'alpha_rel_elem': rel_elem(alpha_d, ref['alpha']),
    'mu_rel': rel(mu, cat('mu')), 'mu_rel_elem': rel_elem(mu, cat('mu')),
    'sd_rel': rel(sd, cat('sd')), 'sd_rel_elem': rel_elem(sd, cat('sd')),
    'ts_draw_rel': rel(samp, cat('draw')), 'ts_draw_rel_elem': rel_elem(samp, cat('draw')),
    'ts_argmax_equal': [int(np.argmax(samp[b * TS_BLOCK:(b + 1) * TS_BLOCK])) == ref['blocks'][b]['argmax']
                        for b in range(nb)],
    'jitter_power_fit': [gp.jitter_power, ref['jitter_power']],
    'refine_steps_per_block': gp.refine_steps(),
    'jitter_power_blocks': [list(jps), [b['jitter_power'] for b in ref['blocks']]],
  }
  # ---- config 3's posterior over all 65536 seed-203 candidates: device time, parity on the oracle's chunks ----
  Xs3d = eng.to_device(Xs3)
  gp.predict(Xs3d)
  eng.sync()
  ts = []
  for _ in range(3):
    t0 = time.perf_counter()
    mu3, sd3 = gp.predict(Xs3d)
    eng.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
  Xs3d.free()
  mu3 = mu3 + mean_c
  ms3 = sorted(ts)[1]
  mu_o = np.concatenate([b['mu'] for b in ref['posterior']])
  sd_o = np.concatenate([b['sd'] for b in ref['posterior']])
  pick = np.r_[0:TS_BLOCK, len(Xs3) - TS_BLOCK:len(Xs3)]
  c3 = {'workload': 'config 3 candidate stage: posterior mean + std (GP.eval(X, "std"), gp_core.py:165-190) over '
                    'RandomState(203).random_sample((65536, 32)) at n=16384; results copied to the host',
        'm': int(len(Xs3)), 'ms': round(ms3, 3), 'candidates_per_s': round(len(Xs3) / (ms3 * 1e-3), 1),
        'frac_of_fp64_mfma_peak_incl_cross_matrix_and_copies': round(float(N_TRAIN) ** 2 * len(Xs3) / (ms3 * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
        'parity_rows': 'the first and the last %d candidates (the oracle needs %.1f s for them)' % (TS_BLOCK, t['posterior']),
        'mu_rel': both(mu3[pick], mu_o), 'sd_rel': both(sd3[pick], sd_o), 'rel_format': '[norm-wise, element-wise]'}
  gp.free()
  return cpu, par, c3


def accuracy_vs_truth(eng, prob, spec, n=4096, m=1024):
  """ SURVEY.md section 7, hard part 1(c): device AND oracle against the extended-precision truth
      (oracle/ld_truth.c: x87 long double, rounded once), norm-wise and element-wise.  The truth costs O(n^3)
      long-double operations, so this runs on the first n training points of config 3 (same kernel, noise and
      mean) with m seed-203 candidates and one joint Thompson draw over them. """
  from oracle import ref_longdouble as T
  from oracle import ref_numpy as O
  X, Y, mean_c, noise = prob['X'][:n], prob['Y'][:n], prob['mean_c'], prob['noise']
  Xs = BC.config3_candidates(m)
  U = np.random.RandomState(304).standard_normal(m)
  og = O.GPOracle(X, Y, O.KernelSpec('se', DIM, prob['scale'], prob['bw']), mean_c, noise)
  mu_o, sd_o = og.eval(Xs, 'std')
  draw_o = og.draw_samples_blocked(Xs, U, m)
  _, cov_o = og.eval(Xs, 'covar')
  _, pw = O.stable_cholesky(cov_o, return_power=True)
  jit = 0.0 if pw is None else (10.0 ** pw) * float(np.diag(cov_o).max())
  t0 = time.perf_counter()
  tr = T.gp_truth('se', prob['bw'], prob['scale'], X, Y - mean_c, noise, Xs, mean_c, 0.0, ts_normals=U, ts_jitter=jit)
  truth_s = time.perf_counter() - t0
  gp = eng.gp_fit(spec, X, Y - mean_c, noise)
  mu_d, sd_d = gp.predict(Xs)
  mu_d = mu_d + mean_c
  _, _, draw_d, pw_d = gp.thompson(Xs, U, block=m, mean_const=mean_c, return_samples=True)
  dev = dict(alpha=gp.get_alpha(), lml=[gp.lml], mu=mu_d, sd=sd_d, ts_draw=draw_d)
  orc = dict(alpha=og.alpha, lml=[og.lml()], mu=mu_o, sd=sd_o, ts_draw=draw_o)
  tru = dict(alpha=tr['alpha'], lml=[tr['lml']], mu=tr['mu'], sd=tr['sd'], ts_draw=tr['draw'])
  gp.free()
  out = {'what': 'first %d training points of config 3, %d seed-203 candidates, one joint Thompson draw; truth = the same '
                 'mathematics in x87 long double (%.1f s); every entry [norm-wise, element-wise over |b_i| >= %g max|b|]'
                 % (n, m, truth_s, ELEM_MASK),
         'ts_jitter_power': [None if pw_d[0] is None else int(pw_d[0]), pw]}
  for key in ('alpha', 'lml', 'mu', 'sd', 'ts_draw'):
    out[key] = {'device_vs_oracle': both(dev[key], orc[key]), 'device_vs_truth': both(dev[key], tru[key]),
                'oracle_vs_truth': both(orc[key], tru[key])}
  return out


# ---- other BASELINE configs (untimed extras) ---------------------------------------------------
def other_configs(eng):
  """ BASELINE configs 2 and 5 on one GPU: fit + candidate stage, inputs resident in HBM, median of
      three; the posterior TRSM's share of the fp64 MFMA peak and the kernel-matrix build's share of
      the HBM peak from the section timers (HIP events) of one extra pass. """
  from dragonfly_amd.engine import KernelSpec
  out = {}

  def timed(fn, reps=3):
    fn()
    eng.sync()
    ts = []
    for _ in range(reps):
      t0 = time.perf_counter()
      fn()
      eng.sync()
      ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]

  def sections(fn):
    eng.timings(True)
    fn()
    eng.sync()
    return eng.timings(False)

  c = BC.config2()
  spec = KernelSpec('matern', c['d'], c['scale'], c['bw'], nu=c['nu'])
  Xd, yd, cd = eng.to_device(c['X']), eng.to_device(c['Y'] - c['mean_c']), eng.to_device(c['cands'])
  box = {}

  def c2():
    gp = eng.gp_fit(spec, Xd, yd, c['noise'])
    box['r'] = gp.acq_argmax('ei', cd, params=(c['best'], 0.0), mean_const=c['mean_c'])
    gp.free()
  ms = timed(c2)
  s = sections(c2)
  n, m = c['n'], c['m']
  out['C2'] = {
    'workload': 'Hartmann6 n=4096 d=6 Matern-2.5: fit + EI arg-max over 65536 candidates',
    'ms': round(ms, 3), 'sections_ms': {k: round(v, 3) for k, v in s.items() if v > 0},
    'trsm_frac_of_fp64_mfma_peak': round(float(n) * n * m / (s['trsm'] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
    # (the fit writes the lower triangle's 64 x 64 tiles only from n = 2048 on: bytes actually moved, then SURVEY 8d's whole matrix)
    'kernel_matrix_frac_of_hbm_peak': round(km_bytes_moved(n, c['d']) / (s['kernmat'] * 1e-3) / 1e12 / HBM_PEAK_TBS, 4),
    'kernel_matrix_frac_by_section8d_bytes': round(8.0 * (n * n + 2 * n * c['d']) / (s['kernmat'] * 1e-3) / 1e12 / HBM_PEAK_TBS, 4),
    'cross_matrix_frac_of_hbm_peak': round(8.0 * (float(m) * n + (m + n) * c['d']) / (s['cross'] * 1e-3) / 1e12 / HBM_PEAK_TBS, 4),
    'chol_frac_of_fp64_mfma_peak': round(float(n) ** 3 / 3 / (s['chol'] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
    'argmax': int(box['r'][1]),
  }
  for a in (Xd, yd, cd):
    a.free()

  c = BC.config5()
  G = c['G']
  spec = KernelSpec('additive', c['d'], c['scale'], groups=c['groups'], sub_kinds=['se'] * G,
                    sub_scales=[1.0] * G, sub_nus=[0.0] * G, sub_bandwidths=c['bws'])
  Xd, yd = eng.to_device(c['X']), eng.to_device(c['Y'] - c['mean_c'])
  flat = eng.to_device(np.concatenate([x.ravel() for x in c['cands']]))
  sizes = [c['m_j']] * G
  beta = float(np.sqrt(0.2 * 5 * np.log(2 * 5 * c['n'] + 1)))      # gpb_acquisitions.py:135-137, t = n

  def c5():
    gp = eng.gp_fit(spec, Xd, yd, c['noise'])
    box['r'] = gp.add_ucb_all([beta] * G, flat, sizes=sizes)
    gp.free()
  ms = timed(c5)
  s = sections(c5)
  n, M = c['n'], c['m_j'] * G
  out['C5'] = {
    'workload': 'additive GP d=100, 20 groups x 5, n=4096: fit + add-UCB over 20 x 3276 candidates (one call)',
    'ms': round(ms, 3), 'sections_ms': {k: round(v, 3) for k, v in s.items() if v > 0},
    'trsm_frac_of_fp64_mfma_peak': round(float(n) * n * M / (s['trsm'] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
    'kernel_matrix_frac_of_hbm_peak': round(8.0 * n * n / (s['kernmat'] * 1e-3) / 1e12 / HBM_PEAK_TBS, 4),
    'chol_frac_of_fp64_mfma_peak': round(float(n) ** 3 / 3 / (s['chol'] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
    # twenty exponentials per element: this Gram matrix is bound by the vector fp64 unit, not by its 134 MB of HBM
    # writes (lower-triangle tiles are computed, n (n + 64) / 2 elements, and mirrored)
    'kernel_matrix_fp64_pipe': fp64_pipe_frac('se', 8 * G, n * (n + 64) / 2.0, G, s['kernmat']),
  }
  for a in (Xd, yd, flat):
    a.free()
  return out


def c4_shards_on_one_gpu(runner, eng, spec, prob, result):
  """ N = 1, strong scaling: the runner holds all 2 097 152 candidates of config 4 on the device. """
  from dragonfly_amd import parallel
  cd, ud, m8 = runner.cd[0], runner.ud[0], CANDS_PER_GPU
  shard = lambda r: (cd.view(r * m8 * DIM, (m8, DIM)), ud.view(r * m8, (m8,)))
  times = []
  for _ in range(3):
    eng.sync()
    t0 = time.perf_counter()
    gp = eng.gp_fit(spec, runner.Xd[0], runner.yd[0], prob['noise'])
    v0, i0 = gp.thompson(*shard(0), block=TS_BLOCK, mean_const=prob['mean_c'])[:2]
    eng.sync()
    times.append((time.perf_counter() - t0) * 1e3)
    if len(times) < 3:
      gp.free()
  vals, idxs = [v0], [int(i0)]
  for r in range(1, 8):
    v, i = gp.thompson(*shard(r), block=TS_BLOCK, mean_const=prob['mean_c'])[:2]
    vals.append(v); idxs.append(int(i) + r * m8)
  gp.free()
  v_red, i_red = parallel.reduce_argmax(vals, idxs)
  return {'workload': 'the eight 262144-candidate shards of config 4 (one per GPU of an 8-GPU run) evaluated one after '
                      'the other on this GPU and reduced like the ranks\' all-gather',
          'fit_plus_one_shard_ms': round(sorted(times)[1], 3),
          'reduced_argmax': int(i_red), 'reduced_best': v_red,
          'equals_timed_step_over_all_candidates': bool(int(i_red) == int(result['idx']) and v_red == result['best'])}


def pmc_traffic(scaling):
  """ HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes over
      this same command (profiles/rNN_pmc_traffic.json, written by tools/rocpd_pmc_traffic.py).
      Counters cannot be collected inside the timed run itself; the source file is named.  Only a record taken
      under the same --scaling mode counts: the launches of a weak-scaling step are an eighth as long. """
  for name in PMC_TRAFFIC_FILES:
    path = os.path.join(ROOT, 'profiles', name)
    try:
      with open(path) as f:
        rec = json.load(f)
        if rec.get('scaling', 'weak') != scaling:
          continue
        note = (' (%d launches per step in the counter passes: %s)' % (rec['launches'], rec['schedule_note'])) \
               if 'schedule_note' in rec else ''
        return rec['hbm_bytes_per_launch'], 'profiles/' + name + note
    except (OSError, KeyError, ValueError):
      continue
  return None, None


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3)
  # two by default: the ROCm runtime's one-off first-use work (code objects, queue resources: a 50 - 60 ms hole
  # in the first or second fit of a process, docs/NOTES_r03.md) should not land in the timed steps
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-extras', action='store_true')
  ap.add_argument('--no-accuracy', action='store_true', help='skip the device / oracle vs extended-precision truth comparison (~5 s)')
  ap.add_argument('--no-c4-full', action='store_true', help='skip the untimed 2 097 152-candidate single-GPU extra (~35 s)')
  ap.add_argument('--scaling', choices=('weak', 'strong'), default='strong',
                  help='strong (default): the 2 097 152 candidates of BASELINE config 4 split over the GPUs, the same problem '
                       'at every N; weak: 262144 candidates per GPU (config 4\'s per-GPU shard at every N)')
  args = ap.parse_args()

  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC (RCCL across processes)
  world_env = int(os.environ.get('WORLD_SIZE', '1'))
  if world_env > 1 and world_env != args.gpus:
    raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d processes' % (args.gpus, world_env))

  from dragonfly_amd.engine import KernelSpec
  prob = BC.config3()
  spec = KernelSpec('se', DIM, prob['scale'], prob['bw'])
  # DFH_BENCH_PER_PROCESS=1: take the process-per-GPU route even with one process (exercises
  # ncclGetUniqueId / file rendezvous / ncclCommInitRank on a one-GPU box)
  per_process = world_env > 1 or os.environ.get('DFH_BENCH_PER_PROCESS', '0') == '1'
  if args.scaling == 'strong' and (BC.CANDS_TOTAL_8 % (args.gpus * TS_BLOCK)) != 0:
    raise SystemExit('bench.py: --scaling strong needs --gpus to divide %d' % (BC.CANDS_TOTAL_8 // TS_BLOCK))
  cpg = CANDS_PER_GPU if args.scaling == 'weak' else BC.CANDS_TOTAL_8 // args.gpus
  runner = (PerProcess if per_process else InProcess)(args.gpus, prob, spec, cpg)
  rank, world, eng = runner.rank, runner.world, runner.eng0

  for _ in range(args.warmup):
    runner.step()
  runner.sync()
  eng.gemm_profile(enable=True, fetch=False)      # event pairs only, no host synchronisation
  t0 = time.perf_counter()
  for _ in range(args.steps):
    runner.step()
  runner.sync()
  elapsed = time.perf_counter() - t0
  gstats = eng.gemm_profile(enable=False, fetch=True)
  elapsed = runner.max_over_ranks(elapsed)
  ms_per_step = elapsed * 1e3 / args.steps
  # section breakdown from one extra, untimed step (section timers synchronise the host)
  eng.timings(True)
  runner.step()
  runner.sync()
  sections = eng.timings(False)
  result = dict(runner.result)

  out = None
  if rank == 0:
    g0 = gstats[0]                       # 128x128 NT tiles: the throughput configuration
    all_ms = sum(g['ms'] for g in gstats)
    all_flop = sum(g['flop'] for g in gstats)
    # launches on the look-ahead / pipeline streams overlap and then share the CUs: the kernel's
    # wall-clock is the union of its launch intervals (busy_ms), not the sum of their durations
    achieved = g0['flop'] / (g0['busy_ms'] * 1e-3) / 1e12 if g0['busy_ms'] > 0 else 0.0
    achieved_sum = g0['flop'] / (g0['ms'] * 1e-3) / 1e12 if g0['ms'] > 0 else 0.0
    traffic, traffic_src = pmc_traffic(args.scaling)
    out = {
      'metric': 'GP-fit+acq-batch ms at n=16384,d=32',
      'value': round(ms_per_step, 3), 'unit': 'ms', 'n_gpus': world, 'steps': args.steps,
      'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': False,
      'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
      'config': {'workload': ('C3 fit (n=16384, d=32, SE-ARD: kernel matrix + Cholesky + alpha + lml) '
                              '+ C4 %s: blocked-joint Thompson sampling, block 4096, over %d '
                              'candidates per GPU with arg-max' % ('shard' if args.scaling == 'weak' else 'whole, split over the GPUs', cpg)),
                 'n': N_TRAIN, 'd': DIM, 'candidates_per_gpu': cpg,
                 'candidates_total': cpg * world, 'ts_block': TS_BLOCK,
                 'candidate_rows': 'rank r: rows [r*%d, (r+1)*%d) of RandomState(204).random_sample((2097152, 32)), '
                                   'normals RandomState(304).standard_normal(2097152)' % (cpg, cpg),
                 'parallelism': 'candidate shards x%d, replicated fit, RCCL all-gather of (val,idx)' % world,
                 'launch': runner.mode},
      'candidates_per_s': round(cpg * world / (ms_per_step * 1e-3), 1),
      'sections_ms_extra_untimed_step_rank0': {k: round(v, 3) for k, v in sections.items() if v > 0},
      # the cross matrices K(X*, X) at d = 32 (kernmat_strip_kernel, posterior mean fused in): fp64-pipe-bound
      'cross_matrix_fp64_pipe': fp64_pipe_frac('se', DIM, float(cpg) * N_TRAIN, 1, sections['cross']) if sections.get('cross', 0) > 0 else None,
      'result': {'lml': result['lml'], 'ts_best': result['best'], 'ts_argmax': int(result['idx'])},
      'roofline': {
        'bound': 'mfma', 'kernel': 'gemm_f64_kernel<NT,128x128> (v_mfma_f64_16x16x4_f64)',
        'achieved': round(achieved, 2), 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
        'launches_per_step': g0['launches'] / args.steps,
        'avg_launch_us': round(g0['busy_ms'] * 1e3 / max(1, g0['launches']), 2),
        'avg_launch_us_incl_overlap': round(g0['ms'] * 1e3 / max(1, g0['launches']), 2),
        'achieved_from_sum_of_durations': round(achieved_sum, 2),
        'algorithmic_gflop_per_launch': round(g0['flop'] / max(1, g0['launches']) / 1e9, 3),
        'algorithmic_bytes_per_launch': round(g0['bytes'] / max(1, g0['launches'])),
        'all_gemm_variants': {'ms_per_step': round(all_ms / args.steps, 3),
                              'tflops': round(all_flop / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else 0.0},
        'side_targets': {
          'cholesky_frac_of_fp64_mfma_peak': round(N_TRAIN ** 3 / 3.0 / (sections['chol'] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)
                                             if sections.get('chol', 0) > 0 else None,
          # the fit path writes the lower triangle of the Gram matrix only (64 x 64 tiles on and below the diagonal:
          # the factorisation reads nothing else), so TWO figures: the fraction by the bytes actually moved -- the
          # kernel's roofline fraction -- and the one by SURVEY section 8d's formula 8 (n^2 + 2 n d), which counts the
          # whole matrix and is what earlier rounds' 0.59 - 0.62 were quoted on
          'kernel_matrix_frac_of_hbm_peak': round(KM_BYTES_MOVED / (sections['kernmat'] * 1e-3) / 1e12 / HBM_PEAK_TBS, 4)
                                            if sections.get('kernmat', 0) > 0 else None,
          'kernel_matrix_frac_by_section8d_bytes': round(KM_BYTES_8D / (sections['kernmat'] * 1e-3) / 1e12 / HBM_PEAK_TBS, 4)
                                                   if sections.get('kernmat', 0) > 0 else None,
          'kernel_matrix_bytes_moved': KM_BYTES_MOVED, 'kernel_matrix_bytes_section8d': KM_BYTES_8D,
          'kernel_matrix_lower_triangle_only': os.environ.get('DFH_KM_LOWER_ONLY', '1') != '0',
        },
      },
      'device': eng.name(),
      # factorisations that had to be repeated on the hand-off-free schedule so far in this process (expected: 0)
      'chol_fallbacks': eng.counters()['chol_fallbacks'],
      # the communicator as RCCL itself reports it (ncclCommCount / ncclCommUserRank / ncclGetVersion on rank 0):
      # a scaling record is checked for "RCCL formed N ranks" against ranks_formed, not against --gpus
      'comm': runner.comm_info(),
    }
    # north_star's two side targets and the fit's sections as flat top-level scalars (a driver that keeps only
    # scalars keeps these): from the extra, untimed step whose section timers synchronise the host
    st = out['roofline']['side_targets']
    out['cholesky_frac_of_fp64_mfma_peak'] = st['cholesky_frac_of_fp64_mfma_peak']
    out['kernel_matrix_frac_of_hbm_peak'] = st['kernel_matrix_frac_of_hbm_peak']
    out['kernel_matrix_frac_by_section8d_bytes'] = st['kernel_matrix_frac_by_section8d_bytes']
    for key in ('kernmat', 'chol', 'solve', 'cross', 'trsm', 'ts'):
      out[key + '_ms'] = round(sections[key], 4) if sections.get(key, 0) > 0 else None
    out['kernel_matrix_bytes_moved'] = KM_BYTES_MOVED
    out['kernel_matrix_bytes_section8d'] = KM_BYTES_8D
    if not args.no_extras and world == 1:       # (ranks of a multi-process run stay in step: no rank-0-only extras)
      # conditioning of the matrix that was factored (quoted next to the parity numbers):
      # lambda_max(K) by power iteration on the host, lambda_min(K + noise I) >= noise
      gp = eng.gp_fit(spec, prob['X'], prob['Y'] - prob['mean_c'], prob['noise'])
      Kh = gp.get_K()
      gp.free()
      v = np.ones(N_TRAIN) / np.sqrt(N_TRAIN)
      lam = 0.0
      for _ in range(12):
        w = Kh.dot(v)
        lam = float(np.linalg.norm(w))
        v = w / lam
      del Kh
      out['conditioning'] = {'lambda_max_K': round(lam, 3), 'noise_var': prob['noise'],
                             'cond_upper_bound': round((lam + prob['noise']) / prob['noise'], 1)}
      # untimed extra: drawing this rank's candidates (oper_utils.py:62 -- np.random.random((m, d))) on
      # the device instead of on the host + PCIe; inputs of the timed step are resident either way
      gen = {}
      buf = eng.empty((CANDS_PER_GPU, DIM))
      for label, rng in (('mt19937_continuing_numpy_state', np.random.RandomState(11)),
                         ('philox4x64', np.random.Generator(np.random.Philox(key=11)))):
        eng.random_candidates(64, DIM, rng=rng, out=buf)
        eng.sync()
        t0 = time.perf_counter()
        eng.random_candidates(CANDS_PER_GPU, DIM, rng=rng, out=buf)
        eng.sync()
        gen[label + '_ms'] = round((time.perf_counter() - t0) * 1e3, 3)
      t0 = time.perf_counter()
      host_draw = np.random.RandomState(11).random_sample((CANDS_PER_GPU, DIM))
      t1 = time.perf_counter()
      buf.upload(host_draw)
      eng.sync()
      gen['host_numpy_draw_ms'] = round((t1 - t0) * 1e3, 3)
      gen['host_upload_ms'] = round((time.perf_counter() - t1) * 1e3, 3)
      buf.free()
      # the TS normals (general_utils.py:230 -- np.random.normal(size=(m, 1))): device (bit for bit) vs host + upload
      nb = eng.empty((CANDS_PER_GPU,))
      rng = np.random.RandomState(12)
      eng.random_normals(64, rng=rng, out=nb)
      eng.sync()
      t0 = time.perf_counter()
      eng.random_normals(CANDS_PER_GPU, rng=rng, out=nb)
      eng.sync()
      gen['normals_mt19937_device_ms'] = round((time.perf_counter() - t0) * 1e3, 3)
      t0 = time.perf_counter()
      host_norm = np.random.RandomState(12).standard_normal(CANDS_PER_GPU)
      t1 = time.perf_counter()
      nb.upload(host_norm)
      eng.sync()
      gen['normals_host_numpy_draw_ms'] = round((t1 - t0) * 1e3, 3)
      gen['normals_host_upload_ms'] = round((time.perf_counter() - t1) * 1e3, 3)
      nb.free()
      out['candidate_generation_untimed_rank0'] = gen
      out['configs'] = other_configs(eng)
      # round 3: config 1 through the mirrors, the tuning / append / tree-search workloads and the whole of
      # config 4 on one GPU -- each with the oracle beside it and an equality / parity flag (bench_extras.py)
      # (the whole of config 4 on one GPU is the timed step itself under --scaling strong; its eight
      #  per-GPU shards -- what each rank of an 8-GPU run evaluates -- one after the other, reduced as the
      #  ranks' all-gather is: the same winner; the first shard timed, fit included = the weak-scaling step)
      if args.scaling == 'strong' and not args.no_c4_full and not per_process:
        try:
          out['configs']['C4_shards_on_1gpu'] = c4_shards_on_one_gpu(runner, eng, spec, prob, result)
        except Exception as e:      # pylint: disable=broad-except
          out['configs']['C4_shards_on_1gpu'] = {'error': repr(e)}
      out['configs'].update(BX.run_all(eng, prob, spec, include_c4_full=not args.no_c4_full and args.scaling == 'weak'))
    if not args.no_cpu_baseline and world == 1:
      out['cpu_baseline'], out['parity_vs_oracle'], c3 = cpu_baseline_and_parity(prob, runner.cands0, runner.U0, eng, spec, cpg)
      out.setdefault('configs', {})['C3_posterior'] = c3
      if not args.no_accuracy:
        out['accuracy_vs_truth'] = accuracy_vs_truth(eng, prob, spec)
    elif not args.no_cpu_baseline:
      out['cpu_baseline'] = None
  runner.close()
  if rank == 0:
    sys.stdout.flush()
    # Everything measured goes out on a line of its own that is NOT a JSON line ('BENCH_DETAILS ' in front; also
    # written to gpurun_out/bench_details.json when that directory exists); the ONE JSON line, last thing on stdout,
    # is the contract's line: short enough to survive a tail, with the side targets as scalars INSIDE `roofline`.
    # (whichever of the two lines a reader picks up, the side targets are scalars inside its `roofline`)
    contract = contract_line(out)
    for key, val in contract['roofline'].items():
      out['roofline'].setdefault(key, val)
    details = json.dumps(out)
    print('BENCH_DETAILS ' + details, flush=True)
    scratch = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(scratch):
      with open(os.path.join(scratch, 'bench_details.json'), 'w') as f:
        f.write(details + '\n')
    print(json.dumps(contract), flush=True)


def _dig(d, *path):
  for key in path:
    if not isinstance(d, dict) or key not in d:
      return None
    d = d[key]
  return d


def contract_line(out):
  """ The one JSON line of the bench contract, cut from the full record: the contract's keys, `roofline` with every
      side target as a SCALAR inside it (a record that keeps the scalars of `roofline` keeps north_star's two targets,
      the fit's sections, the other configs' factorisation / row-solve fractions and the tuning-call latencies),
      `cpu_baseline` without its per-section timings.  The full record is the BENCH_DETAILS line before it. """
  keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
          'vs_baseline', 'dtype', 'data')
  line = {k: out[k] for k in keep}
  cfg = dict(out['config'])
  cfg['workload'] = 'C3 fit (n=16384,d=32,SE-ARD) + C4 Thompson sampling, block 4096, %d candidates per GPU, arg-max' % cfg['candidates_per_gpu']
  cfg.pop('candidate_rows', None)
  cfg['launch'] = cfg['launch'].split(',')[0].split('(')[0].strip()
  line['config'] = cfg
  r = {k: v for k, v in out['roofline'].items() if not isinstance(v, dict)}
  r['traffic_source'] = (r.get('traffic_source') or '').split(' ')[0] or None
  st = out['roofline']['side_targets']
  for k in ('cholesky_frac_of_fp64_mfma_peak', 'kernel_matrix_frac_of_hbm_peak', 'kernel_matrix_frac_by_section8d_bytes',
            'kernel_matrix_bytes_moved', 'kernel_matrix_bytes_section8d'):
    r[k] = st.get(k)
  for k in ('kernmat', 'chol', 'solve', 'cross', 'trsm', 'ts'):
    r[k + '_ms'] = out.get(k + '_ms')
  r['comm_ranks_formed'] = _dig(out, 'comm', 'ranks_formed')
  r['chol_fallbacks'] = out.get('chol_fallbacks')
  cf = out.get('configs') or {}
  for c in ('C2', 'C5'):
    for k, short in (('ms', 'ms'), ('trsm_frac_of_fp64_mfma_peak', 'trsm_frac'), ('chol_frac_of_fp64_mfma_peak', 'chol_frac'),
                     ('kernel_matrix_frac_by_section8d_bytes', 'kernmat_frac_8d')):
      r['%s_%s' % (c, short)] = _dig(cf, c, k)
    for k in ('kernmat', 'chol', 'solve'):
      v = _dig(cf, c, 'sections_ms', k)
      r['%s_%s_ms' % (c, k)] = v
  for n, keys in (('n50', ('ms_batch_of_8', 'ms_one_candidate')), ('n200', ('ms_batch_of_8', 'ms_10000')),
                  ('n1000', ('ms_batch_of_64', 'ms_10000')), ('n2000', ('us_each_of_512',))):
    for k in keys:
      r['hp_%s_%s' % (n, k)] = _dig(cf, 'hp_tuning', n, k)
  r['hp_lml_rel_max'] = max([v for v in (_dig(cf, 'hp_tuning', n, 'lml_rel_max') for n in ('n50', 'n200', 'n1000', 'n2000', 'n4096'))
                            if v is not None] or [0.0]) if cf.get('hp_tuning') else None
  for n in ('n4096', 'n8192'):
    r['chol_%s_ms' % n] = _dig(cf, 'chol_sizes', n, 'ms')
    r['solve_%s_ms' % n] = _dig(cf, 'chol_sizes', n, 'solve_ms')
  line['roofline'] = r
  cb = out.get('cpu_baseline')
  if isinstance(cb, dict):
    cb = {k: cb.get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample')}
    cb['sample'] = (cb['sample'] or '')[:200]
  line['cpu_baseline'] = cb
  pv = out.get('parity_vs_oracle')
  if isinstance(pv, dict):
    line['parity_vs_oracle'] = {k: pv[k] for k in ('alpha_rel', 'lml_rel', 'mu_rel', 'sd_rel', 'ts_draw_rel', 'ts_argmax_equal') if k in pv}
  line['result'] = out.get('result')
  line['device'] = out.get('device')
  line['details'] = 'BENCH_DETAILS line above / gpurun_out/bench_details.json'
  return line


if __name__ == '__main__':
  main()
