"""bench.py -- GP-fit + acquisition-batch time at n = 16384, d = 32 on N MI355X GPUs.

    python bench.py [--gpus N --steps K --warmup W]                      (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of Dragonfly's GP hot path over one batch of synthetic input, everything
already resident in HBM when the timed region starts:
    fit   : kernel matrix K(X,X) (SE-ARD) -> K + noise I -> blocked Cholesky -> alpha -> lml
            (GP.build_posterior + compute_log_marginal_likelihood, BASELINE config 3)
    batch : blocked-joint Thompson sampling (block 4096) over this rank's 262144 candidates and the
            arg-max of the draw (asy_ts, BASELINE config 4: 2 097 152 candidates over 8 GPUs)
Weak scaling: per-GPU candidates are fixed; every rank fits the (replicated) GP -- the n = 16384
fit does not shard profitably -- and the only exchange is the all-gather of
one (value, index) pair per rank over RCCL.  `value` is the step time in ms (max over ranks).

Also reported: `roofline` for the dominant kernel (the fp64 MFMA GEMM behind Cholesky SYRK/TRSM,
posterior TRSM and TS SYRK) from per-launch HIP events recorded on the launch streams, and
`cpu_baseline`: the NumPy oracle (a port of the reference's CPU path) timed on this host's cores
on a bounded sample of the same workload and scaled by algorithmic work.
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

N_TRAIN, DIM, TS_BLOCK = 16384, 32, 4096
CANDS_PER_GPU = 262144
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X dense fp64 matrix peak: 256 CU x 128 flop/clk x 2.4 GHz


def make_problem(rank):
  """ X ~ U[0,1)^32 (seed 103), Y = sum_j j/d x_j^2 + noise,
      SE-ARD bandwidths 0.2 sqrt(32) (0.5 + j/32), mean = median(Y), noise = Var(Y)/20; candidates
      seed 204 (+rank: each rank owns its contiguous shard), normals seed 304 (+rank). """
  rs = np.random.RandomState(103)
  X = rs.random_sample((N_TRAIN, DIM))
  w = (np.arange(DIM) + 1.0) / DIM
  Y = (X ** 2).dot(w) + 0.01 * rs.randn(N_TRAIN)
  bw = 0.2 * np.sqrt(DIM) * (0.5 + np.arange(DIM) / 32.0)
  mean_c = float(np.median(Y))
  noise = float(Y.var() / 20)
  cands = np.random.RandomState(204 + rank).random_sample((CANDS_PER_GPU, DIM))
  U = np.random.RandomState(304 + rank).standard_normal(CANDS_PER_GPU)
  return X, Y, bw, mean_c, noise, cands, U


def cpu_baseline(X, Y, bw, mean_c, noise, budget_note=True):
  """ The oracle (NumPy/SciPy restatement of the reference path) on a bounded sample, all host
      cores through OpenBLAS: fit at n_s = 12288 and one full Thompson block of 4096 candidates
      (about 10-20 s of CPU work), each
      stage scaled to the full step by its algorithmic work (SURVEY.md section 8d formulas). """
  from oracle import ref_numpy as O
  n_s, b_s = 12288, 4096
  Xs, Ys = X[:n_s], Y[:n_s]
  kern = O.KernelSpec('se', DIM, float(Y.var()), bw)
  t = {}
  t0 = time.perf_counter()
  K = kern(Xs, Xs)
  t['kernel'] = time.perf_counter() - t0
  t0 = time.perf_counter()
  L = O.stable_cholesky(K + noise * np.eye(n_s))
  t['chol'] = time.perf_counter() - t0
  t0 = time.perf_counter()
  yc = Ys - mean_c
  alpha = O.solve_upper_triangular(L.T, O.solve_lower_triangular(L, yc))
  _ = -0.5 * yc.dot(alpha) - np.log(np.diag(L)).sum()
  t['solve'] = time.perf_counter() - t0
  Xc = np.random.RandomState(1).random_sample((b_s, DIM))
  t0 = time.perf_counter()
  K_tetr = kern(Xc, Xs)
  mu = mean_c + K_tetr.dot(alpha)
  K_tete = kern(Xc, Xc)
  t['cross'] = time.perf_counter() - t0
  t0 = time.perf_counter()
  V = O.solve_lower_triangular(L, K_tetr.T)
  t['trsm'] = time.perf_counter() - t0
  t0 = time.perf_counter()
  cov = K_tete - V.T.dot(V)
  t['syrk'] = time.perf_counter() - t0
  t0 = time.perf_counter()
  Lc = O.stable_cholesky(cov)
  s = Lc.dot(np.random.RandomState(2).standard_normal((b_s, 1))).T + mu
  _ = s.argmax()
  t['blockchol'] = time.perf_counter() - t0
  r_n = N_TRAIN / float(n_s)
  r_b = TS_BLOCK / float(b_s)
  fit_full = t['kernel'] * r_n ** 2 + t['chol'] * r_n ** 3 + t['solve'] * r_n ** 2
  block_full = (t['cross'] * r_n * r_b + t['trsm'] * r_n ** 2 * r_b + t['syrk'] * r_n * r_b ** 2 +
                t['blockchol'] * r_b ** 3)
  n_blocks = CANDS_PER_GPU // TS_BLOCK
  full_ms = (fit_full + n_blocks * block_full) * 1e3
  try:
    import threadpoolctl
    threads = max([p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()] + [1])
  except Exception:    # pylint: disable=broad-except
    threads = os.cpu_count()
  return {
    'value': round(full_ms, 1), 'unit': 'ms', 'cores': int(threads), 'kind': 'port',
    'sample': ('oracle/ref_numpy.py (NumPy %s) fit at n=%d + one TS block of %d candidates, '
               'measured %.1f s; each stage scaled to n=%d, block=%d, %d blocks by its algorithmic '
               'work (n^2 kernel, n^3 chol, n^2 b trsm, n b^2 syrk, b^3 block chol)'
               % (np.__version__, n_s, b_s, sum(t.values()), N_TRAIN, TS_BLOCK, n_blocks)),
    'measured_sample_s': {k: round(v, 3) for k, v in t.items()},
    'host_cpus': os.cpu_count(),
  }


def pmc_traffic():
  """ HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes over
      this same command (profiles/r01_pmc_traffic.json, written by tools/rocpd_pmc_traffic.py); None when
      that file is absent.  Counters cannot be collected inside the timed run itself. """
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
  try:
    with open(path) as f:
      return json.load(f)['hbm_bytes_per_launch']
  except (OSError, KeyError, ValueError):
    return None


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=3)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  dist = None
  torch = None
  force_dist = os.environ.get('DFH_BENCH_FORCE_DIST', '0') == '1'   # exercise the RCCL path on one GPU
  if world > 1 or force_dist:
    # torch is plumbing only: process group (RCCL) for the barrier and the 16-byte all-gather
    import torch                      # pylint: disable=import-outside-toplevel
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel
    torch.cuda.set_device(local_rank)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if force_dist and 'RANK' not in os.environ:
      os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_PORT=os.environ.get('MASTER_PORT', '29533'))
    dist.init_process_group('nccl')
  os.environ['DFH_DEVICE'] = str(local_rank)

  from dragonfly_amd.engine import KernelSpec, get_engine
  from dragonfly_amd import parallel
  eng = get_engine()
  X, Y, bw, mean_c, noise, cands, U = make_problem(rank)
  spec = KernelSpec('se', DIM, float(Y.var()), bw)
  Xd = eng.to_device(X)
  yd = eng.to_device(Y - mean_c)
  cd = eng.to_device(cands)
  ud = eng.to_device(U)

  def sync_all():
    eng.sync()
    if dist is not None:
      torch.cuda.synchronize()
      dist.barrier()

  results = {}

  def step():
    gp = eng.gp_fit(spec, Xd, yd, noise)
    v, i = gp.thompson(cd, ud, block=TS_BLOCK, mean_const=mean_c)
    i += rank * CANDS_PER_GPU
    if dist is not None:
      v, i = parallel.allgather_argmax(v, i, device='cuda:%d' % local_rank)
    results['lml'], results['best'], results['idx'] = gp.lml, v, i
    gp.free()

  for _ in range(args.warmup):
    step()
  sync_all()
  eng.gemm_profile(enable=True, fetch=False)      # event pairs only, no host synchronisation
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  sync_all()
  elapsed = time.perf_counter() - t0
  gstats = eng.gemm_profile(enable=False, fetch=True)
  # section breakdown from one extra, untimed step (section timers synchronise the host)
  eng.timings(True)
  step()
  sync_all()
  sections = eng.timings(False)
  cond_est = None
  if rank == 0:
    # conditioning of the matrix that was factored (quoted next to the parity numbers):
    # lambda_max(K) by power iteration on the host, lambda_min(K + noise I) >= noise
    gp = eng.gp_fit(spec, Xd, yd, noise)
    Kh = gp.get_K()
    gp.free()
    v = np.ones(N_TRAIN) / np.sqrt(N_TRAIN)
    lam = 0.0
    for _ in range(12):
      w = Kh.dot(v)
      lam = float(np.linalg.norm(w))
      v = w / lam
    cond_est = {'lambda_max_K': round(lam, 3), 'noise_var': noise,
                'cond_upper_bound': round((lam + noise) / noise, 1)}
    del Kh
  if dist is not None:
    tt = torch.tensor([elapsed], dtype=torch.float64, device='cuda:%d' % local_rank)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
  ms_per_step = elapsed * 1e3 / args.steps

  out = None
  if rank == 0:
    g0 = gstats[0]                       # 128x128 NT tiles: the throughput configuration
    all_ms = sum(g['ms'] for g in gstats)
    all_flop = sum(g['flop'] for g in gstats)
    # launches on the look-ahead / pipeline streams overlap and then share the CUs: the kernel's
    # wall-clock is the union of its launch intervals (busy_ms), not the sum of their durations
    achieved = g0['flop'] / (g0['busy_ms'] * 1e-3) / 1e12 if g0['busy_ms'] > 0 else 0.0
    achieved_sum = g0['flop'] / (g0['ms'] * 1e-3) / 1e12 if g0['ms'] > 0 else 0.0
    out = {
      'metric': 'GP-fit+acq-batch ms at n=16384,d=32',
      'value': round(ms_per_step, 3), 'unit': 'ms', 'n_gpus': world, 'steps': args.steps,
      'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': False,
      'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
      'config': {'workload': ('C3 fit (n=16384, d=32, SE-ARD: kernel matrix + Cholesky + alpha + lml) '
                              '+ C4 shard: blocked-joint Thompson sampling, block 4096, over 262144 '
                              'candidates per GPU with arg-max'),
                 'n': N_TRAIN, 'd': DIM, 'candidates_per_gpu': CANDS_PER_GPU,
                 'candidates_total': CANDS_PER_GPU * world, 'ts_block': TS_BLOCK,
                 'parallelism': 'candidate shards x%d, replicated fit, all-gather of (val,idx)' % world},
      'candidates_per_s': round(CANDS_PER_GPU * world / (ms_per_step * 1e-3), 1),
      'sections_ms_extra_untimed_step_rank0': {k: round(v, 3) for k, v in sections.items() if v > 0},
      'result': {'lml': results['lml'], 'ts_best': results['best'], 'ts_argmax': int(results['idx'])},
      'conditioning': cond_est,
      'roofline': {
        'bound': 'mfma', 'kernel': 'gemm_f64_kernel<NT,128x128> (v_mfma_f64_16x16x4_f64)',
        'achieved': round(achieved, 2), 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), 'traffic': pmc_traffic(),
        'launches_per_step': g0['launches'] / args.steps,
        'avg_launch_us': round(g0['busy_ms'] * 1e3 / max(1, g0['launches']), 2),
        'avg_launch_us_incl_overlap': round(g0['ms'] * 1e3 / max(1, g0['launches']), 2),
        'achieved_from_sum_of_durations': round(achieved_sum, 2),
        'algorithmic_gflop_per_launch': round(g0['flop'] / max(1, g0['launches']) / 1e9, 3),
        'all_gemm_variants': {'ms_per_step': round(all_ms / args.steps, 3),
                              'tflops': round(all_flop / (all_ms * 1e-3) / 1e12, 2) if all_ms > 0 else 0.0},
      },
      'device': eng.name(),
    }
    # untimed extra: drawing this rank's candidates (oper_utils.py:62 -- np.random.random((m, d))) on
    # the device instead of on the host + PCIe; inputs of the timed step are resident either way
    import time as _time
    gen = {}
    buf = eng.empty((CANDS_PER_GPU, DIM))
    for label, rng in (('mt19937_continuing_numpy_state', np.random.RandomState(11)),
                       ('philox4x64', np.random.Generator(np.random.Philox(key=11)))):
      eng.random_candidates(64, DIM, rng=rng, out=buf)
      eng.sync()
      t0 = _time.perf_counter()
      eng.random_candidates(CANDS_PER_GPU, DIM, rng=rng, out=buf)
      eng.sync()
      gen[label + '_ms'] = round((_time.perf_counter() - t0) * 1e3, 3)
    t0 = _time.perf_counter()
    host_draw = np.random.RandomState(11).random_sample((CANDS_PER_GPU, DIM))
    t1 = _time.perf_counter()
    buf.upload(host_draw)
    eng.sync()
    gen['host_numpy_draw_ms'] = round((t1 - t0) * 1e3, 3)
    gen['host_upload_ms'] = round((_time.perf_counter() - t1) * 1e3, 3)
    buf.free()
    out['candidate_generation_untimed_rank0'] = gen
    if not args.no_cpu_baseline and world == 1:
      out['cpu_baseline'] = cpu_baseline(X, Y, bw, mean_c, noise)
    elif not args.no_cpu_baseline:
      out['cpu_baseline'] = None
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
  if rank == 0:
    sys.stdout.flush()
    print(json.dumps(out), flush=True)      # the ONE JSON line, last thing on stdout


if __name__ == '__main__':
  main()
