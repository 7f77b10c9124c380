"""MI355X: the tuning objective with one workgroup per candidate (csrc/chol.hip: lml_wg_kernel, 128 < n <= 2047;
GPFitter._tuning_objective, dragonfly/gp/gp_core.py:551-574 -> build_posterior :155-163 -> :222-227): sizes around
the 64-row tile edges (the augmented row n falls into a tile of its own when n is a multiple of 64), more
candidates than one launch holds, candidates that need the stable_cholesky ladder inside a large group, kernels
that are not structurally uniform, and the no-jitter failure."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref_numpy as O

HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _data(n, d, seed):
  rs = np.random.RandomState(seed)
  X = rs.rand(n, d)
  Y = np.sin(4 * X.sum(axis=1)) + 0.1 * rs.randn(n)
  return rs, X, Y


def _specs(rs, d, nb, y_var):
  from dragonfly_amd.engine import KernelSpec
  specs, ospecs, means, noises = [], [], [], []
  for c in range(nb):
    scale = float(np.exp(rs.uniform(np.log(0.2 * y_var), np.log(5 * y_var))))
    bw = np.exp(rs.uniform(np.log(0.1), np.log(3.0), size=d))
    if c % 4 == 3:
      nu = [1.5, 2.5][(c // 4) % 2]
      specs.append(KernelSpec('matern', d, scale, bw, nu=nu))
      ospecs.append(O.KernelSpec('matern', d, scale, bw, nu=nu))
    else:
      specs.append(KernelSpec('se', d, scale, bw))
      ospecs.append(O.KernelSpec('se', d, scale, bw))
    means.append(float(0.3 * rs.randn()))
    noises.append(float(np.exp(rs.uniform(np.log(0.003 * y_var), np.log(0.2 * y_var)))))
  return specs, ospecs, means, noises


@pytest.mark.parametrize('n', [129, 191, 192, 193, 255, 256, 257, 448, 639, 640, 641, 1000, 1024, 1471, 2047])
def test_tile_edges_match_oracle_and_single_fits(engine, n):
  d, nb = 4, 5
  rs, X, Y = _data(n, d, 3 * n + 1)
  specs, ospecs, means, noises = _specs(rs, d, nb, float(Y.var()))
  lml, powers = engine.gp_lml_batch(specs, X, Y, means, noises, return_powers=True)
  for c in (range(nb) if n <= 700 else (0, 3)):
    ref = O.GPOracle(X, Y, ospecs[c], means[c], noises[c]).lml()
    assert abs(lml[c] - ref) <= TOL * abs(ref), (n, c, lml[c], ref)
    one = engine.gp_fit(specs[c], X, Y - means[c], noises[c])
    assert abs(lml[c] - one.lml) <= 1e-12 * abs(one.lml) and powers[c] is None and one.jitter_power is None
    one.free()


def test_more_candidates_than_one_launch_with_ladder_cases_inside(engine):
  """ 700 candidates at n = 150 (three launches of one candidate per CU); every 97th has no noise on duplicated
      points and must come back through the ladder with the reference's jitter power, the others untouched """
  from dragonfly_amd.engine import KernelSpec
  n, d, nb = 150, 3, 700
  rs, X, Y = _data(n, d, 11)
  X[75:] = X[:75]
  specs, ospecs, means, noises = _specs(rs, d, nb, float(Y.var()))
  hard = list(range(5, nb, 97))
  for c in hard:
    noises[c] = 0.0 if c % 2 else 1e-19
    specs[c] = KernelSpec('se', d, 1.0, np.full(d, 2.0)); ospecs[c] = O.KernelSpec('se', d, 1.0, np.full(d, 2.0))
  lml, powers = engine.gp_lml_batch(specs, X, Y, means, noises, return_powers=True)
  for c in sorted(set(list(range(0, nb, 53)) + [255, 256, 257, 511, 512, nb - 1])):
    if c in hard:
      continue
    ref = O.GPOracle(X, Y, ospecs[c], means[c], noises[c]).lml()
    assert powers[c] is None and abs(lml[c] - ref) <= TOL * abs(ref), (c, lml[c], ref)
  for c in hard:
    og = O.GPOracle(X, Y, ospecs[c], means[c], noises[c])
    assert og.jitter_power is not None and powers[c] == og.jitter_power
    one = engine.gp_fit(specs[c], X, Y - means[c], noises[c])       # the same ladder, one candidate at a time
    assert abs(lml[c] - one.lml) <= 1e-12 * abs(one.lml) and one.jitter_power == powers[c]
    one.free()


def test_non_uniform_kernels_and_device_inputs(engine):
  """ additive and product kernels (one Gram launch per candidate) mixed with plain ones; X and y resident on the device """
  from dragonfly_amd.engine import KernelSpec
  n, d = 333, 6
  rs, X, Y = _data(n, d, 21)
  yv = float(Y.var())
  bw = lambda k: np.exp(rs.uniform(np.log(0.2), np.log(2.0), size=k))
  pairs = []
  for rep in range(3):
    perm = list(rs.permutation(d))
    groups = [perm[:2], perm[2:4], perm[4:]]
    for multi in ('additive', 'product'):
      bws = [bw(2) for _ in groups]
      pairs.append((KernelSpec(multi, d, 0.8 * yv, groups=groups, sub_kinds=['se', 'matern', 'se'], sub_scales=[1.0] * 3,
                               sub_nus=[0.0, 2.5, 0.0], sub_bandwidths=bws),
                    O.KernelSpec(multi, d, 0.8 * yv, groups=groups,
                                 subs=[O.KernelSpec('se', 2, 1.0, bws[0]), O.KernelSpec('matern', 2, 1.0, bws[1], nu=2.5),
                                       O.KernelSpec('se', 2, 1.0, bws[2])])))
    b = bw(d)
    pairs.append((KernelSpec('se', d, yv, b), O.KernelSpec('se', d, yv, b)))
  means = [float(0.2 * rs.randn()) for _ in pairs]
  noises = [float(yv * np.exp(rs.uniform(np.log(0.01), np.log(0.2)))) for _ in pairs]
  Xd, yd = engine.to_device(X), engine.to_device(Y)
  lml = engine.gp_lml_batch([p[0] for p in pairs], Xd, yd, means, noises)
  for c, (_, ospec) in enumerate(pairs):
    ref = O.GPOracle(X, Y, ospec, means[c], noises[c]).lml()
    assert abs(lml[c] - ref) <= TOL * abs(ref), (c, ospec.kind, lml[c], ref)
  Xd.free(); yd.free()


def test_failures_surface_like_the_single_fit(engine):
  from dragonfly_amd.engine import KernelSpec
  n, d = 200, 2
  rs, X, Y = _data(n, d, 31)
  X[100:] = X[:100]
  specs = [KernelSpec('se', d, 1.0, np.full(d, b)) for b in (0.3, 2.0, 0.5)]
  noises = [1e-3, 0.0, 1e-2]
  with pytest.raises(np.linalg.LinAlgError):
    engine.gp_lml_batch(specs, X, Y, None, noises, allow_jitter=False)
  Xnan = X.copy()
  Xnan[3, 1] = np.nan                               # NaN in the Gram matrix: the ladder cannot help
  with pytest.raises(ValueError):
    engine.gp_lml_batch(specs[:1], Xnan, Y, None, noises[:1])
  lml = engine.gp_lml_batch(specs, X, Y, None, [1e-3, 1e-3, 1e-2])      # (the context is fine afterwards)
  assert np.all(np.isfinite(lml))


# the schedule switches are read once per process: each variant runs tests/lml_wg_check.py in a subprocess
VARIANTS = {
  'defaults': ({}, 0),
  'one-workgroup-per-candidate-only': ({'DFH_LML_TEAM': '0'}, 0),
  'teams-of-two': ({'DFH_LML_TEAM': '2'}, 0),
  'teams-of-four': ({'DFH_LML_TEAM': '4'}, 0),
  'teams-of-sixteen': ({'DFH_LML_TEAM': '16'}, 0),
  'small-groups': ({'DFH_LML_WG_GROUP': '7'}, 0),
  'small-problems-too': ({'DFH_LML_TINY': '0'}, 0),
  'small-problems-too-no-teams': ({'DFH_LML_TINY': '0', 'DFH_LML_TEAM': '0'}, 0),
  # every hand-off between the members of a team expires at once: the status word sends the group back through
  # one workgroup per candidate (counted in dfh_ctx_counters)
  'forced-handoff-timeout': ({'DFH_TEST_SPIN_LIMIT': '0'}, 1),
  'lock-step-schedule': ({'DFH_LML_WG': '0'}, 0),
  # (round 6: groups of at most 16 candidates at 64 <= n <= 128 take lml_wgf_kernel by default -- tests/test_gpu_lml_fused.py;
  #  without it those sizes are back on the teams / one workgroup per candidate)
  'no-one-launch-small-groups': ({'DFH_LML_FUSED': '0'}, 0),
  # (the one-launch form stops at n = 128 by default -- beyond, the team schedule is as fast; the kernel itself goes to 255)
  'one-launch-small-groups-up-to-255': ({'DFH_LML_FUSED_MAX_N': '255'}, 0),
  # (round 6, n <= 128: descriptors / results through the mapped pinned buffer or by copies; n <= 63 on the 64 x 64
  #  factorisation or on k_lml_tiny's column loop)
  'small-problems-by-copies': ({'DFH_LML_DIRECT': '0'}, 0),
  'small-problems-old-kernel': ({'DFH_LML_TINY64': '0'}, 0),
  'small-problems-old-kernel-by-copies': ({'DFH_LML_TINY64': '0', 'DFH_LML_DIRECT': '0', 'DFH_LML_FUSED': '0'}, 0),
  'substitutions-general-route': ({'DFH_TRSV_FAST': '0'}, 0),
  'no-one-launch-small-groups-no-teams': ({'DFH_LML_FUSED': '0', 'DFH_LML_TEAM': '0'}, 0),
}


@pytest.mark.parametrize('name', sorted(VARIANTS))
def test_schedule_variant(engine, name):
  env = dict(os.environ)
  env.update(VARIANTS[name][0])
  res = subprocess.run([sys.executable, os.path.join(HERE, 'lml_wg_check.py')], env=env, capture_output=True, text=True,
                       timeout=600)
  assert res.returncode == 0 and res.stdout.strip().endswith('OK'), (res.stdout[-2000:], res.stderr[-4000:])
  fallbacks = int(re.search(r'fallbacks (\d+)', res.stdout).group(1))
  assert (fallbacks > 0) == bool(VARIANTS[name][1]), res.stdout
