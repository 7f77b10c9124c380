"""MI355X: a handful of mid-sized candidates in ONE launch (csrc/chol.hip: lml_wgf_kernel, 64 <= n <= 128 by default, at most 16
candidates per call -- the slice sampler's and the tree search's calls of GPFitter._tuning_objective,
dragonfly/gp/gp_core.py:551-574 -> build_posterior :155-163 -> :222-227): the workgroup builds its candidate's Gram
matrix itself (get_scaled_repr kernel.py:179-181, dist_squared general_utils.py:58-70, SE / Matern / additive /
product), factors the augmented matrix and publishes sum log L_ii and z.z straight into pinned host memory.  Sizes around
the 64-row tile edges, every group size, the kernels' families, candidates that need the stable_cholesky ladder or have
no noise at all, the failures, and inputs too wide for the kernel's LDS (which take the other schedules)."""
import numpy as np
import pytest

from oracle import ref_numpy as O

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


# (n > 128: past the kernel's default range -- the same call takes the team schedule; the kernel itself up to n = 255 is
#  forced in tests/test_gpu_lml_wg.py::test_schedule_variant[one-launch-small-groups-up-to-255])
@pytest.mark.parametrize('n', [64, 65, 100, 126, 127, 128, 129, 150, 190, 191, 192, 255])
@pytest.mark.parametrize('nb', [1, 3, 16])
def test_tile_edges_and_group_sizes(engine, n, nb):
  d = 4
  rs, X, Y = _data(n, d, 5 * n + nb)
  specs, ospecs, means, noises = _specs(rs, d, nb, float(Y.var()))
  Xd = engine.to_device(X)
  lml, powers = engine.gp_lml_batch(specs, Xd, Y, means, noises, return_powers=True)
  for c in sorted(set([0, nb // 2, nb - 1])):
    ref = O.GPOracle(X, Y, ospecs[c], means[c], noises[c]).lml()
    assert abs(lml[c] - ref) <= TOL * abs(ref), (n, nb, c, lml[c], ref)
    one = engine.gp_fit(specs[c], X, Y - means[c], noises[c])
    assert abs(lml[c] - one.lml) <= 1e-12 * abs(one.lml) and powers[c] is None and one.jitter_power is None
    one.free()
  # the same candidates in a group too large for the one-launch form: the other schedules agree to rounding
  big = engine.gp_lml_batch(specs * 6, Xd, Y, means * 6, noises * 6)
  assert np.max(np.abs(big[:nb] - lml) / np.abs(lml)) <= 1e-12
  Xd.free()


def test_additive_and_product_kernels(engine):
  from dragonfly_amd.engine import KernelSpec
  n, d = 150, 6
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


def test_ladder_and_no_noise_candidates_inside_a_small_group(engine):
  from dragonfly_amd.engine import KernelSpec
  n, d = 150, 3
  rs, X, Y = _data(n, d, 11)
  X[75:] = X[:75]                                   # duplicated points: singular without noise
  specs, ospecs, means, noises = _specs(rs, d, 8, float(Y.var()))
  for c, noise in ((2, 0.0), (5, 1e-19)):
    noises[c] = noise
    specs[c] = KernelSpec('se', d, 1.0, np.full(d, 2.0)); ospecs[c] = O.KernelSpec('se', d, 1.0, np.full(d, 2.0))
  lml, powers = engine.gp_lml_batch(specs, X, Y, means, noises, return_powers=True)
  for c in range(8):
    og = O.GPOracle(X, Y, ospecs[c], means[c], noises[c])
    if c in (2, 5):
      assert og.jitter_power is not None and powers[c] == og.jitter_power
      one = engine.gp_fit(specs[c], X, Y - means[c], noises[c])       # the same ladder, one candidate at a time
      assert abs(lml[c] - one.lml) <= 1e-12 * abs(one.lml) and one.jitter_power == powers[c]
      one.free()
    else:
      assert powers[c] is None and abs(lml[c] - og.lml()) <= TOL * abs(og.lml()), (c, lml[c], og.lml())


def test_failures_surface_like_the_single_fit(engine):
  from dragonfly_amd.engine import KernelSpec
  n, d = 180, 2
  rs, X, Y = _data(n, d, 31)
  X[90:] = X[:90]
  specs = [KernelSpec('se', d, 1.0, np.full(d, b)) for b in (0.3, 2.0, 0.5)]
  with pytest.raises(np.linalg.LinAlgError):
    engine.gp_lml_batch(specs, X, Y, None, [1e-3, 0.0, 1e-2], allow_jitter=False)
  Xnan = X.copy()
  Xnan[3, 1] = np.nan                               # NaN in the Gram matrix: the ladder cannot help
  with pytest.raises(ValueError):
    engine.gp_lml_batch(specs[:1], Xnan, Y, None, [1e-3])
  lml = engine.gp_lml_batch(specs, X, Y, None, [1e-3, 1e-3, 1e-2])      # (the context is fine afterwards)
  assert np.all(np.isfinite(lml))


def test_inputs_too_wide_for_the_kernels_lds_take_the_other_schedules(engine):
  n, d, nb = 180, 60, 3                             # 180 x (60 + 1) doubles of scaled inputs and norms do not fit
  rs, X, Y = _data(n, d, 41)
  specs, ospecs, means, noises = _specs(rs, d, nb, float(Y.var()))
  lml = engine.gp_lml_batch(specs, X, Y, means, noises)
  for c in range(nb):
    ref = O.GPOracle(X, Y, ospecs[c], means[c], noises[c]).lml()
    assert abs(lml[c] - ref) <= TOL * abs(ref)
