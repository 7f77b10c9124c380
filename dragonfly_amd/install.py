"""Install the MI355X engine under a real Dragonfly (the drop-in of SURVEY.md section 8b).

    import dragonfly_amd.install as dfi
    dfi.install()          # before building optimisers / calling maximise_function
    ...
    dfi.uninstall()

What is rebound (each is a *name looked up at call time* in the reference, so no reference file is
edited):
  S1 kernels   dragonfly.gp.kernel.SEKernel / MaternKernel / AdditiveKernel  -> dragonfly_amd.kernel
               (the Euclidean kernel factory resolves `gp_kernel.SEKernel` etc. at call time:
               dragonfly/gp/euclidean_gp.py:17,850,859,896)
  S2/S3 GP     dragonfly.gp.euclidean_gp.EuclideanGP -> dragonfly_amd.euclidean_gp.EuclideanGP
               (every Euclidean fitter constructs its GP through this module global,
               dragonfly/gp/euclidean_gp.py:338)
  S2' MF GP    (only with install(multi_fidelity=True)) dragonfly.gp.euclidean_gp.EuclideanMFGP ->
               dragonfly_amd.mf_gp.EuclideanMFGP.  SE / Matern fidelity and domain kernels give the
               coordinate-product kernel on the device; any other factor (the reference's poly /
               expdecay kernels) puts the GP into host-kernel mode -- kernel evaluated by its own
               class on the host, factorisation and posterior on the device.  The MF fitter
               constructs its GP through this module global (dragonfly/gp/euclidean_gp.py:707).
               Opt-in because the reference class cannot be reached once the name is rebound (its
               __init__ calls super(EuclideanMFGP, self)).
  S2'' CP GP   (only with install(cartesian_product=True)) dragonfly.gp.cartesian_product_gp.CPGP ->
               dragonfly_amd.cartesian_product_gp.device_cpgp_class(...): the reference's own class body over the
               device GP; the kernel (the reference's
               CartesianProductKernel) stays on the host, the 'project_first' eigen-projection of
               the Gram matrix and of every posterior covariance (gp_core.py:838-841, 849-857), the
               factorisation and the posterior run on the device.  The CP fitter constructs its GP
               through this module global.  Opt-in for the same reason as the MF GP.
  S4 acquisitions  dispatchers are written into the namespaces
               dragonfly.opt.gpb_acquisitions.asy / syn / seq (looked up with getattr at
               dragonfly/opt/gp_bandit.py:490,510,651,681): the fused callables on Euclidean
               domains, the reference's own callables on every other domain
  S3 fitter    dragonfly.opt.gp_bandit.EuclideanGPFitter and
               dragonfly.opt.multiobjective_gp_bandit.EuclideanGPFitter (the names the bandits
               construct their fitters by, gp_bandit.py:22,584; multiobjective_gp_bandit.py:27,481)
               -> a subclass OF THE REFERENCE'S fitter that only changes how the maximum-likelihood
               tuners see their objective: as a batch (candidates in, log marginal likelihoods
               out, one dfh_gp_lml_batch call) instead of one fit per callback.  'rand' and
               'rand_exp_sampling' evaluate their whole sample in one call, 'pdoo' -- and 'direct'
               without the Fortran library, the reference's own fall-back and its default -- run
               the tree search of dragonfly_amd.doo a frontier per call, and the slice sampler of
               the posterior-sampling criterion evaluates its loops' next candidates a batch at
               a time (dragonfly_amd.slice_sampler: the same chain, draw for draw).  Everything
               else of the fitter (options, bounds, priors, the bandit bookkeeping) is the
               reference's code.  install(batched_tuning=False) leaves the fitter alone.
  S4' MOO      dragonfly.opt.multiobjective_gpb_acquisitions.maximise_acquisition -> ours for
               Euclidean domains: the multi-objective closures are maximised by the batched tree
               search / the vectorised random search too.
A compiled Fortran DIRECT, when present, keeps working: it calls gp.eval per point, which now
runs on the device, through `external_maximise_with_method`.
"""
import os

import numpy as np

from . import gaplog


_saved = []     # (object, attribute name, original value)


def install(multi_fidelity=False, batched_tuning=True, cartesian_product=False):
  """ Rebinds the names listed above; returns the list of patched attributes. """
  import dragonfly.gp.kernel as ref_kernel
  import dragonfly.gp.euclidean_gp as ref_egp
  import dragonfly.opt.gpb_acquisitions as ref_acq
  from dragonfly.exd.exd_utils import maximise_with_method
  from . import kernel, euclidean_gp, gpb_acquisitions, mf_gp
  patched = []
  def _set(mod, name, new):
    _saved.append((mod, name, getattr(mod, name)))
    setattr(mod, name, new)
    patched.append('%s.%s' % (mod.__name__, name))
  for name in ('SEKernel', 'MaternKernel', 'AdditiveKernel'):
    _set(ref_kernel, name, getattr(kernel, name))
  _set(ref_egp, 'EuclideanGP', euclidean_gp.EuclideanGP)

  if multi_fidelity:
    _set(ref_egp, 'EuclideanMFGP', mf_gp.EuclideanMFGP)
  if cartesian_product:
    import dragonfly.gp.cartesian_product_gp as ref_cpgp
    from . import cartesian_product_gp
    _set(ref_cpgp, 'CPGP', cartesian_product_gp.device_cpgp_class(ref_cpgp))
  for ns_name in ('asy', 'syn', 'seq'):
    ref_ns = getattr(ref_acq, ns_name)
    our_ns = getattr(gpb_acquisitions, ns_name)
    for acq in ('ucb', 'ei', 'pi', 'ttei', 'ts', 'add_ucb'):
      _saved.append((ref_ns, acq, getattr(ref_ns, acq)))
      setattr(ref_ns, acq, _euclidean_dispatch(getattr(our_ns, acq), getattr(ref_ns, acq)))
      patched.append('dragonfly.opt.gpb_acquisitions.%s.%s' % (ns_name, acq))
  gpb_acquisitions.external_maximise_with_method = maximise_with_method
  # the multi-objective acquisitions (opt/multiobjective_gpb_acquisitions.py:19-107) are closures
  # over rows of points handed to maximise_acquisition, a name they import at module level: with
  # ours they get the batched tree search / vectorised random search on Euclidean domains
  import dragonfly.opt.multiobjective_gpb_acquisitions as ref_moo_acq
  ref_maximise = ref_moo_acq.maximise_acquisition
  def _moo_maximise_acquisition(acq_fn, anc_data, *args, **kwargs):
    if anc_data.domain.get_type() == 'euclidean':
      return gpb_acquisitions.maximise_acquisition(acq_fn, anc_data, *args, **kwargs)
    return ref_maximise(acq_fn, anc_data, *args, **kwargs)
  _set(ref_moo_acq, 'maximise_acquisition', _moo_maximise_acquisition)
  if batched_tuning:
    import dragonfly.opt.gp_bandit as ref_gp_bandit
    import dragonfly.opt.multiobjective_gp_bandit as ref_moo_bandit
    batched = make_batched_fitter(ref_egp.EuclideanGPFitter)
    _set(ref_gp_bandit, 'EuclideanGPFitter', batched)
    _set(ref_moo_bandit, 'EuclideanGPFitter', batched)
  return patched


def _euclidean_dispatch(ours, theirs):
  """ The acquisition entry installed in the reference's namespaces: ours on Euclidean domains,
      the reference's own callable on every other domain (Cartesian-product, neural-network, ...
      domains keep running exactly as before).  Works for asy (gp, anc_data) and syn
      (num_workers, gps, anc_datas) signatures: the ancillary data is the last argument. """
  def acquisition(*args, **kwargs):
    anc_data = args[-1]
    if isinstance(anc_data, (list, tuple)):
      anc_data = anc_data[0]
    if anc_data.domain.get_type() == 'euclidean':
      return ours(*args, **kwargs)
    return theirs(*args, **kwargs)
  acquisition.__wrapped__ = ours
  acquisition.reference_callable = theirs
  return acquisition


# DFH_SLICE_MERGE=0: the speculative slice sampler's stepping-out and shrinking candidates in separate density calls again
_SLICE_MERGE_FIRST = os.environ.get('DFH_SLICE_MERGE', '1') != '0'


class _KernelMeanNoise(object):
  """ What BatchedEuclideanGPFitter._lml_batch needs of the GP the reference's build_gp constructs -- kernel, mean
      function, noise variance -- with EuclideanGP's constructor signature (gp/euclidean_gp.py:151-152). """
  __slots__ = ('kernel', 'mean_func', 'noise_var', 'host_kernel')

  def __init__(self, X, Y, kernel, mean_func, noise_var, kernel_hyperparams=None, build_posterior=True, reporter=None):
    # pylint: disable=unused-argument
    if isinstance(kernel, str) or build_posterior:
      raise RuntimeError('dragonfly_amd.install: the candidate stand-in was asked for a real GP')
    self.kernel, self.mean_func, self.noise_var = kernel, mean_func, noise_var
    # (gp_core.GP._generic for a plain EuclideanGP: a kernel without a device description is evaluated by the host)
    self.host_kernel = not (hasattr(kernel, 'to_spec') and getattr(kernel, 'has_device_spec', lambda: True)())


def make_batched_fitter(ref_fitter_cls):
  """ A subclass of the reference's EuclideanGPFitter whose maximum-likelihood tuners evaluate the
      tuning objective (gp_core.py:551-564) in batches on the device. """
  from .doo import pdoo_maximise_batched
  from .engine import get_engine, KernelSpec
  from .gpb_acquisitions import _fortran_direct_available
  from .kernel import _as_2d_array
  from .oper_utils import random_maximise, random_sample_cts_dscr

  class BatchedEuclideanGPFitter(ref_fitter_cls):
    """ dragonfly.gp.euclidean_gp.EuclideanGPFitter with batched ML tuning (dragonfly_amd.install). """
    pdoo_frontier = 32

    def _set_up_ml_hp_tune(self):
      """ gp_core.py:423-474, then the optimisers are swapped for their batch-objective forms
          (same random draws / same boxes, so the same hyper-parameters win). """
      ref_fitter_cls._set_up_ml_hp_tune(self)
      self._X_dev = None
      method = self.ml_hp_tune_opt_method
      def _rand(obj, max_evals):
        val, pt, _ = random_maximise(obj, self.cts_hp_bounds, max_evals, vectorised=True)
        return val, pt, None
      def _tree(obj, max_evals):
        val, pt, _ = pdoo_maximise_batched(obj, self.cts_hp_bounds, max_evals, frontier=self.pdoo_frontier,
                                           depth=2 if self.pdoo_frontier > 0 else 0)
        return val, pt, None
      def _rand_exp(obj, max_evals):
        cts, dscr, lml_vals = random_sample_cts_dscr(obj, self.cts_hp_bounds, self.dscr_hp_vals, max_evals,
                                                     vectorised=True)
        probs = np.exp(lml_vals - max(lml_vals))
        return cts, dscr, probs / probs.sum()
      self._batched_ml = True
      if method == 'rand':
        self.cts_hp_optimise = _rand
      elif method == 'pdoo' or (method == 'direct' and not _fortran_direct_available()):
        self.cts_hp_optimise = _tree
      elif method == 'rand_exp_sampling':
        self.hp_sampler = _rand_exp
      else:
        self._batched_ml = False          # a compiled DIRECT drives the search one point at a time

    def _decode_candidates(self, cts_hps_list, dscr_hps, per_cand, other_gp_params):
      """ (specs, means, noises) of the candidates WITHOUT a call of the reference's build_gp per candidate, for the
          usual cases -- an SE or Matern kernel over all dimensions or an additive model of such kernels, no user mean
          function, the fitter's own build_gp / _child_build_gp -- or None (then _lml_batch takes the general route below).
          The arithmetic is the reference's, operation for operation and on operands of the same shape (gp_core.py:
          501-543: the tuned mean is the first entry, exp of the next is the noise variance; gp/euclidean_gp.py:796-866:
          exp of the next is the scale, np.exp of the next `dim` entries -- one call on the slice -- the bandwidths, or
          exp of one entry repeated; a tuned Matern nu is the first discrete entry), so the numbers are the same bits.
          What it skips are the objects: per candidate a kernel, a mean closure, a stand-in GP and a second
          description of the kernel, ~50 us of Python that a real run pays some 300 000 times per 60 evaluations.
          The first calls of every fitter check candidate 0 against the general route and switch this one off for
          good on any difference. """
      state = getattr(self, '_amd_decode', None)
      if state is None:
        state = self._amd_decode = {'ok': self._decode_applies(), 'checks': 0}
      if not state['ok']:
        return None
      try:
        dim = self.dim
        # what does not depend on the candidate is worked out once per fitter (and set of labels): the option fields,
        # the kernel's fixed hyper-parameters, the constant means gp_core.py:516-523 computes from the labels
        plan = state.get('plan')
        if plan is None or plan[0] is not self.Y or plan[1] != len(self.Y):
          o = self.options
          kh = self._prep_init_kernel_hyperparams(self.kernel_type)
          if kh['dim'] != dim:
            return None
          mean_type, noise_type = o.mean_func_type, o.noise_var_type
          if mean_type == 'mean':
            mean_c = np.mean(self.Y)
          elif mean_type == 'median':
            mean_c = np.median(self.Y)
          elif mean_type == 'upper_bound':
            mean_c = np.mean(self.Y) + 3 * np.std(self.Y)
          elif mean_type == 'const':
            mean_c = o.mean_func_const
          else:
            mean_c = 0
          noise_c = None
          if noise_type == 'label':
            noise_c = o.noise_var_label * (self.Y.std() ** 2)
          elif noise_type != 'tune':
            noise_c = o.noise_var_value
          matern = self.kernel_type == 'matern'
          nu_fixed = kh['nu'] if matern and 'nu' in kh and not kh['nu'] < 0 else None
          plan = state['plan'] = (self.Y, len(self.Y), mean_type, noise_type, mean_c, noise_c, matern, nu_fixed,
                                  o.use_same_bandwidth, bool(o.use_additive_gp))
        _, _, mean_type, noise_type, mean_c, noise_c, matern, nu_fixed, same_bw, additive = plan
        if additive:
          # an additive model (gp/euclidean_gp.py:329-337, 820-826, 893-897): the groups come with the call, every
          # group's kernel has scale 1 and its columns' bandwidths, the sum carries the scale
          groupings = other_gp_params.add_gp_groupings
          groups = [[int(c) for c in grp] for grp in groupings]
          n_groups = len(groups)
        specs, means, noises = [], [], []
        for i, cts in enumerate(cts_hps_list):
          dscr = list(dscr_hps[i]) if per_cand else list(dscr_hps)
          if self.num_hps != len(cts) + len(dscr):
            return None                                 # (the general route raises the reference's error)
          if additive:
            dscr = dscr[:-1]                            # the group size's index
          if mean_type == 'tune':
            mean = cts[0].item()
            cts = cts[1:]
          else:
            mean = mean_c
          if noise_type == 'tune':
            noise = np.exp(cts[0])
            cts = cts[1:]
          else:
            noise = noise_c
          scale = np.exp(cts[0])
          cts = cts[1:]
          if same_bw:
            bws = [np.exp(cts[0])] * dim
            cts = cts[1:]
          else:
            bws = np.exp(cts[0:dim])
            cts = cts[dim:]
          if len(cts) != 0:
            return None
          if matern:
            if nu_fixed is None:
              nu = dscr[0]
              dscr = dscr[1:]
            else:
              nu = nu_fixed
            if nu % 1 != 0.5:
              return None
          if len(dscr) != 0 or len(bws) != dim:
            return None
          if additive:
            specs.append(KernelSpec('additive', dim, scale, groups=groups, sub_kinds=['matern' if matern else 'se'] * n_groups,
                                    sub_scales=[1.0] * n_groups, sub_nus=[nu if matern else 0.0] * n_groups,
                                    sub_bandwidths=[np.ravel(np.asarray([bws[idx] for idx in grp], dtype=float))
                                                    for grp in groupings]))
          else:
            specs.append(KernelSpec('matern', dim, scale, bws, nu=nu) if matern else KernelSpec('se', dim, scale, bws))
          means.append(float(mean))
          noises.append(float(noise))
      except Exception:             # pylint: disable=broad-except
        return None                 # whatever it was, the general route meets it the way the reference does
      if state['checks'] < 3:
        state['checks'] += 1
        want = self._build_one(cts_hps_list[0], list(dscr_hps[0]) if per_cand else list(dscr_hps), other_gp_params)
        got = (specs[0].signature(), means[0], noises[0])
        if want is None or want != got:
          state['ok'] = False
          return None
      return specs, means, noises

    def _decode_applies(self):
      import dragonfly.gp.kernel as _ref_kernel
      from . import kernel as _kernel
      o = self.options
      return (type(self)._child_build_gp is ref_fitter_cls._child_build_gp and type(self).build_gp is ref_fitter_cls.build_gp
              and self.kernel_type in ('se', 'matern') and getattr(o, 'mean_func', None) is None
              and _ref_kernel.SEKernel is _kernel.SEKernel and _ref_kernel.MaternKernel is _kernel.MaternKernel
              and _ref_kernel.AdditiveKernel is _kernel.AdditiveKernel)

    def _build_one(self, cts, dscr, other_gp_params=None):
      """ (kernel signature, mean, noise) of one candidate by the general route: the reference's build_gp. """
      import dragonfly.gp.euclidean_gp as _ref_egp
      saved_cls = _ref_egp.EuclideanGP
      _ref_egp.EuclideanGP = _KernelMeanNoise
      try:
        gp = self.build_gp(cts, dscr, other_gp_params=other_gp_params, build_posterior=False)
      finally:
        _ref_egp.EuclideanGP = saved_cls
      if getattr(gp, 'host_kernel', getattr(gp, '_generic', True)):
        return None
      return (gp.kernel.to_spec(self.dim).signature(), float(gp.mean_func([np.zeros(self.dim)])[0]), float(gp.noise_var))

    def _lml_batch(self, cts_hps_list, dscr_hps, other_gp_params=None):
      """ Log marginal likelihoods of a list of candidates.  Each candidate goes through the
          reference's own build_gp (gp_core.py:501-543) without building a posterior -- that gives
          the kernel, the constant mean and the noise variance it would be fitted with -- and the
          whole list is fitted in one device call. """
      per_cand = len(dscr_hps) > 0 and isinstance(dscr_hps[0], (list, tuple, np.ndarray))
      if len(cts_hps_list) == 0:
        return np.zeros((0,))
      user_mean = getattr(self.options, 'mean_func', None) is not None
      decoded = self._decode_candidates(cts_hps_list, dscr_hps, per_cand, other_gp_params)
      if decoded is not None:
        specs, means, noises = decoded
        if getattr(self, '_X_dev', None) is None:
          self._X_dev = get_engine().to_device(_as_2d_array(self.X))
        lmls = get_engine().gp_lml_batch(specs, self._X_dev, self._labels_array(), means, noises)
        if gaplog.ENABLED and len(lmls) >= 16:       # (a random-search batch: its arg-max is the fitter's choice)
          gaplog.top2('hp_batch', lmls)
        return lmls
      specs, means, noises = [], [], []
      probe = [np.zeros(self.dim)]
      # build_gp ends in `EuclideanGP(self.X, self.Y, kernel, mean_func, noise_var, build_posterior=False)`, a module
      # global of the reference looked up at call time (gp/euclidean_gp.py:338).  All that is wanted of that object here
      # are its three arguments: for the length of this loop the name is bound to a stand-in that keeps them and does
      # nothing else (a GP object per candidate -- data lists copied and checked -- was a fifth of a small run's time).
      import dragonfly.gp.euclidean_gp as _ref_egp
      saved_cls = _ref_egp.EuclideanGP
      _ref_egp.EuclideanGP = _KernelMeanNoise
      try:
        for i, cts in enumerate(cts_hps_list):
          dscr = list(dscr_hps[i]) if per_cand else list(dscr_hps)
          gp = self.build_gp(cts, dscr, other_gp_params=other_gp_params, build_posterior=False)
          # (a fitter subclass that builds its GP through its own import gets a real GP here, not the stand-in: it has
          #  no `host_kernel`; what the mirror's GP calls `_generic` says the same, and anything else takes the safe route)
          if user_mean or getattr(gp, 'host_kernel', getattr(gp, '_generic', True)):
            specs = None
            break
          specs.append(gp.kernel.to_spec(self.dim))
          means.append(float(gp.mean_func(probe)[0]))
          noises.append(float(gp.noise_var))
      finally:
        _ref_egp.EuclideanGP = saved_cls
      if specs is None:
        # an arbitrary mean function, or a kernel the host evaluates: one fit per candidate
        return np.array([self._tuning_objective(c, list(dscr_hps[j]) if per_cand else list(dscr_hps),
                                                other_gp_params=other_gp_params)
                         for j, c in enumerate(cts_hps_list)])
      if getattr(self, '_X_dev', None) is None:
        self._X_dev = get_engine().to_device(_as_2d_array(self.X))
      return get_engine().gp_lml_batch(specs, self._X_dev, self._labels_array(), means, noises)

    def _labels_array(self):
      """ self.Y as the float64 array every batch call hands to the engine (converted once per list of labels) """
      cached = getattr(self, '_amd_labels', None)
      if cached is None or cached[0] is not self.Y or cached[1] != len(self.Y):
        cached = self._amd_labels = (self.Y, len(self.Y), np.asarray(self.Y, dtype=np.float64))
      return cached[2]

    # -- posterior sampling (gp_core.py:476-487, 592-726) -------------------------------------------
    def _set_up_post_sampling_hp_tune(self):
      """ The continuous hyper-parameters are slice-sampled one coordinate at a time
          (gp_core.py:687-697); the sampler is swapped for the speculative one, which follows the
          same chain with its density evaluations batched. """
      ref_fitter_cls._set_up_post_sampling_hp_tune(self)
      if self.options.post_hp_tune_method == 'slice':
        self.hp_sampler_cts = self._speculative_slice

    def _speculative_slice(self, model, init_sample, num_samples, burn):
      # pylint: disable=unused-argument
      from .slice_sampler import SpeculativeSlice
      return SpeculativeSlice(self._post_logp_batch, merge_first=_SLICE_MERGE_FIRST).sample(init_sample, num_samples, burn)

    def _post_logp_batch(self, xs):
      """ The log density `_logp` of gp_core.py:597-622 -- log priors of all hyper-parameters, summed
          in index order, plus the log marginal likelihood -- at each value in xs of the
          coordinate being sampled (self.curr_hp), the others as they currently are. """
      num_cts = len(self.cts_hp_bounds)
      out = np.empty(len(xs))
      pending = []
      base = np.array(self.hps, dtype=np.float64)

      def prior_term(i, value):
        prior = self.hp_priors[i]
        if type(prior).__name__ == 'Categorical':
          return prior.logp(prior.get_id(value))
        return prior.logp(value)
      # only the coordinate being sampled changes between the xs: the other priors' terms are evaluated once, and
      # added in index order per x exactly as the reference's loop adds them (same partial sums, same bits)
      # (and the sampler asks several times per step of a coordinate -- stepping out, shrinking -- with the others unchanged)
      cur = self.curr_hp
      key = (cur, base.tobytes())
      cached = getattr(self, '_amd_prior_terms', None)
      if cached is not None and cached[0] == key and cached[1] is self.hp_priors:
        terms = cached[2]
      else:
        terms = [None if i == cur else prior_term(i, base[i]) for i in range(len(self.hp_priors))]
        self._amd_prior_terms = (key, self.hp_priors, terms)
      for k, x in enumerate(xs):
        hps = base.copy()
        hps[cur] = x
        lp = 0
        for i, t in enumerate(terms):
          lp += prior_term(i, hps[i]) if t is None else t
        if not np.isfinite(lp):
          out[k] = lp
        else:
          pending.append((k, lp, hps))
      if pending:
        lmls = self._lml_batch([h[:num_cts] for _, _, h in pending],
                               [h[num_cts:self.num_hps] for _, _, h in pending], self.other_gp_params)
        for (k, lp, _), lml in zip(pending, lmls):
          out[k] = lp + lml
      return out

    def _optimise_cts_hps_for_given_dscr_hps(self, given_dscr_hps):
      """ gp_core.py:576-583 / euclidean_gp.py:303-313 with the batch objective. """
      if not getattr(self, '_batched_ml', False):
        return ref_fitter_cls._optimise_cts_hps_for_given_dscr_hps(self, given_dscr_hps)
      if self.options.use_additive_gp:
        from dragonfly.gp.euclidean_gp import optimise_cts_hps_for_given_dscr_hps_in_add_model
        return optimise_cts_hps_for_given_dscr_hps_in_add_model(
            given_dscr_hps, self.options.num_groups_per_group_size, self.dim, self.hp_tune_max_evals,
            self.cts_hp_optimise, self._lml_batch)
      objective = lambda arg: self._lml_batch(arg, list(given_dscr_hps))
      val, cts_hps, _ = self.cts_hp_optimise(objective, self.hp_tune_max_evals)
      return val, cts_hps, None

    def _sample_cts_dscr_hps_for_rand_exp_sampling(self):
      """ gp_core.py:585-590 (non-additive models) with the batch objective. """
      if not getattr(self, '_batched_ml', False) or self.options.use_additive_gp:
        return ref_fitter_cls._sample_cts_dscr_hps_for_rand_exp_sampling(self)
      cts, dscr, probs = self.hp_sampler(self._lml_batch, self.hp_tune_max_evals)
      return cts, dscr, [None] * len(cts), probs

  return BatchedEuclideanGPFitter


def uninstall():
  """ Restores every name install() rebound. """
  from . import gpb_acquisitions
  for obj, name, old in reversed(_saved):
    setattr(obj, name, old)
  del _saved[:]
  gpb_acquisitions.external_maximise_with_method = None
