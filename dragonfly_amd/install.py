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
  S4 acquisitions  the fused callables are written into the namespaces
               dragonfly.opt.gpb_acquisitions.asy / syn / seq (looked up with getattr at
               dragonfly/opt/gp_bandit.py:490,510,651,681)
Dragonfly's own serial maximisers (DIRECT / PDOO) keep working: they call gp.eval per point, which
now runs on the device, through `external_maximise_with_method`.
"""
_saved = []     # (object, attribute name, original value)


def install(multi_fidelity=False):
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
  for ns_name in ('asy', 'syn', 'seq'):
    ref_ns = getattr(ref_acq, ns_name)
    our_ns = getattr(gpb_acquisitions, ns_name)
    for acq in ('ucb', 'ei', 'pi', 'ttei', 'ts', 'add_ucb'):
      _saved.append((ref_ns, acq, getattr(ref_ns, acq)))
      setattr(ref_ns, acq, getattr(our_ns, acq))
      patched.append('dragonfly.opt.gpb_acquisitions.%s.%s' % (ns_name, acq))
  gpb_acquisitions.external_maximise_with_method = maximise_with_method
  return patched


def uninstall():
  """ Restores every name install() rebound. """
  from . import gpb_acquisitions
  for obj, name, old in reversed(_saved):
    setattr(obj, name, old)
  del _saved[:]
  gpb_acquisitions.external_maximise_with_method = None
