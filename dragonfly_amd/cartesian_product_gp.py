"""GPs on Cartesian-product domains on the MI355X: the reference's own CPGP
(dragonfly/gp/cartesian_product_gp.py:208-248) re-based onto the device GP at install() time.

A CPGP's kernel is a product over the parts of the domain (Euclidean, integral, discrete, neural-
network, ...), not guaranteed to be positive semi-definite, so the reference builds its posterior
through _get_cholesky_decomp's 'project_first' branch (gp_core.py:838-841: eigen-projection of the
Gram matrix onto the PSD cone) and projects every posterior covariance as well (gp_core.py:849-857).
Here the kernel object stays whatever the caller built (the reference's CartesianProductKernel and
its parts are evaluated on the host, also from pre-computed distance lists), while the projection
(csrc/psdproj.hip), the factorisation, the solves and the posterior run on the device: the GP of
gp_core.py in host-kernel mode with handle_non_psd_kernels='project_first'.

Nothing of the reference class is restated here.  device_cpgp_class(ref_module) makes a class whose
base is dragonfly_amd.gp_core.GP and whose body IS the reference class's body -- its constructor, its
distance-list setter, its string form and its training-kernel-matrix hook, the function objects
themselves.  Overriding that documented hook is what puts the device GP into host-kernel mode
(dfh_gp_fit_gram with DFH_FIT_PROJECT_FIRST), so build_posterior / eval / the hallucinated posterior
are the device's and everything else is the reference's.

dragonfly_amd.install(cartesian_product=True) rebinds dragonfly.gp.cartesian_product_gp.CPGP to that
class: the reference's CPGPFitter constructs its GPs through the module global
(cartesian_product_gp.py: `CPGP(X, Y, kernel, mean_func, noise_var, ...)`), so Cartesian-product
runs -- the reference's default for every non-Euclidean domain -- get the device posterior.
"""
import types

from .gp_core import GP


def device_cpgp_class(ref_module):
  """ ref_module: dragonfly.gp.cartesian_product_gp (as imported by the caller).  Returns the class described above.
      The reference's constructor names its own class in `super(CPGP, self)`, a module global: the functions are
      re-made over a copy of the module's globals in which that name is the new class, so the class works whether or
      not the module global has been rebound. """
  ref_cls = ref_module.__dict__.get('_dfh_reference_CPGP', ref_module.CPGP)
  env = dict(ref_module.__dict__)
  body = {'__doc__': ref_cls.__doc__, '__module__': __name__}
  for name, fn in vars(ref_cls).items():
    if isinstance(fn, types.FunctionType):
      body[name] = types.FunctionType(fn.__code__, env, fn.__name__, fn.__defaults__, fn.__closure__)
      body[name].__kwdefaults__ = fn.__kwdefaults__
      body[name].__doc__ = fn.__doc__
  cls = type('CPGP', (GP,), body)
  env['CPGP'] = cls
  return cls
