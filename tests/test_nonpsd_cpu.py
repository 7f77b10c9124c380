"""CPU: GPs with kernels that are not guaranteed PSD ('project_first' / 'try_before_project',
the Cartesian-product GP class) through the mirrors over the stand-in engine, against the real
reference's outputs; the oracle's restatement of the projection against its definition."""
import numpy as np

from nonpsd_replay import check
from oracle import ref_numpy as O


def test_nonpsd_gp_and_cpgp_against_reference_outputs(monkeypatch):
  from oracle_engine import patch_engine
  from dragonfly_amd import cartesian_product_gp    # noqa: F401  (imports gp_core before patching)
  patch_engine(monkeypatch)
  check(tol=1e-10)


def test_oracle_projection_definition():
  rs = np.random.RandomState(3)
  A = rs.randn(30, 30)
  A = (A + A.T) / 2
  P = O.project_symmetric_to_psd_cone(A, epsilon=0.1)
  w = np.linalg.eigvalsh(P)
  assert w.min() > 0.1 - 1e-12 and np.allclose(P, P.T)
  wa, V = np.linalg.eigh(A)
  assert np.allclose(P, (V * np.maximum(wa, 0.1)).dot(V.T))
