"""``pyEXP.basis`` (pyEXP/BasisWrappers.cc) -- the classes of exp_amd.basis under the reference's names."""
from ..basis import (Basis, BiorthBasis, CovarianceReader, Cylindrical, SphericalSL)  # noqa: F401
