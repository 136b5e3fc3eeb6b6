"""``pyEXP.basis`` (pyEXP/BasisWrappers.cc) -- the classes of exp_amd.basis under the reference's names."""
from ..basis import (AccelFunc, AllTimeAccel, Basis, BiorthBasis, CovarianceReader, Cylindrical,  # noqa: F401
                     IntegrateOrbits, SingleTimeAccel, SphericalSL)
