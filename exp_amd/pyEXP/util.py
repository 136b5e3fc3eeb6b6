"""``pyEXP.util`` (pyEXP/UtilWrappers.cc) -- the centre estimators and the particle iterator of exp_amd.util."""
from ..util import getCenterOfMass, getDensityCenter, getVersionInfo, particleIterator  # noqa: F401
