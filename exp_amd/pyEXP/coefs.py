"""``pyEXP.coefs`` (pyEXP/CoefWrappers.cc) -- the containers of exp_amd.coefs under the reference's names."""
from ..basis import CylStruct, SphStruct                      # noqa: F401
from ..coefs import Coefs, CylCoefs, SphCoefs                 # noqa: F401
