"""``pyEXP.field`` (pyEXP/FieldWrappers.cc) -- the field generator of exp_amd.field under the reference's name."""
from ..field import FieldGenerator  # noqa: F401
