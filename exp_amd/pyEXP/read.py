"""``pyEXP.read`` (pyEXP/ParticleReaderWrappers.cc) -- the particle readers of exp_amd.reader under the reference's names."""
from ..reader import (GadgetNative, Particle, ParticleReader, PSP, PSPout, PSPspl, Tipsy)  # noqa: F401


def __getattr__(name):
    if name in ("GadgetHDF5", "PSPhdf5"):
        from .. import reader_h5
        return getattr(reader_h5, name)
    raise AttributeError(f"exp_amd.pyEXP.read has no attribute <{name}>")
