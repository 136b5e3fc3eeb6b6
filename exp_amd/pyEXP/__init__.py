"""``import exp_amd.pyEXP as pyEXP``: the pyEXP sub-modules this path covers, under their names.

    pyEXP.basis.Basis.factory(yaml) / SphericalSL / Cylindrical / CovarianceReader
    pyEXP.coefs.Coefs.factory(file) / SphCoefs / CylCoefs / SphStruct / CylStruct
    pyEXP.field.FieldGenerator(times, lower, upper, gridsize) / (times, mesh)
    pyEXP.util.getDensityCenter(reader, stride, Nsort, Ndens) / getCenterOfMass(reader) / particleIterator(reader, f)
    pyEXP.read.ParticleReader.createReader(type, files) / PSPout / PSPspl / PSPhdf5 / GadgetNative / GadgetHDF5 / Tipsy

so that a script written against the reference's Python module (tests/Halo/createCoefs.py,
tests/Halo/changeCoefs.py, tests/Disk/cyl_basis.py) runs with the import line changed.  Everything else
of pyEXP (mSSA) is outside this repository's scope and raises on access."""
from . import basis, coefs, field, read, util


def __getattr__(name):
    raise AttributeError(f"exp_amd.pyEXP has no sub-module <{name}>: only basis, coefs, field, read and util are in scope")
