"""EXP's SLGridSph HDF5 cache format (exputil/SLGridMP2.cc:490-696) through the HDF5 C library:
write -> inspect with the HDF5 tools -> read back; the older [numr][nmax] matrix layout is
transposed on reading.  CPU only; skipped where the shim could not be built (no hdf5.h)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests.conftest import make_grid


@pytest.fixture(scope="module")
def h5():
    from exp_amd import h5cache
    if not h5cache.available():
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.run(["make", "-s", "h5"], cwd=root, check=False)
    if not h5cache.available():
        pytest.skip("HDF5 C headers/library not available")
    return h5cache


def test_slgrid_cache_roundtrip_and_layout(h5, tmp_path):
    model, g = make_grid("plummer", 4, 8, 400)
    path = str(tmp_path / "SLGridSph.cache.run0")
    h5.write_slgrid_cache(path, g, "SLGridSph.model")
    hdr = h5.read_slgrid_header(path)
    assert hdr["geometry"] == "sphere" and hdr["forceID"] == "SLGridSph" and hdr["version"] == "1.0"
    assert hdr["model"] == "SLGridSph.model"
    assert (hdr["lmax"], hdr["nmax"], hdr["numr"], hdr["cmap"], hdr["diverge"]) == (4, 8, 400, 1, 0)
    assert hdr["rmin"] == g.rmin and hdr["rmax"] == g.rmax and hdr["rmapping"] == g.rmap
    back = h5.read_slgrid_cache(path, model, check={"lmax": 4, "nmax": 8, "numr": 400,
                                                    "rmapping": g.rmap, "model": "SLGridSph.model"})
    for k in ("ev", "ef", "xi", "r", "p0", "d0"):
        assert np.array_equal(getattr(back, k), getattr(g, k)), k
    assert back.dxi == g.dxi and back.xmin == g.xmin
    with pytest.raises(RuntimeError):
        h5.read_slgrid_cache(path, model, check={"nmax": 9})
    # structure as the reference writes it: Harmonic/<l>/{ev [nmax], ef [nmax, numr]}
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        txt = subprocess.run([h5dump, "-H", path], capture_output=True, text=True).stdout
        assert 'GROUP "Harmonic"' in txt and 'GROUP "4"' in txt
        assert "DATASPACE  SIMPLE { ( 8, 400 ) / ( 8, 400 ) }" in txt
        assert 'ATTRIBUTE "rmapping"' in txt and "H5T_VARIABLE" in txt and "H5T_CSET_UTF8" in txt


def test_old_matrix_layout_is_transposed(h5, tmp_path):
    """Caches written through the older HighFive Eigen path hold ef as [numr][nmax] (what the
    "Version" attribute guards against, exputil/SLGridMP2.cc:550-560): the reader accepts both,
    and refuses shapes that are neither."""
    model, g = make_grid("plummer", 4, 8, 400)
    path = str(tmp_path / "old.cache")
    h5.write_slgrid_cache(path, g, "m", old_layout=True)
    back = h5.read_slgrid_cache(path, model)
    assert np.array_equal(back.ef, g.ef) and np.array_equal(back.ev, g.ev)
    import ctypes
    ev = np.zeros((g.lmax + 1, 7)); ef = np.zeros((g.lmax + 1, 7, 400))
    rc = h5._load().exp_h5_slgrid_read_tables(path.encode(), g.lmax, 7, 400,
                                              ev.ctypes.data_as(ctypes.c_void_p),
                                              ef.ctypes.data_as(ctypes.c_void_p))
    assert rc != 0


def test_empcyl_cache_roundtrip(h5, tmp_path):
    """EmpCylSL::WriteH5Cache / ReadH5Cache layout (exputil/EmpCylSL.cc:7378-7640)."""
    from exp_amd.empcyl import build_empcyl
    g = build_empcyl(mmax=2, norder=3, numx=16, numy=8, lmaxfid=8, nmaxfid=6, numr=300, rnum=40, tnum=20)
    path = str(tmp_path / ".eof.cache.run0")
    h5.write_empcyl_cache(path, g, lmaxfid=8, nmaxfid=6, cmass=0.25)
    hdr = h5.read_empcyl_header(path)
    assert hdr["geometry"] == "cylinder" and hdr["forceID"] == "Cylinder" and hdr["model"] == "Exponential"
    assert (hdr["mmax"], hdr["numx"], hdr["numy"], hdr["nmax"], hdr["lmaxfid"], hdr["nmaxfid"]) == (2, 16, 8, 3, 8, 6)
    assert hdr["ascl"] == g.ascale and hdr["hscl"] == g.hscale and hdr["cmass"] == 0.25
    back = h5.read_empcyl_cache(path, check={"mmax": 2, "nmax": 3, "ascl": g.ascale})
    assert np.array_equal(back.tab[:3], g.tab[:3])                       # cosine tables, all m
    assert np.array_equal(back.tab[3:, 1:], g.tab[3:, 1:])               # sine tables exist for m >= 1
    assert np.all(back.tab[3:, 0] == 0.0)
    assert np.array_equal(back.dens[0], g.dens[0]) and np.array_equal(back.dens[1, 1:], g.dens[1, 1:])
    for k in ("rtable", "xmin", "xmax", "dx", "ymin", "ymax", "dy"):
        assert getattr(back, k) == pytest.approx(getattr(g, k), rel=1e-15), k
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        txt = subprocess.run([h5dump, "-H", path], capture_output=True, text=True).stdout
        assert 'GROUP "Cosine"' in txt and 'GROUP "Sine"' in txt and 'DATASET "densC"' in txt
        assert "DATASPACE  SIMPLE { ( 17, 9 ) / ( 17, 9 ) }" in txt
        assert txt.count('DATASET "potS"') == 2 * 3 and txt.count('DATASET "potC"') == 3 * 3


def test_sph_coefficient_file_roundtrip(h5, tmp_path):
    """pyEXP's HDF5 coefficient file layout (expui/Coefficients.cc:228-330, :841-944, :3100-3163)."""
    from exp_amd.basis import SphStruct
    from exp_amd.coefs import SphCoefs
    rng = np.random.default_rng(3)
    cs = SphCoefs("dark halo")
    for k, t in enumerate((0.0, 0.02, 0.04)):
        cf = rng.standard_normal((10, 5)) + 1j * rng.standard_normal((10, 5))
        cs.add(SphStruct(3, 5, 0.7, t, cf, np.array([0.1 * k, 0.0, -0.2]), np.eye(3) * (1 + k)))
    path = str(tmp_path / "outcoef.halo.run0.h5")
    with pytest.raises(RuntimeError, match="expected 4 units"):          # Coefs::WriteH5Units (expui/Coefficients.cc:152-160)
        cs.WriteH5Coefs(path, config="id: sphereSL")
    with pytest.raises(RuntimeError, match="incompatible or not recognized"):
        cs.setUnits("length", "furlong", 1.0)
    cs.setUnits("Len", "kiloparsec", 1.0)                   # aliases are stored under the canonical spellings
    cs.setUnits([("M", "solar_mass", 1.0e10), ("t", "Gyr", 1.0)])
    cs.setUnits("G", "mixed", 43007.1)                      # (an update of the default entry, not a fifth unit)
    assert cs.getUnits() == [("G", "mixed", float(np.float32(43007.1))), ("length", "kpc", 1.0), ("mass", "Msun", 1.0e10),
                             ("time", "Gyr", 1.0)]
    assert cs.getGravConstant() == float(np.float32(43007.1))
    assert "kpc" in cs.getAllowedUnitNames("length") and cs.getAllowedTypeAliases("mass") == ["M", "Mass", "m", "mass"]
    assert cs.getAllowedUnitTypes() == ["mass", "length", "time", "velocity", "G"]
    cs.WriteH5Coefs(path, config="id: sphereSL")
    back = SphCoefs.readH5Coefs(path)
    assert back.name == "dark halo" and back.Times() == cs.Times()
    assert back.getUnits() == cs.getUnits()                 # Coefs::ReadH5Units
    for t in cs.Times():
        a, b = cs.getCoefStruct(t), back.getCoefStruct(t)
        assert np.array_equal(a.coefs, b.coefs) and np.array_equal(a.ctr, b.ctr) and np.array_equal(a.rot, b.rot)
        assert (b.lmax, b.nmax, b.scale, b.time) == (3, 5, 0.7, t)
    assert SphCoefs.readH5Coefs(path, stride=2).Times() == [0.0, 0.04]
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        txt = subprocess.run([h5dump, "-H", path], capture_output=True, text=True).stdout
        assert 'GROUP "snapshots"' in txt and 'GROUP "00000002"' in txt and 'DATASET "count"' in txt
        assert 'H5T_IEEE_F64LE "r"' in txt and 'H5T_IEEE_F64LE "i"' in txt
        assert "DATASPACE  SIMPLE { ( 10, 5 ) / ( 10, 5 ) }" in txt
        assert 'ATTRIBUTE "CoefficientOutputVersion"' in txt and 'ATTRIBUTE "Rotation"' in txt


def test_cyl_coefficient_file_roundtrip(h5, tmp_path):
    """Cylindrical HDF5 coefficient files (expui/Coefficients.cc:1075-1176, :1323-1405): attributes
    mmax / nmax / forceID / geometry "cylinder", (mmax+1) x nmax complex snapshots; the m = 0 row
    comes back real; a spherical file is refused by the cylindrical reader and vice versa."""
    from exp_amd.basis import CylStruct
    from exp_amd.coefs import CylCoefs, SphCoefs
    rng = np.random.default_rng(4)
    cs = CylCoefs("star disk")
    for k, t in enumerate((0.0, 0.01, 0.02, 0.03)):
        cf = rng.standard_normal((5, 7)) + 1j * rng.standard_normal((5, 7))
        cf[0] = cf[0].real
        cs.add(CylStruct(4, 7, t, cf, np.array([0.0, 0.1 * k, 0.0]), np.eye(3)))
    path = str(tmp_path / "outcoef.disk.run0.h5")
    cs.setUnits([("length", "kpc", 1.0), ("mass", "Msun", 1.0e10), ("time", "Gyr", 1.0), ("G", "mixed", 43007.1)])
    cs.removeUnits("time")
    cs.setUnits("velocity", "km/s", 1.0)                    # (length, mass, velocity, G) is the other accepted set
    cs.WriteH5Coefs(path, config="id: cylinder")
    back = CylCoefs.readH5Coefs(path)
    assert back.name == "star disk" and back.Times() == cs.Times()
    assert back.getUnits() == cs.getUnits() and [u[0] for u in back.getUnits()] == ["G", "length", "mass", "velocity"]
    for t in cs.Times():
        a, b = cs.getCoefStruct(t), back.getCoefStruct(t)
        assert np.array_equal(a.coefs, b.coefs) and np.array_equal(a.ctr, b.ctr) and np.array_equal(a.rot, b.rot)
        assert (b.mmax, b.nmax, b.time) == (4, 7, t)
    assert CylCoefs.readH5Coefs(path, tmin=0.005, tmax=0.025).Times() == [0.01, 0.02]
    a, ok = back.interpolate(0.02)
    assert ok and np.allclose(a, cs.getCoefStruct(0.02).coefs, rtol=0, atol=1e-15)
    with pytest.raises(RuntimeError):
        SphCoefs.readH5Coefs(path)
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        txt = subprocess.run([h5dump, "-H", path], capture_output=True, text=True).stdout
        assert 'ATTRIBUTE "mmax"' in txt and 'ATTRIBUTE "lmax"' not in txt and 'ATTRIBUTE "scale"' not in txt
        assert "DATASPACE  SIMPLE { ( 5, 7 ) / ( 5, 7 ) }" in txt and 'GROUP "00000003"' in txt


def test_covariance_store_roundtrip_and_layout(h5, tmp_path):
    """``SubsampleCovariance`` files (expui/Covariance.cc, include/Covariance.H; written by
    ``Spherical::writeCoefCovariance``, expui/BiorthBasis.H:433-463): create, extend, read back --
    summed upper triangles (the default), per-sample upper triangles, per-sample diagonals; the
    structure the reference's reader walks is checked with h5dump."""
    rng = np.random.default_rng(5)
    T, ltot, nmax = 4, 6, 5
    counts = rng.integers(5, 50, T).astype(np.int32)
    masses = rng.uniform(0.1, 1.0, T)
    mean = rng.standard_normal((T, ltot, nmax)) + 1j * rng.standard_normal((T, ltot, nmax))
    a = rng.standard_normal((T, ltot, nmax, nmax)) + 1j * rng.standard_normal((T, ltot, nmax, nmax))
    covr = a + np.conj(np.swapaxes(a, 2, 3))               # Hermitian like g g^dagger sums
    iu = np.triu_indices(nmax)
    for summed, covar, tag in ((True, True, "sum"), (False, True, "full"), (False, False, "diag")):
        path = str(tmp_path / f"coefcovar.halo.{tag}.h5")
        assert h5.covar_append(path, "SphereSL", 0, (2, nmax), (1.0, 1e-4, 1.95), 0.1234567891, counts, masses,
                               mean, covr, summed=summed, covar=covar)
        assert h5.covar_append(path, "SphereSL", 0, (2, nmax), (1.0, 1e-4, 1.95), 0.5, counts, masses, 2 * mean,
                               2 * covr, summed=summed, covar=covar)        # extendCoefCovariance
        # nothing to write: no file is touched, False comes back
        assert not h5.covar_append(path, "SphereSL", 0, (2, nmax), (1.0, 1e-4, 1.95), 0.9, 0 * counts, masses,
                                   mean, covr, summed=summed, covar=covar)
        rd = h5.SubsampleCovariance(path)
        assert rd.BasisID == "SphereSL" and rd.Times() == [0.12345679, 0.5]          # times rounded to 1e-8
        c, m, mu, cv = rd.getCoefCovariance(0.12345679)
        assert np.array_equal(c, counts) and np.array_equal(m, masses) and np.array_equal(mu, mean)
        want = np.zeros_like(covr)
        if covar:
            src = covr.sum(0, keepdims=True) / T if summed else covr
            src = np.repeat(src, T, axis=0) if summed else src
            want[:, :, iu[0], iu[1]] = src[:, :, iu[0], iu[1]]
            want[:, :, iu[1], iu[0]] = src[:, :, iu[0], iu[1]]              # the reader's unconjugated mirror
        else:
            idx = np.arange(nmax)
            want[:, :, idx, idx] = covr[:, :, idx, idx]
        assert np.abs(cv - want).max() <= 1e-14 * np.abs(want).max()
        assert np.array_equal(rd.getCoefCovariance(0.5)[2], 2 * mean)
        with pytest.raises(RuntimeError):
            rd.getCoefCovariance(0.7)
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        txt = subprocess.run([h5dump, "-H", str(tmp_path / "coefcovar.halo.sum.h5")], capture_output=True, text=True).stdout
        for s in ('ATTRIBUTE "CovarianceFileVersion"', 'ATTRIBUTE "BasisID"', 'ATTRIBUTE "FloatSize"', 'ATTRIBUTE "lmax"',
                  'ATTRIBUTE "rmax"', 'DATASET "count"', 'GROUP "snapshots"', 'GROUP "00000001"', 'ATTRIBUTE "Time"',
                  'ATTRIBUTE "sampleSize"', 'ATTRIBUTE "angularSize"', 'ATTRIBUTE "rankSize"', 'DATASET "sampleCounts"',
                  'DATASET "coefficients_real"', 'DATASET "covariance_imag_total"'):
            assert s in txt, s
        assert f"( {ltot * nmax * (nmax + 1) // 2}, 1 )" in txt           # summed upper triangles, [n][1]


def test_even_odd_function_split_of_the_disk_basis():
    """``ncylodd`` (EmpCylSL's constructor argument ``nodd``, exputil/EmpCylSL.cc:178-185; pyEXP's default 9,
    expui/BiorthBasis.cc:1389; the n-body default nmax/4, src/Cylinder.cc:553-555): norder - nodd vertically symmetric
    functions first, then nodd antisymmetric ones (compute_eof_grid, :1680-1760), each the largest-variance
    combinations within its parity; out of range = no split.  The symmetric functions of a split basis are the
    symmetric functions of the plain one, in the same order."""
    from exp_amd.empcyl import build_empcyl
    kw = dict(mmax=2, norder=8, numx=32, numy=16, acyl=0.01, hcyl=0.001, lmaxfid=16, nmaxfid=12, numr=400, rnum=40, tnum=24)
    plain, split, out_of_range = build_empcyl(**kw), build_empcyl(nodd=3, **kw), build_empcyl(nodd=9, **kw)

    def parity(g, m, n):           # +1: symmetric about the plane
        a = g.tab[0, m, n]
        return 1 if np.abs(a - a[:, ::-1]).max() < np.abs(a + a[:, ::-1]).max() else -1

    assert np.array_equal(out_of_range.tab, plain.tab)
    for m in range(3):
        assert [parity(split, m, n) for n in range(8)] == [1] * 5 + [-1] * 3
        ev = [n for n in range(8) if parity(plain, m, n) == 1][:5]
        for k, n in enumerate(ev):
            for kind in (0, 1, 2):
                ref = plain.tab[kind, m, n]
                assert np.abs(split.tab[kind, m, k] - ref).max() <= 1e-12 * np.abs(ref).max()
        # the z-force of a symmetric potential is antisymmetric, and the other way round
        assert np.abs(split.tab[2, m, 0] + split.tab[2, m, 0][:, ::-1]).max() <= 1e-12 * np.abs(split.tab[2, m, 0]).max()
        assert np.abs(split.tab[2, m, 7] - split.tab[2, m, 7][:, ::-1]).max() <= 1e-12 * np.abs(split.tab[2, m, 7]).max()
    # biorthogonality survives the selection: density x potential of different functions integrates to ~0
    # (checked on the device by Cylindrical.orthoCheck; here only that the split functions are not degenerate)
    for m in range(3):
        flat = split.tab[0, m].reshape(8, -1)
        assert np.linalg.matrix_rank(flat) == 8


def test_azimuthal_knots_and_shifted_conditioning_of_the_disk_basis():
    """``pnum`` / ``ashift`` (generate_eof's phi quadrature and ``dcond``, exputil/EmpCylSL.cc:2455-2500, src/Cylinder.cc:325-348):
    several knots without a shift reproduce the one-knot basis (an axisymmetric target weights cos^2 and sin^2 alike);
    with a shift the sine functions of m >= 1 part from the cosine functions while m = 0 -- never shifted -- stays put."""
    from exp_amd.empcyl import build_empcyl
    kw = dict(mmax=2, norder=6, numx=24, numy=12, acyl=0.01, hcyl=0.001, lmaxfid=12, nmaxfid=10, numr=300, rnum=30, tnum=20)
    one = build_empcyl(**kw)
    assert np.array_equal(one.tab[3:, 1:], one.tab[:3, 1:])
    many = build_empcyl(pnum=8, **kw)
    scale = np.abs(one.tab[:3]).max()
    assert np.abs(many.tab[:3] - one.tab[:3]).max() <= 1e-10 * scale
    assert np.abs(many.tab[3:, 1:] - many.tab[:3, 1:]).max() <= 1e-10 * scale
    shifted = build_empcyl(pnum=8, ashift=0.5, **kw)
    assert np.array_equal(shifted.tab[:3, 0], one.tab[:3, 0]) and np.all(shifted.tab[3:, 0] == 0.0)
    assert np.abs(shifted.tab[3:, 1:] - shifted.tab[:3, 1:]).max() > 0.05 * scale
    assert np.abs(shifted.tab[:3, 1:] - one.tab[:3, 1:]).max() > 0.01 * scale
    assert np.isfinite(shifted.tab).all() and np.isfinite(shifted.dens).all()
    # pnum below one is one (src/Cylinder.cc:168)
    assert np.array_equal(build_empcyl(pnum=0, **kw).tab, one.tab)
