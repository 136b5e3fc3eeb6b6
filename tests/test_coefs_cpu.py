"""EXP's native coefficient stream format and the Coefs container (exp_amd/coefs.py): host-side
data formats either side of the hot path (src/SphericalBasis.cc:1829-1904, expui/CoefStruct.cc:
372-506, expui/Coefficients.cc:183-226, :796-838).  CPU only."""
import math
import struct

import numpy as np
import pytest


def _sets(lmax=3, nmax=5, times=(0.0, 0.1, 0.2, 0.3), seed=1):
    from exp_amd.basis import SphStruct
    from exp_amd.coefs import SphCoefs, real_rows_to_complex
    rng = np.random.default_rng(seed)
    cs = SphCoefs("halo")
    rows = {}
    for t in times:
        rows[t] = rng.standard_normal(((lmax + 1) ** 2, nmax))
        cs.add(SphStruct(lmax, nmax, 0.5, t, real_rows_to_complex(rows[t], lmax), np.zeros(3), np.eye(3)))
    return cs, rows


def test_native_roundtrip_and_layout(tmp_path):
    from exp_amd.coefs import CMAGIC, SphCoefs, complex_to_real_rows
    cs, rows = _sets()
    path = str(tmp_path / "outcoef.halo.run0")
    cs.writeNativeCoefs(path)
    raw = open(path, "rb").read()
    magic, hsize = struct.unpack("<II", raw[:8])
    assert magic == CMAGIC == 0xc0a57a2
    import yaml
    hdr = yaml.safe_load(raw[8:8 + hsize].decode())
    assert hdr["lmax"] == 3 and hdr["nmax"] == 5 and hdr["normed"] is True and hdr["scale"] == 0.5
    # the doubles are [n][row] with rows in the reference's real order (cos, sin interleaved)
    first = np.frombuffer(raw[8 + hsize:8 + hsize + 8 * 16 * 5], dtype="<f8").reshape(5, 16)
    assert np.array_equal(first.T, rows[0.0])
    back = SphCoefs.readNativeCoefs(path)
    assert back.Times() == [0.0, 0.1, 0.2, 0.3]
    assert np.array_equal(back.getAllCoefs(), cs.getAllCoefs())
    assert np.array_equal(complex_to_real_rows(back.getCoefStruct(0.2).coefs, 3), rows[0.2])
    sub = SphCoefs.readNativeCoefs(path, stride=2, tmin=0.05)
    assert sub.Times() == [0.2]


def test_legacy_header_is_normalised(tmp_path):
    """88-byte SphCoefHeader records (include/coef.H:18-25) hold un-normalised coefficients: the
    reader applies sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!) [x sqrt 2 for m > 0]."""
    from exp_amd.coefs import SphCoefs, complex_to_real_rows
    lmax, nmax = 2, 3
    rows = np.arange(9 * 3, dtype=np.float64).reshape(9, 3) + 1.0
    path = str(tmp_path / "legacy.coef")
    with open(path, "wb") as f:
        f.write(struct.pack("<64sddii", b"sphereSL", 1.25, 1.0, nmax, lmax))
        f.write(np.ascontiguousarray(rows.T, dtype="<f8").tobytes())
    c = SphCoefs.readNativeCoefs(path).getCoefStruct(1.25)
    got = complex_to_real_rows(c.coefs, lmax)
    off = 0
    for l in range(lmax + 1):
        for m in range(l + 1):
            fac = math.sqrt((2 * l + 1) / (4 * math.pi) * math.factorial(l - m) / math.factorial(l + m))
            if m:
                fac *= math.sqrt(2.0)
            for _ in range(1 if m == 0 else 2):
                assert np.allclose(got[off], rows[off] * fac, rtol=1e-14)
                off += 1


def test_interpolate_follows_the_reference():
    """Coefs::interpolate picks (lower_bound, lower_bound+1): exact at stored times, the last pair
    at/after the end, and -- strictly between two stored times -- a linear EXTRAPOLATION from the
    two sets at and after `time` (the reference's behaviour, expui/Coefficients.cc:199-213)."""
    cs, _ = _sets()
    c = {t: np.asarray(cs.getCoefStruct(t).coefs) for t in cs.Times()}
    a, ok = cs.interpolate(0.1)
    assert ok and np.allclose(a, c[0.1], rtol=0, atol=1e-15)
    a, ok = cs.interpolate(0.15)
    assert ok and np.allclose(a, 1.5 * c[0.2] - 0.5 * c[0.3], rtol=0, atol=1e-14)
    a, ok = cs.interpolate(0.25)                    # lower_bound = last element -> (0.2, 0.3)
    assert ok and np.allclose(a, 0.5 * c[0.2] + 0.5 * c[0.3], rtol=0, atol=1e-14)
    a, ok = cs.interpolate(0.35)
    assert not ok and np.allclose(a, -0.5 * c[0.2] + 1.5 * c[0.3], rtol=0, atol=1e-14)
    with pytest.raises(RuntimeError):
        cs.getCoefStruct(0.123)


def test_cylinder_native_stream(tmp_path):
    """EmpCylSL::dump_coefs_binary / CylStruct::read (exputil/EmpCylSL.cc:5868-5920,
    expui/CoefStruct.cc:258-370): magic 0xc0a57a3, YAML {time, mmax, nmax}, rows cos(m), sin(m>0);
    16-byte legacy header accepted."""
    from exp_amd.basis import CylStruct
    from exp_amd.coefs import CMAGIC_CYL, CylCoefs
    rng = np.random.default_rng(2)
    cs = CylCoefs("disk")
    for t in (0.0, 0.5, 1.0):
        cf = rng.standard_normal((4, 6)) + 1j * rng.standard_normal((4, 6))
        cf[0] = cf[0].real
        cs.add(CylStruct(3, 6, t, cf, np.zeros(3), np.eye(3)))
    path = str(tmp_path / "outcoef.disk.run0")
    cs.writeNativeCoefs(path)
    raw = open(path, "rb").read()
    assert struct.unpack("<I", raw[:4])[0] == CMAGIC_CYL == 0xc0a57a3
    back = CylCoefs.readNativeCoefs(path)
    assert back.Times() == [0.0, 0.5, 1.0]
    assert np.array_equal(back.getAllCoefs(), cs.getAllCoefs())
    a, ok = back.interpolate(0.5)
    assert ok and np.allclose(a, cs.getCoefStruct(0.5).coefs, rtol=0, atol=1e-15)
    # legacy header
    leg = str(tmp_path / "legacy.cyl")
    c = cs.getCoefStruct(0.5)
    with open(leg, "wb") as f:
        f.write(struct.pack("<dii", 0.5, 3, 6))
        for m in range(4):
            f.write(np.ascontiguousarray(c.coefs[m].real).tobytes())
            if m:
                f.write(np.ascontiguousarray(c.coefs[m].imag).tobytes())
    got = CylCoefs.readNativeCoefs(leg).getCoefStruct(0.5)
    assert np.array_equal(got.coefs, c.coefs)


def test_coefs_container_surface(tmp_path):
    """``CoefClasses::Coefs`` beyond add / Times (expui/Coefficients.H, expui/Coefficients.cc): factory by file
    type, makecoefs / addcoef, the pybind ``__call__``, setMatrix / setData on existing times only, deepcopy
    and zerodata, CompareStanzas, Power / EvenOddPower, ExtendH5Coefs with its parameter check."""
    from exp_amd.basis import CylStruct, SphStruct
    from exp_amd.coefs import Coefs, CylCoefs, SphCoefs
    rng = np.random.default_rng(4)

    def S(t, L=3, N=5):
        r = (L + 1) * (L + 2) // 2
        cf = rng.normal(size=(r, N)) + 1j * rng.normal(size=(r, N))
        for l in range(L + 1):
            cf[l * (l + 1) // 2] = cf[l * (l + 1) // 2].real          # m = 0 rows carry no sine part
        return SphStruct(L, N, 0.5, t, cf, np.zeros(3), np.eye(3))

    def C(t, M=4, N=6):
        cf = rng.normal(size=(M + 1, N)) + 1j * rng.normal(size=(M + 1, N))
        cf[0] = cf[0].real
        return CylStruct(M, N, t, cf, np.zeros(3), np.eye(3))

    sph = cyl = None
    for t in (0.0, 0.1, 0.2):
        sph, cyl = Coefs.addcoef(sph, S(t)), Coefs.addcoef(cyl, C(t))
    assert isinstance(sph, SphCoefs) and isinstance(cyl, CylCoefs)
    assert sph.getGeometry() == "sphere" and cyl.getGeometry() == "cylinder"
    with pytest.raises(RuntimeError):
        Coefs.makecoefs(object())
    # values at a time: the matrix, or an EMPTY one (expui/Coefficients.cc:683-696)
    assert sph(0.1).shape == (10, 5) and sph(0.15).shape == (0, 0)
    assert np.array_equal(sph.getData(0.1), sph(0.1).reshape(-1, order="F"))
    m = sph(0.1) * 2.0
    sph.setMatrix(0.1, m)
    assert np.array_equal(sph(0.1), m)
    sph.setData(0.1, (m * 0.5).reshape(-1, order="F"))
    assert np.array_equal(sph(0.1), m * 0.5)
    for fn in (sph.setMatrix, sph.setData):
        with pytest.raises(RuntimeError, match="not found"):
            fn(0.7, m)
    # makeKeys (expui/Coefficients.cc:750-792, :1257-1287): the keys under a leading sub-key, clamped to the orders
    assert SphCoefs("x").makeKeys() == [] and len(sph.makeKeys()) == 10 * 5 and sph.makeKeys()[:2] == [[0, 0, 0], [0, 0, 1]]
    assert sph.makeKeys([2]) == [[2, m, n] for m in range(3) for n in range(5)]
    assert sph.makeKeys([9, 7]) == [[3, 3, n] for n in range(5)]                       # l -> Lmax, m -> l
    assert cyl.makeKeys() == [[m, n] for m in range(5) for n in range(6)] and cyl.makeKeys([11]) == [[4, n] for n in range(6)]
    with pytest.raises(RuntimeError, match="rank 0, 1 or 2"):
        sph.makeKeys([1, 1, 1])
    with pytest.raises(RuntimeError, match="rank 1"):
        cyl.makeKeys([1, 1])
    # power: |c|^2 summed over m and the radial window
    P = sph.Power(1, 4)
    a2 = np.abs(sph(0.0)[:, 1:4]) ** 2
    assert P.shape == (3, 4) and np.allclose(P[0], [a2[0].sum(), a2[1:3].sum(), a2[3:6].sum(), a2[6:10].sum()])
    assert np.allclose(cyl.Power()[2], (np.abs(cyl(0.2)) ** 2).sum(axis=1))
    ev, od = cyl.EvenOddPower(2)
    assert np.allclose(ev + od, cyl.Power()) and np.allclose(od[1], (np.abs(cyl(0.1)[:, 4:]) ** 2).sum(axis=1))
    with pytest.raises(RuntimeError, match="ncylodd"):
        cyl.EvenOddPower()
    # copies
    cp = sph.deepcopy()
    assert sph.CompareStanzas(cp) and cp.coefs[0.0] is not sph.coefs[0.0]
    cp.zerodata()
    assert np.abs(cp.getAllCoefs()).max() == 0.0 and not sph.CompareStanzas(cp) and cp.Times() == sph.Times()
    cp.clear()
    assert cp.Times() == [] and cp.Power().shape == (0, 0)
    # files: factory picks the class from the HDF5 geometry attribute or the native magic number
    for obj, tag in ((sph, "s"), (cyl, "c")):
        obj.setName("comp" + tag)
        h5, nat = str(tmp_path / (tag + ".h5")), str(tmp_path / (tag + ".native"))
        obj.setUnits([("length", "kpc", 1.0), ("mass", "Msun", 1.0e10), ("time", "Gyr", 1.0), ("G", "mixed", 43007.1)])
        obj.WriteH5Coefs(h5)
        obj.writeNativeCoefs(nat)
        for path in (h5, nat):
            back = Coefs.factory(path)
            assert type(back) is type(obj) and back.Times() == obj.Times()
            assert np.abs(back.getAllCoefs() - obj.getAllCoefs()).max() <= 1e-15
        assert Coefs.factory(h5).getName() == "comp" + tag
        ext = type(obj)()
        ext.add(S(0.3) if tag == "s" else C(0.3))
        ext.ExtendH5Coefs(h5)                     # (a fresh container assumes the default forceID)
        assert Coefs.factory(h5).Times() == [0.0, 0.1, 0.2, 0.3]
        assert Coefs.factory(h5, stride=2).Times() == [0.0, 0.2] and Coefs.factory(h5, tmin=0.05, tmax=0.25).Times() == [0.1, 0.2]
        wrong = type(obj)()
        wrong.add(S(0.4, L=2) if tag == "s" else C(0.4, M=3))
        with pytest.raises(RuntimeError, match="parameter check failed"):
            wrong.ExtendH5Coefs(h5)
    with pytest.raises(RuntimeError, match="does not exist"):
        Coefs.factory(str(tmp_path / "nothing"))


def test_coefstruct_surface():
    """``CoefStruct`` / ``SphStruct`` / ``CylStruct`` as pyEXP exposes them (pyEXP/CoefWrappers.cc:720-1000; expui/CoefStruct.H):
    default construction, assign / create, the flat column-major store, in-place access, time / centre / orientation,
    deep copies, and the gravitational constant of a free set against that of its container."""
    from exp_amd.basis import CylStruct, SphStruct
    from exp_amd.coefs import SphCoefs
    rng = np.random.default_rng(9)
    s = SphStruct()
    assert (s.lmax, s.nmax, s.scale, s.geometry, s.time) == (0, 0, 1.0, "sphere", 0.0)
    with pytest.raises(RuntimeError, match="nmax must be >0"):
        s.create()
    mat = rng.normal(size=(6, 4)) + 1j * rng.normal(size=(6, 4))
    s.assign(mat, 2, 4)
    assert (s.lmax, s.nmax) == (2, 4) and np.array_equal(s.coefs, mat)
    assert np.array_equal(s.getCoefs(), mat.reshape(-1, order="F"))                 # Eigen's column-major store
    with pytest.raises(ValueError, match="size does not match"):
        s.setCoefs(np.zeros(5))
    s.setCoefs(2.0 * s.getCoefs())
    assert np.array_equal(s.coefs, 2.0 * mat)
    ref = s.setCoefs()                                       # read-write access: changes land in the structure
    ref[1, 2] = 7.0 - 1.0j
    assert s.coefs[1, 2] == 7.0 - 1.0j
    s.setCoefTime(0.75); s.setCoefCenter([0.1, 0.2, 0.3]); s.setCoefRotation(2.0 * np.eye(3))
    assert s.getCoefTime() == 0.75 and s.time == 0.75 and np.array_equal(s.center, [0.1, 0.2, 0.3])
    assert np.array_equal(s.getCoefCenter(), s.center) and np.array_equal(s.orient, 2.0 * np.eye(3))
    assert np.array_equal(s.getCoefRotation(), s.orient)
    c = s.deepcopy()
    c.coefs[0, 0] = 0.0
    assert s.coefs[0, 0] != 0.0 and c.time == 0.75
    s.zerodata()
    assert not s.coefs.any() and s.coefs.shape == (6, 4)
    z = SphStruct(lmax=1, nmax=3)
    z.create()
    assert z.coefs.shape == (3, 3) and not z.coefs.any()
    # G: the structure's own value until a container owns it (expui/CoefStruct.cc:16-22)
    z.setGravConstant(4.3)
    assert z.getGravConstant() == 4.3
    cs = SphCoefs("x")
    cs.setUnits("G", "mixed", 43007.1)
    cs.add(z)
    assert z.getGravConstant() == cs.getGravConstant() == float(np.float32(43007.1))
    y = CylStruct()
    assert (y.mmax, y.nmax, y.geometry) == (0, 0, "cylinder")
    m2 = rng.normal(size=(3, 5)) + 1j * rng.normal(size=(3, 5))
    y.assign(m2, 2, 5)
    assert np.array_equal(y.getCoefs(), m2.reshape(-1, order="F"))
    with pytest.raises(ValueError):
        y.assign(m2, 3, 5)
    with pytest.raises(RuntimeError, match="nmax must be >0"):
        CylStruct().create()
