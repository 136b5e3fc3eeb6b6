"""``pyEXP.field.FieldGenerator`` (expui/FieldGenerator.cc): slices, line probes, point meshes and volumes of both bases
over a coefficient container -- every pixel against the ORACLE's field evaluation at the coordinates the reference
would hand its basis (r + 1e-18, cos theta, phi | R + 1e-18, z, phi), the index conventions of the returned arrays,
the midplane column of the disk basis (EmpCylSL::accumulated_midplane_eval) and the files.  GPU only."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def halo(tmp_path_factory):
    from exp_amd.basis import Basis
    d = tmp_path_factory.mktemp("fld")
    return Basis.factory(f"""
id : sphereSL
parameters :
  numr: 1000
  rmin: 0.0001
  rmax: 1.95
  Lmax: 4
  nmax: 8
  rmapping : 0.0667
  modelname: {os.path.join(GOLD, 'SLGridSph.model')}
  cachename: {d / 'SLGridSph.cache.fld'}
""")


@pytest.fixture(scope="module")
def disk(tmp_path_factory):
    from exp_amd.basis import Basis
    d = tmp_path_factory.mktemp("fldc")
    return Basis.factory(f"""
id: cylinder
parameters:
  acyl: 0.01
  hcyl: 0.001
  mmax: 3
  nmax: 12
  ncylnx: 48
  ncylny: 24
  ncylr: 600
  lmaxfid: 20
  nmaxfid: 16
  rnum: 60
  tnum: 30
  cachename: {d / 'eof.cache.fld'}
""")


def _halo_coefs(halo):
    """two coefficient sets of two different lumpy particle clouds"""
    from exp_amd.coefs import SphCoefs
    rng = np.random.default_rng(21)
    cs = SphCoefs("halo")
    for t, squash in ((0.0, 0.5), (0.25, 1.5)):
        pos = rng.normal(0, 0.3, (4000, 3))
        pos[:, 2] *= squash
        pos[:, 0] += 0.05
        cs.add(halo.createFromArray(np.full(4000, 1.0 / 4000), pos, time=t))
    return cs


def _sph_ref(oracle, halo, cstruct, x, y, z):
    """the oracle at the coordinates FieldGenerator passes to a spherical basis"""
    halo.set_coefs(cstruct)
    prm = oracle.params(scale=1.0, rmin=halo.rmin, rmax=halo.rmax)
    coef = halo.force.get_coefs()
    r = np.sqrt(x * x + y * y + z * z) + 1.0e-18
    return oracle.sph_fields(halo.grid, prm, coef, r, z / r, np.arctan2(y, x), "spherical")


def _close(got, ref, tol=1e-6):
    """float32 frames against fp64 oracle values: relative to the field's scale"""
    ref = np.asarray(ref)
    return np.abs(np.asarray(got, np.float64) - ref).max() <= tol * max(np.abs(ref).max(), 1e-300)


def test_slices_lines_points_volumes_of_the_halo(halo, oracle, tmp_path):
    from exp_amd.pyEXP.field import FieldGenerator
    cs = _halo_coefs(halo)
    times = cs.Times()
    labels = halo.getFieldLabels("spherical")
    assert halo.coordinates == "spherical" and labels[6:] == ["rad force", "mer force", "azi force"]

    # -- slice: x-z plane at y = pmin[1] (the one zero of gridsize); array index (i, j) = (x, z) ----------------
    fg = FieldGenerator(times, [-0.8, 0.1, -0.5], [0.8, 0.3, 0.5], [9, 0, 7])
    db = fg.slices(halo, cs)
    assert sorted(db) == times and sorted(db[times[0]]) == sorted(labels)
    xs, zs = np.linspace(-0.8, 0.8, 9), np.linspace(-0.5, 0.5, 7)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    for t in times:
        ref = _sph_ref(oracle, halo, cs.getCoefStruct(t), X.ravel(), np.full(X.size, 0.1), Z.ravel())
        for n, s in enumerate(labels):
            f = db[t][s]
            assert f.dtype == np.float32 and f.shape == (9, 7)
            assert _close(f.ravel(), ref[:, n]), (t, s)
    assert np.abs(db[times[0]]["potl"] - db[times[1]]["potl"]).max() > 1e-3 * np.abs(db[times[0]]["potl"]).max()
    with pytest.raises(RuntimeError, match="bad grid specification"):
        FieldGenerator(times, [0, 0, 0], [1, 1, 1], [4, 4, 4]).slices(halo, cs)
    with pytest.raises(RuntimeError, match="not in DB"):
        FieldGenerator([0.1], [-1, 0, -1], [1, 0, 1], [4, 0, 4]).slices(halo, cs)

    # -- line probe: x, y, z, arc join the field labels -----------------------------------------------------------
    beg, end, num = [-0.5, -0.2, 0.0], [0.7, 0.4, 0.3], 13
    ln = fg.lines(halo, cs, beg, end, num)
    s = np.arange(num) / (num - 1)
    px, py, pz = (beg[k] + (end[k] - beg[k]) * s for k in range(3))
    fr = ln[times[1]]
    assert sorted(fr) == sorted(labels + ["x", "y", "z", "arc"])
    assert np.allclose(fr["x"], px, atol=1e-7) and np.allclose(fr["z"], pz, atol=1e-7)
    assert np.allclose(fr["arc"], np.linalg.norm(np.subtract(end, beg)) * s, atol=1e-6)
    ref = _sph_ref(oracle, halo, cs.getCoefStruct(times[1]), px, py, pz)
    for n, lab in enumerate(labels):
        assert _close(fr[lab], ref[:, n]), lab
    with pytest.raises(RuntimeError, match="rank 3"):
        fg.lines(halo, cs, [0, 0], end, 5)
    with pytest.raises(RuntimeError, match="must be > 0"):
        fg.lines(halo, cs, beg, end, 0)

    # -- arbitrary mesh ------------------------------------------------------------------------------------------
    mesh = np.random.default_rng(4).normal(0, 0.4, (37, 3))
    pts = FieldGenerator(times, mesh).points(halo, cs)
    ref = _sph_ref(oracle, halo, cs.getCoefStruct(times[0]), *mesh.T.copy())
    for n, lab in enumerate(labels):
        assert pts[times[0]][lab].shape == (37,) and _close(pts[times[0]][lab], ref[:, n]), lab
    with pytest.raises(RuntimeError, match="Nx3"):
        FieldGenerator(times, np.zeros((5, 2)))
    with pytest.raises(RuntimeError, match="mesh constructor"):
        fg.points(halo, cs)

    # -- volume: index (i, j, k) = (x, y, z) ----------------------------------------------------------------------
    fv = FieldGenerator(times[:1], [-0.4, -0.3, -0.2], [0.4, 0.3, 0.2], [5, 4, 3])
    vol = fv.volumes(halo, cs)
    Xv, Yv, Zv = np.meshgrid(np.linspace(-0.4, 0.4, 5), np.linspace(-0.3, 0.3, 4), np.linspace(-0.2, 0.2, 3), indexing="ij")
    ref = _sph_ref(oracle, halo, cs.getCoefStruct(times[0]), Xv.ravel(), Yv.ravel(), Zv.ravel())
    assert vol[times[0]]["dens"].shape == (5, 4, 3)
    for n, lab in enumerate(labels):
        assert _close(vol[times[0]][lab].ravel(), ref[:, n]), lab

    # -- the Cartesian field type goes through crt_eval ---------------------------------------------------------------
    halo.setFieldType("cartesian")
    try:
        dbc = fg.slices(halo, cs)
        assert "x force" in dbc[times[0]] and "rad force" not in dbc[times[0]]
        assert np.allclose(dbc[times[0]]["potl"], db[times[0]]["potl"], rtol=0, atol=1e-6 * np.abs(db[times[0]]["potl"]).max())
    finally:
        halo.setFieldType("spherical")

    # -- files ------------------------------------------------------------------------------------------------------
    out = tmp_path / "fields"
    with pytest.raises(RuntimeError, match="does not exist"):
        fg.file_slices(halo, cs, "run", str(out))
    out.mkdir()
    fg.file_slices(halo, cs, "run", str(out))
    fv.file_volumes(halo, cs, "run", str(out))
    fg.file_lines(halo, cs, beg, end, num, "run", str(out))
    assert sorted(os.listdir(out)) == ["run_probe_0.txt", "run_probe_1.txt", "run_surface_0.vtr", "run_surface_1.vtr",
                                       "run_volume_0.vtr"]
    root = ET.parse(out / "run_surface_1.vtr").getroot()
    assert root.attrib["type"] == "RectilinearGrid" and root[0].attrib["WholeExtent"] == "0 8 0 6 0 0"
    arrays = {a.attrib["Name"]: np.array(a.text.split(), float) for a in root.iter("DataArray")}
    assert "dens m.gt.0" in arrays and "dens m>0" not in arrays              # (XML-sensitive characters spelled out)
    assert np.allclose(arrays["X"], xs, atol=1e-7) and np.allclose(arrays["Y"], zs, atol=1e-7) and arrays["Z"].tolist() == [0.0]
    # data[j*nx + i] = frame(i, j): x fastest
    assert np.allclose(arrays["potl"].reshape(7, 9).T, db[times[1]]["potl"], rtol=1e-7, atol=0)
    rootv = ET.parse(out / "run_volume_0.vtr").getroot()
    av = {a.attrib["Name"]: np.array(a.text.split(), float) for a in rootv.iter("DataArray")}
    assert np.allclose(av["dens"].reshape(3, 4, 5).transpose(2, 1, 0), vol[times[0]]["dens"], rtol=1e-7, atol=0)
    txt = (out / "run_probe_1.txt").read_text().split("\n")
    assert txt[0] == f"# T={times[1]:g}" and txt[1].startswith("#") and txt[1].split()[1:4] == ["arc", "azi", "force"]
    cols = sorted(fr)
    row5 = np.array(txt[4 + 5].split(), float)
    assert np.allclose(row5, [fr[k][5] for k in cols], rtol=1e-5, atol=1e-12)          # (six significant digits)


def test_disk_slices_and_the_midplane_column(disk, oracle):
    from exp_amd.coefs import CylCoefs
    from exp_amd.pyEXP.field import FieldGenerator
    rng = np.random.default_rng(5)
    n = 20000
    R = -disk.acyl * np.log(rng.random(n) * rng.random(n))           # ~ exponential disk
    ph = rng.uniform(0, 2 * np.pi, n)
    z = 2 * disk.hcyl * np.arctanh(rng.uniform(-0.999, 0.999, n))
    # a warp: the density peak leaves z = 0 at large R
    z += 1.2 * disk.hcyl * (R / (3 * disk.acyl)) ** 2 * np.cos(ph)
    pos = np.stack([R * np.cos(ph), R * np.sin(ph), z], axis=1)
    cs = CylCoefs("disk")
    cs.add(disk.createFromArray(np.full(n, 1.0 / n), pos, time=0.5))
    disk.set_coefs(cs.getCoefStruct(0.5))
    cc, ss = disk.force.get_coefs()
    labels = disk.getFieldLabels("cylindrical")
    assert disk.coordinates == "cylindrical" and labels[6:] == ["rad force", "ver force", "azi force"]
    fg = FieldGenerator([0.5], [-0.04, -0.04, 0.0005], [0.04, 0.04, 0.0], [8, 6, 0])
    db = fg.slices(disk, cs)[0.5]
    X, Y = np.meshgrid(np.linspace(-0.04, 0.04, 8), np.linspace(-0.04, 0.04, 6), indexing="ij")
    x, y = X.ravel(), Y.ravel()
    Rr, pp = np.sqrt(x * x + y * y) + 1.0e-18, np.arctan2(y, x)
    ref = oracle.cyl_fields(disk.grid, cc, ss, Rr, np.full(x.size, 0.0005), pp, "cylindrical")
    for k, lab in enumerate(labels):
        assert db[lab].shape == (8, 6) and _close(db[lab].ravel(), ref[:, k]), lab
    # midplane evaluation: a tenth field, the height of the density peak in the column |z| <= colheight * hcyl
    fg.setMidplane(True)
    fg.setColumnHeight(3.0)
    dbm = fg.slices(disk, cs)[0.5]
    assert sorted(dbm) == sorted(labels + ["midplane"]) and disk.getFieldLabels("cylindrical")[-1] == "midplane"
    num = 40
    zk = np.linspace(-3.0 * disk.hcyl, 3.0 * disk.hcyl, num)
    dz = zk[1] - zk[0]
    want = np.zeros(x.size)
    for i in range(x.size):
        d = oracle.cyl_fields(disk.grid, cc, ss, np.full(num, Rr[i]), zk, np.full(num, pp[i]), "cylindrical")[:, 2]
        kp = int(np.argmax(d))
        if kp in (0, num - 1):
            want[i] = d[kp]
        else:
            z0 = zk[0] + dz * kp
            den = d[kp - 1] - 2 * d[kp] + d[kp + 1]
            want[i] = z0 if abs(den) < 1e-16 else ((2 * z0 + dz) * d[kp - 1] * 0.5 - 2 * z0 * d[kp] + (2 * z0 - dz) * d[kp + 1] * 0.5) / den
    got = dbm["midplane"].ravel().astype(np.float64)
    inner = np.abs(want) < 2.9 * disk.hcyl
    assert inner.sum() > 10
    assert np.abs(got - want)[inner].max() <= 1e-5 * disk.hcyl
    assert np.ptp(want[inner]) > 0.01 * disk.hcyl                    # (the warp is there to be found)
    assert _close(dbm["dens"].ravel(), ref[:, 2])                     # the nine other fields are what they were
    fg.setMidplane(False)
    assert "midplane" not in fg.slices(disk, cs)[0.5] and disk.getFieldLabels("cylindrical")[-1] == "azi force"
