"""The pyEXP.basis-shaped front end, exercised the way the reference's own tests do
(tests/Halo/sph_basis.py, tests/Halo/createCoefs.py, tests/Disk/cyl_basis.py) -- plus the numeric
checks those tests lack.  GPU only."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# tests/Halo/createCoefs.py configuration (pcavar/subsamp dropped: variance analysis is out of scope)
HALO_CFG = """
---
id : sphereSL
parameters :
  numr: 1000
  rmin: 0.0001
  rmax: 1.95
  Lmax: 2
  nmax: 10
  rmapping : 0.0667
  modelname: {model}
  cachename: {cache}
...
"""


@pytest.fixture(scope="module")
def halo_basis(tmp_path_factory):
    from exp_amd.basis import Basis
    d = tmp_path_factory.mktemp("cache")
    cfg = HALO_CFG.format(model=os.path.join(GOLD, "SLGridSph.model"),
                          cache=str(d / "SLGridSph.cache.run0"))
    return Basis.factory(cfg), cfg


def test_factory_and_cache_roundtrip(halo_basis):
    """tests/Halo/sph_basis.py: build from the reference's model file (orthoTest passes at
    construction, otherwise the ctor raises), read the cache back."""
    from exp_amd.basis import Basis
    basis, cfg = halo_basis
    info = basis.cacheInfo()
    assert info["lmax"] == 2 and info["nmax"] == 10 and info["numr"] == 1000
    assert info["rmapping"] == pytest.approx(0.0667)
    again = Basis.factory(cfg)                     # second construction reads the cache
    assert np.array_equal(again.grid.ef, basis.grid.ef)
    oc = basis.orthoCheck(400)
    assert max(np.abs(m - np.eye(10)).max() for m in oc) < 1e-2


def test_unknown_key_rejected(halo_basis):
    from exp_amd.basis import Basis
    _, cfg = halo_basis
    with pytest.raises(RuntimeError):
        Basis.factory(cfg.replace("  Lmax: 2", "  Lmax: 2\n  bogus_key: 1"))


def test_create_from_array_layouts(halo_basis, oracle):
    """tests/Halo/createCoefs.py: lists, ndarray converted from lists, list of arrays, [3,N] array.
    All layouts must give the same coefficients, and those must match the oracle."""
    basis, _ = halo_basis
    random.seed(7)
    mass, xpos, ypos, zpos = [], [], [], []
    for _ in range(100):
        mass.append(0.01)
        xpos.append(random.random() * 2.0 - 1.0)
        ypos.append(random.random() * 2.0 - 1.0)
        zpos.append(random.random() * 2.0 - 1.0)
    coef1 = basis.createFromArray(mass, [xpos, ypos, zpos], time=3.0)
    data = np.array([xpos, ypos, zpos])
    coef2 = basis.createFromArray(np.array(mass), data, time=3.1)
    coef3 = basis.createFromArray(np.array(mass), [np.array(xpos), np.array(ypos), np.array(zpos)],
                                  time=3.2)
    coef4 = basis.createFromArray(np.array(mass), data.T.copy(), time=3.3)       # [N,3]
    assert coef1.time == 3.0 and coef1.coefs.shape == (6, 10)
    for c in (coef2, coef3, coef4):
        assert np.abs(c.coefs - coef1.coefs).max() <= 1e-13 * np.abs(coef1.coefs).max()
    # numeric check against the oracle (real rows -> complex packing, expui/BiorthBasis.cc:482-517)
    g = basis.grid
    prm = oracle.params(scale=1.0, rmin=basis.rmin, rmax=basis.rmax)
    ref, used = oracle.sph_accumulate(g, prm, data.T, np.array(mass))
    assert basis.used == used
    packed = np.zeros((6, 10), dtype=complex)
    L0 = L1 = 0
    for l in range(3):
        for m in range(l + 1):
            if m == 0:
                packed[L0] = ref[L1]; L1 += 1
            else:
                packed[L0] = ref[L1] + 1j * ref[L1 + 1]; L1 += 2
            L0 += 1
    assert np.abs(coef1.coefs - packed).max() <= 1e-10 * np.abs(packed).max()


def test_center_rotation_and_incremental_accumulation(halo_basis):
    basis, _ = halo_basis
    rng = np.random.default_rng(3)
    pos = rng.normal(0, 0.3, (400, 3))
    m = np.full(400, 1.0 / 400)
    full = basis.createFromArray(m, pos).coefs.copy()
    # two addFromArray batches == one (expui/BiorthBasis.cc:4616-4738)
    basis.initFromArray()
    basis.addFromArray(m[:150], pos[:150])
    basis.addFromArray(m[150:], pos[150:])
    two = basis.makeFromArray(1.0).coefs
    assert np.abs(two - full).max() <= 1e-12 * np.abs(full).max()
    # centre: shifting the particles and the centre together changes nothing
    ctr = np.array([0.1, -0.2, 0.05])
    shifted = basis.createFromArray(m, pos + ctr, center=ctr).coefs
    assert np.abs(shifted - full).max() <= 1e-10 * np.abs(full).max()


def test_get_accel_matches_oracle_inside_rmax(halo_basis, oracle):
    basis, _ = halo_basis
    rng = np.random.default_rng(5)
    pos = rng.normal(0, 0.3, (2000, 3))
    pos[:, 2] *= 0.5
    m = np.full(2000, 1.0 / 2000)
    coefs = basis.createFromArray(m, pos)
    basis.set_coefs(coefs)
    test = rng.normal(0, 0.4, (500, 3))
    test = test[np.linalg.norm(test, axis=1) < 0.95 * basis.rmax]
    acc = basis.getAccel(test)
    prm = oracle.params(scale=1.0, rmin=basis.rmin, rmax=basis.rmax)
    ref, _ = oracle.sph_accumulate(basis.grid, prm, pos, m)
    a_ref, _ = oracle.sph_accel(basis.grid, prm, test, ref)
    assert np.abs(acc - a_ref).max() <= 1e-9 * np.linalg.norm(a_ref, axis=1).max()
    one = basis.getAccel(test[0, 0], test[0, 1], test[0, 2])
    assert np.allclose(one, acc[0], rtol=0, atol=1e-14 * np.abs(acc[0]).max())


def test_accumulate_and_get_accel_against_the_pyexp_literal_oracle(halo_basis, oracle):
    """``Spherical::accumulate`` / ``computeAccel`` restated literally (oracle/bfe_oracle.c:
    orc_pyexp_sph_accumulate / orc_pyexp_sph_accel; expui/BiorthBasis.cc:583-665, :818-926): particles
    outside [rmin, rmax] are skipped, dsmall 1e-20 / 1e-18, tables evaluated at r/scale inside rmin and
    beyond rmax alike (no exterior continuation), the azimuthal term over the unguarded x^2 + y^2."""
    basis, _ = halo_basis
    rng = np.random.default_rng(11)
    pos = rng.normal(0, 0.35, (3000, 3))
    pos[:40] *= 12.0                                   # beyond rmax = 1.95: skipped by accumulate
    pos[40:50] *= 1e-5                                 # inside rmin = 1e-4: skipped too
    m = rng.uniform(0.5, 1.5, 3000) / 3000
    coefs = basis.createFromArray(m, pos)
    prm = oracle.params(scale=1.0, rmin=basis.rmin, rmax=basis.rmax)
    ref, used = oracle.pyexp_sph_accumulate(basis.grid, prm, pos, m)
    r = np.linalg.norm(pos, axis=1)
    assert used == basis.used == int(((r >= basis.rmin) & (r <= basis.rmax)).sum()) < 3000
    assert np.abs(basis.expcoef - ref).max() <= 1e-10 * np.abs(ref).max()
    basis.set_coefs(coefs)
    # evaluation points: bulk, inside rmin, beyond rmax, a hair off the polar axis
    test = np.concatenate([rng.normal(0, 0.5, (300, 3)),
                           rng.normal(0, 3e-5, (20, 3)),
                           rng.normal(0, 4.0, (40, 3)),
                           np.array([[4e-4, 0.0, 0.4], [0.0, -9e-4, -0.9], [1e-3, 1e-3, 1.2]])])   # theta ~ 1e-3
    a_ref = oracle.pyexp_sph_accel(basis.grid, prm, ref, test)
    acc = basis.getAccel(test)
    rt = np.linalg.norm(test, axis=1)
    assert (rt < basis.rmin).sum() >= 5 and (rt > basis.rmax).sum() >= 10
    scale = np.linalg.norm(a_ref, axis=1)
    assert np.all(np.isfinite(a_ref))
    assert (np.linalg.norm(acc - a_ref, axis=1) <= 1e-9 * np.maximum(scale, scale.max() * 1e-6)).all()
    # A hair off the axis (theta ~ 1e-7) the transverse components are ill-conditioned in the reference
    # itself: sqrt((1-x)(1+x)) with x = z/r rounded amplifies one ulp of x by 1/(1-|x|) ~ 1e13, so even
    # its own 1e-18 versus the n-body 1e-16 added to r moves them by 1e-3.  z is well conditioned.
    hair = np.array([[1e-7, 0.0, 0.4], [0.0, -3e-8, -0.9]])
    lit, got = oracle.pyexp_sph_accel(basis.grid, prm, ref, hair), basis.getAccel(hair)
    assert np.abs(got[:, 2] - lit[:, 2]).max() <= 1e-9 * np.abs(lit[:, 2]).max()
    assert np.abs(got[:, :2] - lit[:, :2]).max() <= 2e-2 * np.abs(lit[:, :2]).max()
    # ON the axis the reference computes potp * y / (x^2 + y^2) = 0/0: NaN in x and y.  The device
    # guards the azimuthal term and returns the finite limit; z agrees.
    axis = np.array([[0.0, 0.0, 0.7], [0.0, 0.0, -0.3]])
    lit = oracle.pyexp_sph_accel(basis.grid, prm, ref, axis)
    assert np.isnan(lit[:, :2]).all() and np.isfinite(lit[:, 2]).all()
    got = basis.getAccel(axis)
    near = basis.getAccel(axis + np.array([1e-9, 0.0, 0.0]))
    assert np.all(np.isfinite(got)) and np.abs(got[:, 2] - lit[:, 2]).max() <= 1e-9 * np.abs(lit[:, 2]).max()
    assert np.abs(got - near).max() <= 1e-6 * np.abs(near).max()


def test_m0_only_is_an_evaluation_flag_in_pyexp(halo_basis, oracle, tmp_path):
    """``Spherical::accumulate`` applies none of the flags (expui/BiorthBasis.cc:583-665): with M0_ONLY the coefficient
    structure it returns still holds every m and only the evaluation drops m > 0 (:851).  The n-body force skips the
    m > 0 sums in the accumulation itself (src/SphericalBasis.cc:550): ``exp_amd_sph_set_accumulate_all_m`` selects
    between the two, and the basis object asks for pyEXP's (found by tests/fuzz/fuzz_pyexp.py)."""
    from exp_amd.basis import Basis
    from exp_amd.runtime import Component, SphereSL
    basis, cfg = halo_basis
    cfg0 = cfg.replace("  Lmax: 2", "  Lmax: 2\n  M0_ONLY: true")
    cfg0 = cfg0.replace(cfg0.split("cachename: ")[1].split()[0], str(tmp_path / "SLGridSph.cache.m0"))
    b0 = Basis.factory(cfg0)
    rng = np.random.default_rng(17)
    pos = rng.normal(0, 0.3, (2000, 3)) * np.array([1.0, 0.7, 0.4])
    m = np.full(2000, 1.0 / 2000)
    c0, call = b0.createFromArray(m, pos), basis.createFromArray(m, pos)
    assert np.abs(c0.coefs[4]).max() > 1e-3 * np.abs(c0.coefs[0]).max()          # (l, m) = (2, 1): not dropped
    assert np.abs(c0.coefs - call.coefs).max() <= 1e-13 * np.abs(call.coefs).max()
    prm = oracle.params(scale=1.0, rmin=b0.rmin, rmax=b0.rmax, M0_only=True)
    ref, _ = oracle.pyexp_sph_accumulate(b0.grid, prm, pos, m)
    assert np.abs(b0.expcoef - ref).max() <= 1e-10 * np.abs(ref).max()
    b0.set_coefs(c0)
    test = rng.normal(0, 0.4, (100, 3))
    a_ref = oracle.pyexp_sph_accel(b0.grid, prm, ref, test)
    assert np.abs(b0.getAccel(test) - a_ref).max() <= 1e-9 * np.linalg.norm(a_ref, axis=1).max()
    basis.set_coefs(call)
    assert np.abs(basis.getAccel(test) - a_ref).max() > 1e-3 * np.linalg.norm(a_ref, axis=1).max()   # the flag matters
    # the n-body force of the same flag leaves the m > 0 rows at zero
    f = SphereSL(b0.ctx, b0.grid, rmin=b0.rmin, rmax=b0.rmax, M0_only=True)
    c = Component.from_arrays(b0.ctx, m, pos)
    f.determine_coefficients(c)
    nb = f.get_coefs()
    assert np.all(nb[[2, 3, 5, 6, 7, 8]] == 0.0) and np.abs(nb[4]).max() > 0
    assert f.lib.exp_amd_sph_set_accumulate_all_m(f.h, 1) == 0
    f.determine_coefficients(c)
    assert np.abs(f.get_coefs()[5]).max() > 0
    c.close(); f.close()


def test_radial_window_keys_N1_N2(halo_basis, oracle, tmp_path):
    """N1 / N2 restrict the l >= 1 sums of computeAccel / sph_eval to n in [N1, N2] while the monopole
    keeps every n (expui/BiorthBasis.cc:841-849 vs :876, :894).  The reference reads both keys with
    ``.as<bool>()`` (:264-265): ``true`` means 1, anything that is not a boolean does not convert."""
    from exp_amd.basis import Basis
    basis, cfg = halo_basis
    cfgw = cfg.replace("  Lmax: 2", "  Lmax: 2\n  N1: true\n  N2: true")
    cfgw = cfgw.replace(cfgw.split("cachename: ")[1].split()[0], str(tmp_path / "SLGridSph.cache.win"))
    win = Basis.factory(cfgw)
    assert (win.N1, win.N2) == (1, 1)
    rng = np.random.default_rng(13)
    pos = rng.normal(0, 0.3, (2000, 3)) * np.array([1.0, 0.8, 0.5])
    m = np.full(2000, 1.0 / 2000)
    coefs = win.createFromArray(m, pos)
    full_c = basis.createFromArray(m, pos).coefs                       # accumulation is not windowed
    assert np.abs(coefs.coefs - full_c).max() <= 1e-13 * np.abs(full_c).max()
    win.set_coefs(coefs)
    test = rng.normal(0, 0.4, (200, 3))
    prm = oracle.params(scale=1.0, rmin=win.rmin, rmax=win.rmax, N1=1, N2=1)
    ref, _ = oracle.pyexp_sph_accumulate(win.grid, prm, pos, m)
    a_ref = oracle.pyexp_sph_accel(win.grid, prm, ref, test)
    acc = win.getAccel(test)
    assert np.abs(acc - a_ref).max() <= 1e-9 * np.linalg.norm(a_ref, axis=1).max()
    full = oracle.pyexp_sph_accel(win.grid, oracle.params(scale=1.0, rmin=win.rmin, rmax=win.rmax), ref, test)
    assert np.abs(full - a_ref).max() > 1e-3 * np.linalg.norm(a_ref, axis=1).max()   # the window matters
    # the same window in the field evaluation
    f_ref = oracle.sph_fields(win.grid, prm, ref, test[:, 0], test[:, 1], test[:, 2], "cartesian")
    f_got = win.getFields(test[:, 0], test[:, 1], test[:, 2])
    assert np.abs(f_got[:, 3:] - f_ref[:, 3:]).max() <= 1e-9 * np.abs(f_ref[:, 3:]).max()
    with pytest.raises(RuntimeError):
        Basis.factory(cfgw.replace("N1: true", "N1: 5"))


def test_get_fields_matches_oracle(halo_basis, oracle):
    """getFields / __call__ in the three coordinate systems (Spherical::crt_eval, cyl_eval, sph_eval,
    expui/BiorthBasis.cc:711-958) against the oracle's restatement; includes points beyond rmax
    (pyEXP extrapolates the tables linearly there) and the labels of BiorthBasis.cc:71-97."""
    basis, _ = halo_basis
    rng = np.random.default_rng(8)
    pos = rng.normal(0, 0.3, (3000, 3))
    pos[:, 2] *= 0.5
    m = np.full(3000, 1.0 / 3000)
    basis.set_coefs(basis.createFromArray(m, pos))
    prm = oracle.params(scale=1.0, rmin=basis.rmin, rmax=basis.rmax)
    coef = basis.force.get_coefs()
    test = rng.normal(0, 0.5, (800, 3))
    test[:5] *= 8.0                                     # a few points outside rmax
    x, y, z = test.T
    R, ph, r = np.hypot(x, y), np.arctan2(y, x), np.linalg.norm(test, axis=1)
    for ctype, args in (("cartesian", (x, y, z)), ("cylindrical", (R, z, ph)),
                        ("spherical", (r, z / r, ph))):
        got = basis(*args, ctype)
        ref = oracle.sph_fields(basis.grid, prm, coef, *args, ctype)
        scale = np.abs(ref).max(axis=0)
        assert np.abs(got - ref).max(axis=0).max() <= 1e-9 * scale.max()
        assert np.all(np.abs(got - ref).max(axis=0) <= 1e-8 * scale + 1e-300)
    f1 = basis.getFields(x[7], y[7], z[7])
    assert f1.shape == (9,) and np.allclose(f1, basis.getFields(x, y, z)[7], rtol=0, atol=1e-13)
    vals, labels = basis.evaluate(x[7], y[7], z[7])
    assert labels == ["dens m=0", "dens m>0", "dens", "potl m=0", "potl m>0", "potl",
                      "x force", "y force", "z force"]
    assert basis.getFieldLabels("spherical")[6:] == ["rad force", "mer force", "azi force"]
    # inside rmax the Cartesian force is the acceleration getAccel returns
    ins = r < 0.9 * basis.rmax
    acc = basis.getAccel(test[ins])
    assert np.abs(basis.getFields(x, y, z)[ins, 6:9] - acc).max() <= 1e-9 * np.abs(acc).max()


def test_cylinder_basis(tmp_path):
    """tests/Disk/cyl_basis.py shape (smaller fiducial orders so that it builds in seconds)."""
    from exp_amd.basis import Basis
    cfg = f"""
---
id: cylinder
parameters:
  acyl: 0.01
  hcyl: 0.001
  mmax: 4
  nmax: 6
  ncylnx: 48
  ncylny: 24
  ncylr: 600
  lmaxfid: 16
  nmaxfid: 12
  rnum: 60
  tnum: 30
  cachename: {tmp_path / 'eof.cache.run0'}
...
"""
    basis = Basis.factory(cfg)
    info = basis.cacheInfo()
    assert info["mmax"] == 4 and info["nmax"] == 6 and info["numx"] == 48
    from exp_amd.models import sample_disk
    m, pos, _ = sample_disk(5000, 3, a=0.01, h=0.001)
    coefs = basis.createFromArray(m, pos, time=0.5)
    assert coefs.coefs.shape == (5, 6) and coefs.time == 0.5
    acc = basis.getAccel(np.array([[0.02, 0.0, 0.0], [0.0, 0.03, 0.001]]))
    assert acc[0, 0] < 0 and acc[1, 1] < 0 and np.all(np.isfinite(acc))
    # Cylindrical::accumulate / computeAccel restated literally (oracle/cyl_oracle.c:
    # orc_pyexp_cyl_accumulate / orc_pyexp_cyl_accel; expui/BiorthBasis.cc:1804-1857): the table
    # window only, accumulated_eval projected on x, y, z -- no taper, nothing beyond the table
    from tests.oracle_lib import Oracle
    orc = Oracle()
    cc, ss, ongrid = orc.pyexp_cyl_accumulate(basis.grid, pos, m)
    assert np.abs(coefs.coefs.real - cc).max() <= 1e-10 * np.abs(cc).max()
    assert np.abs(coefs.coefs.imag - ss).max() <= 1e-10 * np.abs(cc).max()
    rng = np.random.default_rng(8)
    Rt = basis.grid.rtable * basis.grid.ascale
    test = np.concatenate([rng.normal(0, 0.03, (300, 3)) * np.array([1.0, 1.0, 0.1]),
                           np.array([[1.2 * Rt, 0.0, 0.0], [0.0, 0.9 * Rt, 0.3 * Rt], [1e-8, 0.0, 0.001],
                                     [0.7 * Rt, 0.1 * Rt, 0.0]])])
    a_ref = orc.pyexp_cyl_accel(basis.grid, cc, ss, test)
    a_got = basis.getAccel(test)
    assert np.all(a_ref[-4] == 0.0)                       # beyond the table: accumulated_eval returns zeros
    assert np.abs(a_got - a_ref).max() <= 1e-9 * np.linalg.norm(a_ref, axis=1).max()
    # field evaluation: cylindrical coordinates by default (expui/BiorthBasis.cc:1744-1746)
    basis.set_coefs(coefs)
    assert basis.getFieldLabels()[6:] == ["rad force", "ver force", "azi force"]
    fld = basis.getFields(0.02, 0.0, 0.0)
    assert fld.shape == (9,) and fld[2] > 0 and fld[5] < 0 and fld[6] == pytest.approx(acc[0, 0], rel=1e-9)
    cyl = basis(0.02, 0.0, 0.0, "cylindrical")
    assert cyl[6] == pytest.approx(acc[0, 0], rel=1e-9) and cyl[:6] == pytest.approx(fld[:6], rel=1e-12)
    # sub-sample covariance through the basis object (enableCoefCovariance / getCoefCovariance)
    basis.enableCoefCovariance(True, 5)
    basis.createFromArray(m, pos, time=0.6)
    counts, masses = basis.getCovarSamples()
    assert counts.sum() <= 5000 and counts.min() > 0 and masses.sum() == pytest.approx(m.sum(), rel=1e-3)
    VC, MV = basis.getCoefCovariance()
    assert VC.shape == (5, 5, 6) and MV.shape == (5, 5, 6, 6)
    assert np.all(np.real(np.einsum("tmnn->tmn", MV)) >= 0.0)


def test_coefficient_stream_playback_and_fields_over_time(halo_basis, tmp_path):
    """dump_coefs (native stream) of a short run -> SphCoefs.readNativeCoefs -> playback through
    set_coefs gives back the run's own forces; getFieldsCoefs evaluates a point over time
    (src/SphericalBasis.cc:612-680, :1829-1879; expui/BasisFactory.cc:236-265)."""
    from exp_amd.coefs import SphCoefs, complex_to_real_rows
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component
    basis, _ = halo_basis
    f = basis.force
    m, pos, vel = sample_sphere(basis.model, 20000, seed=3, rlim=1.9)
    c = Component.from_arrays(basis.ctx, m, pos, vel)
    path = str(tmp_path / "outcoef.halo.run0")
    dt, accs, sets = 0.002, {}, {}
    f.determine_coefficients(c); c.zero_acceleration(); f.get_acceleration_and_potential(c)
    with open(path, "wb") as out:
        for k in range(4):
            f.step_kdk(c, dt)
            t = round((k + 1) * dt, 8)
            f.dump_coefs(out, time=t)
            sets[t] = f.get_coefs()
            accs[t] = c.download(("acc", "pos"))
    coefs = SphCoefs.readNativeCoefs(path)
    assert coefs.Times() == sorted(sets)
    for t in coefs.Times():
        assert np.array_equal(complex_to_real_rows(coefs.getCoefStruct(t).coefs, basis.lmax), sets[t])
    # playback: the stored set at time t reproduces that step's accelerations on the same positions
    t = coefs.Times()[2]
    basis.set_coefs(coefs.getCoefStruct(t))
    ins = np.linalg.norm(accs[t]["pos"], axis=1) < 0.95 * basis.rmax
    got = basis.getAccel(accs[t]["pos"][ins])
    assert np.abs(got - accs[t]["acc"][ins]).max() <= 1e-9 * np.abs(accs[t]["acc"]).max()
    # interpolation in time at a stored time is that set; fields of one point over all times
    arr, ok = coefs.interpolate(t)
    assert ok and np.allclose(arr, coefs.getCoefStruct(t).coefs, rtol=0, atol=1e-15 * np.abs(arr).max())
    basis.setFieldType("cartesian")
    ret, times = basis.getFieldsCoefs(0.1, -0.05, 0.02, coefs)
    assert list(times) == coefs.Times() and set(ret) == set(basis.getFieldLabels("cartesian"))
    basis.set_coefs(coefs.getCoefStruct(times[1]))
    assert ret["potl"][1] == pytest.approx(basis.getFields(0.1, -0.05, 0.02)[5], rel=1e-13)
    c.close()


def test_nbody_playback_reproduces_the_live_run(halo_basis, tmp_path):
    """`playback` key of the n-body force (src/SphericalBasis.cc:155-213, :600-680, :1676-1754):
    a live run writes its coefficient stream (native and HDF5); re-running the same initial
    conditions with the force in playback mode -- coefficients from the file, no accumulation --
    gives the same trajectory to round-off (interpolation weights 1 and 0 at the stored times; the
    coefficient get/set round trip through the internal row scaling costs an ulp).
    With coefCompute the particle-derived set is also produced and survives the force call."""
    from exp_amd.coefs import SphCoefs, round_time
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, do_step_single
    basis, _ = halo_basis
    f = basis.force
    m, pos, vel = sample_sphere(basis.model, 20000, seed=11, rlim=1.9)
    dt, nstep = 0.002, 4

    def run(playback):
        c = Component.from_arrays(basis.ctx, m, pos, vel)
        f.play_back = False
        f.determine_coefficients(c); c.zero_acceleration(); f.get_acceleration_and_potential(c)
        if playback is not None:
            f.set_playback(playback[0], dt, coef_compute=playback[1])
        rec, sets = SphCoefs("halo"), []
        for k in range(nstep):
            do_step_single(f, c, dt, tnow=round_time((k + 1) * dt))
            sets.append(f.get_coefs())
            if playback is None:
                import io
                from exp_amd.coefs import read_native_record
                buf = io.BytesIO(); f.dump_coefs(buf, time=round_time((k + 1) * dt)); buf.seek(0)
                rec.add(read_native_record(buf))
        out = c.download(("pos", "vel", "acc"))
        c.close()
        f.play_back = False
        return out, rec, sets

    live, rec, live_sets = run(None)
    native, h5 = str(tmp_path / "outcoef.halo"), str(tmp_path / "outcoef.halo.h5")
    rec.writeNativeCoefs(native)
    rec.setUnits([("length", "kpc", 1.0), ("mass", "Msun", 1.0e10), ("time", "Gyr", 1.0), ("G", "mixed", 43007.1)])        # (four units or the write is refused, expui/Coefficients.cc:152-160)
    rec.WriteH5Coefs(h5)
    for src in (rec, native, h5):
        got, _, _ = run((src, False))
        for k in ("pos", "vel", "acc"):
            assert np.abs(got[k] - live[k]).max() <= 1e-12 * np.abs(live[k]).max(), (type(src), k)
    assert f.stop_signal == 0
    got, _, sets = run((rec, True))                      # coefCompute: both sets exist
    assert np.abs(got["pos"] - live["pos"]).max() <= 1e-12 * np.abs(live["pos"]).max()
    # the particle-derived set is restored after the force
    assert np.abs(sets[-1] - live_sets[-1]).max() <= 1e-11 * np.abs(live_sets[-1]).max()
    # a basis mismatch is refused the way the constructor refuses it
    bad = SphCoefs("bad")
    st = rec.getCoefStruct(rec.Times()[0])
    import copy
    st2 = copy.copy(st); st2.nmax = st.nmax + 1
    bad.add(st2)
    with pytest.raises(RuntimeError, match="nmax for playback"):
        f.set_playback(bad, dt)
    # beyond the stored range (by more than 2 dtime) the stop signal is raised
    f.set_playback(rec, dt)
    c = Component.from_arrays(basis.ctx, m, pos, vel)
    f.determine_coefficients(c, tnow=nstep * dt + 5 * dt)
    assert f.stop_signal == 1
    f.play_back = False
    c.close()


def test_coefficient_covariance_by_subsampling(halo_basis, oracle, tmp_path, monkeypatch):
    """pyEXP's pcavar / subsamp (tests/Halo/createCoefs.py asks for them): sub-sample counts, masses,
    mean vectors and covariance matrices of Spherical::accumulate (expui/BiorthBasis.cc:583-665)
    against the oracle's restatement, over two addFromArray batches (the sub-sample index follows
    the running count of accepted particles in the caller's order); the means add up to the
    coefficients; reset_coefs zeroes everything."""
    basis, _ = halo_basis
    rng = np.random.default_rng(21)
    n, sampT = 6000, 9
    pos = rng.normal(0, 0.4, (n, 3))
    pos[::50] *= 20.0                                   # some outside rmax: not counted, not ranked
    m = rng.uniform(0.5, 1.5, n) / n
    basis.enableCoefCovariance(True, sampT)
    basis.initFromArray()
    basis.addFromArray(m[:2500], pos[:2500])
    basis.addFromArray(m[2500:], pos[2500:])
    coef = basis.makeFromArray(time=1.0)
    prm = oracle.params(scale=1.0, rmin=basis.rmin, rmax=basis.rmax)
    ref = oracle.pyexp_sph_covariance(basis.grid, prm, pos, m, sampT)
    counts, masses = basis.getCovarSamples()
    assert np.array_equal(counts, ref["counts"]) and counts.sum() == basis.used == ref["used"]
    assert np.allclose(masses, ref["masses"], rtol=1e-13, atol=0)
    cov = basis.getCoefCovariance()
    assert len(cov) == sampT and len(cov[0]) == (basis.lmax + 1) * (basis.lmax + 2) // 2
    mean = np.array([[cov[t][lm][0] for lm in range(len(cov[0]))] for t in range(sampT)])
    covr = np.array([[cov[t][lm][1] for lm in range(len(cov[0]))] for t in range(sampT)])
    assert np.abs(mean - ref["mean"]).max() <= 1e-10 * np.abs(ref["mean"]).max()
    assert np.abs(covr.real - ref["covr"]).max() <= 1e-10 * np.abs(ref["covr"]).max() and not covr.imag.any()
    assert np.abs(mean.sum(axis=0) - coef.coefs).max() <= 1e-10 * np.abs(coef.coefs).max()
    # the HDF5 covariance store (expui/BiorthBasis.H:433-463 -> expui/Covariance.cc): written in the
    # working directory as coefcovar.<compname>.<runtag>.h5, extended by the next call, read back
    from exp_amd import h5cache
    if h5cache.available():
        monkeypatch.chdir(tmp_path)
        fname = basis.writeCoefCovariance("halo", "run0", 1.0)
        assert fname == "coefcovar.halo.run0.h5" and basis.writeCoefCovariance("halo", "run0", 2.0) == fname
        rd = h5cache.SubsampleCovariance(fname)
        assert rd.BasisID == "SphereSL" and rd.Times() == [1.0, 2.0] and rd.summed
        c, mm, mu, cv = rd.getCoefCovariance(2.0)
        assert np.array_equal(c, counts) and np.array_equal(mm, masses) and np.array_equal(mu, mean)
        assert np.abs(cv[0] - covr.sum(0) / sampT).max() <= 1e-13 * np.abs(covr).max()     # summed, split T ways
    basis.reset_coefs()
    counts, masses = basis.getCovarSamples()
    assert not counts.any() and not masses.any() and not np.array(basis.getCoefCovariance()[0][0][1]).any()
    basis.enableCoefCovariance(False)
    with pytest.raises(RuntimeError, match="covariance not enabled"):
        basis.force.cov_get()


def test_reference_script_sequences_through_the_pyexp_namespace(tmp_path, monkeypatch):
    """The call sequences of the reference's pyEXP scripts, through ``import exp_amd.pyEXP as pyEXP``:
    tests/Halo/createCoefs.py (pcavar / subsamp basis, enableCoefCovariance, SphCoefs(True), four
    createFromArray layouts into the container, writeCoefCovariance, CovarianceReader),
    tests/Halo/readCoefs.py (Coefs.factory on a native stream, getAllCoefs, getName) and
    tests/Halo/changeCoefs.py (invI over the packing index, zeroing the odd orders with setMatrix, the
    zero test on getAllCoefs)."""
    import exp_amd.pyEXP as pyEXP
    monkeypatch.chdir(tmp_path)
    config = HALO_CFG.format(model=os.path.join(GOLD, "SLGridSph.model"), cache="SLGridSph.cache.run0")
    config = config.replace("  rmapping : 0.0667", "  rmapping : 0.0667\n  pcavar : true\n  subsamp : 10")
    basis = pyEXP.basis.Basis.factory(config)
    basis.enableCoefCovariance(True, 100)
    coefs = pyEXP.coefs.SphCoefs(True)
    random.seed(11)
    mass, xpos, ypos, zpos = [], [], [], []
    for _ in range(100):
        mass.append(0.01)
        xpos.append(random.random() * 2.0 - 1.0)
        ypos.append(random.random() * 2.0 - 1.0)
        zpos.append(random.random() * 2.0 - 1.0)
    coef1 = basis.createFromArray(mass, [xpos, ypos, zpos], time=3.0)
    coefs.add(coef1)
    basis.writeCoefCovariance("halo", "test_covar", coef1.time)
    testcovar = pyEXP.basis.CovarianceReader("coefcovar.halo.test_covar.h5")
    counts, masses, means, covr = testcovar.getCoefCovariance(coef1.time)
    assert testcovar.Times() == [3.0] and int(np.sum(counts)) == basis.used and np.sum(masses) == pytest.approx(basis.used * 0.01)
    assert means.shape[1:] == (6, 10) and covr.shape[1:] == (6, 10, 10)
    mass = np.array(mass)
    data = np.array([xpos, ypos, zpos])
    coefs.add(basis.createFromArray(mass, data, time=3.1))
    rng = np.random.default_rng(5)
    mass = np.ones(100) * 1.0e-02
    xyz = [rng.normal(0.0, 1.0, 100) for _ in range(3)]
    coefs.add(basis.createFromArray(mass, xyz, time=3.2))
    coefs.add(basis.createFromArray(mass, np.array(xyz), time=3.3))
    assert coefs.Times() == [3.0, 3.1, 3.2, 3.3]
    # readCoefs.py: a native stream as the n-body code leaves it, through the factory
    coefs.setName("halo")
    coefs.writeNativeCoefs("outcoef.halo.run0")
    back = pyEXP.coefs.Coefs.factory("outcoef.halo.run0")
    data = back.getAllCoefs()
    assert data.shape == (6, 10, 4) and back.getGeometry() == "sphere" and isinstance(back.getName(), str)
    assert coefs.CompareStanzas(back)
    # changeCoefs.py: zero every odd (l, m) through setMatrix, then look again
    times = back.Times()
    for k in range(data.shape[0]):
        l, m, _ = basis.invI(k)
        assert basis.I(l, m) == k
        if l % 2 != 0 or m % 2 != 0:
            data[k, :, :] *= 0.0
    for i in range(data.shape[2]):
        back.setMatrix(times[i], data[:, :, i])
    data1 = back.getAllCoefs()
    odd = [k for k in range(data1.shape[0]) if basis.invI(k)[0] % 2 or basis.invI(k)[1] % 2]
    assert len(odd) == 3 and np.abs(data1[odd]).max() == 0.0 and np.abs(data1).max() > 0.0
    with pytest.raises(RuntimeError, match="requested time"):
        back.setMatrix(9.0, data[:, :, 0])
    # the HDF5 pair and the per-harmonic power
    back.setUnits([("length", "kpc", 1.0), ("mass", "Msun", 1.0e10), ("time", "Gyr", 1.0), ("G", "mixed", 43007.1)])
    back.WriteH5Coefs("halo.h5")
    more = pyEXP.coefs.SphCoefs()
    more.add(basis.createFromArray(mass, xyz, time=3.4))
    more._force_id = "sphereSL"
    more.ExtendH5Coefs("halo.h5")
    allc = pyEXP.coefs.Coefs.factory("halo.h5")
    assert allc.Times() == [3.0, 3.1, 3.2, 3.3, 3.4] and allc.Power().shape == (5, 3)
    assert np.allclose(allc.Power()[0], [np.sum(np.abs(data1[0, :, 0]) ** 2), 0.0,
                                         np.sum(np.abs(data1[[3, 5], :, 0]) ** 2)])
    bad = pyEXP.coefs.SphCoefs()
    bad.add(pyEXP.coefs.SphStruct(3, 10, 1.0, 3.5, np.zeros((10, 10), complex), np.zeros(3), np.eye(3)))
    with pytest.raises(RuntimeError, match="parameter check failed"):
        bad.ExtendH5Coefs("halo.h5")


def test_dump_coefs_h5_creates_then_extends(tmp_path):
    """``SphericalBasis::dump_coefs_h5`` / ``Cylinder::dump_coefs_h5`` as ``OutCoef`` calls them (src/OutCoef.cc:135,
    src/SphericalBasis.cc:1909-1975, src/Cylinder.cc:1625-1690): the first call creates the file with the component's
    name, the configuration and the default units, later calls append snapshots; what comes back are the
    coefficient sets the force method held at each call."""
    from exp_amd import h5cache
    if not h5cache.available():
        pytest.skip("HDF5 shim not built")
    from exp_amd.coefs import Coefs, complex_to_real_rows
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Context, Cylinder, SphereSL
    from tests.conftest import make_grid
    from tests.test_cyl_gpu import _disk, cyl_grid
    ctx = Context(0)
    model, g = make_grid("plummer", 3, 6, 300)
    m, pos, vel = sample_sphere(model, 20000, seed=11)
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m, pos, vel)
    path = str(tmp_path / "outcoef.halo.h5")
    sets = []
    for k in range(3):
        f.determine_coefficients(c)
        sets.append(f.get_coefs().copy())
        f.dump_coefs_h5(path, time=0.01 * k, name="dark halo", config="id: sphereSL", center=[0.1, 0.0, -0.2])
        c.incr_position(0.05)
    back = Coefs.factory(path)
    assert back.getName() == "dark halo" and back.Times() == [0.0, 0.01, 0.02] and back.getGeometry() == "sphere"
    assert back.getUnits() == [("G", "none", 1.0), ("length", "none", 1.0), ("mass", "none", 1.0), ("time", "none", 1.0)]
    for k, t in enumerate(back.Times()):
        st = back.getCoefStruct(t)
        assert np.array_equal(complex_to_real_rows(st.coefs, 3), sets[k]) and np.array_equal(st.ctr, [0.1, 0.0, -0.2])
    c.close(); f.close()
    gc = cyl_grid(3, 5)
    md, pd, _ = _disk(20000, 7, gc)
    fc = Cylinder(ctx, gc)
    cd = Component.from_arrays(ctx, md, pd)
    pathc = str(tmp_path / "outcoef.disk.h5")
    for k in range(2):
        fc.determine_coefficients(cd)
        fc.dump_coefs_h5(pathc, time=0.5 * k, name="star disk")
    cc, ss = fc.get_coefs()
    backc = Coefs.factory(pathc)
    assert backc.getGeometry() == "cylinder" and backc.Times() == [0.0, 0.5] and len(backc.getUnits()) == 4
    st = backc.getCoefStruct(0.5)
    assert np.array_equal(st.coefs.real, cc) and np.array_equal(st.coefs.imag[1:], ss[1:]) and np.all(st.coefs.imag[0] == 0)
    cd.close(); fc.close(); ctx.close()


def test_reference_disk_script_configuration(tmp_path, capsys):
    """tests/Disk/cyl_basis.py: its parameter set as it stands (the deprecated ``eof_file`` and ``density`` keys, ``ncylodd: 3``,
    ``pnum: 0``, ``ashift``, ``vflag``, ``logr``, ``ignore``) builds a basis and ``cacheInfo(<file>)`` reads the cache back --
    five vertically symmetric and three antisymmetric functions per m are recorded in it."""
    from exp_amd.basis import Basis
    cache = tmp_path / ".eof.cache.run0t"
    cfg = f"""
---
id: cylinder
parameters:
  acyl: 0.01
  hcyl: 0.001
  lmaxfid: 20
  nmaxfid: 20
  mmax: 6
  nmax: 8
  ncylnx: 128
  ncylny: 64
  ncylodd: 3
  rnum: 32
  pnum: 0
  tnum: 16
  ashift: 0.5
  vflag: 16
  logr: false
  density: true
  eof_file: {cache}
  ignore: true
...
"""
    disk_basis = Basis.factory(cfg)
    out = capsys.readouterr().out
    assert "'eof_file' is deprecated" in out and "'density' is deprecated" in out
    node_cyl = disk_basis.cacheInfo(str(cache))
    assert (node_cyl["mmax"], node_cyl["nmax"], node_cyl["numx"], node_cyl["numy"]) == (6, 8, 128, 64)
    assert (node_cyl["lmaxfid"], node_cyl["nmaxfid"], node_cyl["neven"], node_cyl["nodd"]) == (20, 20, 5, 3)
    oc = disk_basis.orthoCheck()
    assert max(np.abs(np.asarray(m) - np.eye(8)).max() for m in oc) < 0.05
    # functions 5..7 change sign across the plane: a particle pair mirrored in z excites none of them
    pos = np.array([[0.02, 0.01, 0.0007], [0.02, 0.01, -0.0007]])
    cs = disk_basis.createFromArray(np.array([0.5, 0.5]), pos)
    c = np.abs(cs.coefs)
    assert c[:, 5:].max() <= 1e-12 * c[:, :5].max()
    # ... and a single particle above the plane does excite them
    one = np.abs(disk_basis.createFromArray(np.array([1.0]), pos[:1]).coefs)
    assert one[:, 5:].max() > 1e-3 * one[:, :5].max()


def test_cylindrical_conditioning_options(tmp_path):
    """The keys that shape the conditioning of the disk basis (expui/BiorthBasis.cc:1265-1366, :1397-1440): ``pnum`` knots with a
    shifted target (``ashift``) give sine functions of their own -- the projection then fetches both table sets --, ``sech2``
    and ``dtype: doubleexpon`` change the target; every variant stays biorthogonal and its coefficients and accelerations
    match the literal pyEXP twin of the oracle on the tables it produced."""
    from exp_amd.basis import Basis
    from exp_amd.models import sample_disk
    from tests.oracle_lib import Oracle
    orc = Oracle()
    base = """
id: cylinder
parameters:
  acyl: 0.01
  hcyl: 0.001
  mmax: 3
  nmax: 6
  ncylodd: 2
  ncylnx: 40
  ncylny: 20
  ncylr: 500
  lmaxfid: 14
  nmaxfid: 10
  rnum: 40
  tnum: 24
"""
    m, pos, _ = sample_disk(4000, 5, a=0.01, h=0.001)
    pos[:, 0] += 0.004                                        # an off-centre disk: odd m are excited
    test = np.random.default_rng(3).normal(0, 0.02, (200, 3)) * np.array([1.0, 1.0, 0.1])
    tabs = {}
    for tag, extra in (("plain", ""), ("shift", "  pnum: 6\n  ashift: 0.6\n"), ("sech2", "  sech2: true\n"),
                       ("double", "  dtype: doubleexpon\n  aratio: 2.5\n  hratio: 0.5\n  dweight: 0.7\n")):
        basis = Basis.factory(base + extra + f"  cachename: {tmp_path / ('eof.' + tag)}\n")
        tabs[tag] = basis.grid.tab.copy()
        oc = basis.orthoCheck()
        assert max(np.abs(np.asarray(q) - np.eye(6)).max() for q in oc) < 0.2, tag        # (a 40 x 20 grid: coarse)
        cs = basis.createFromArray(m, pos)
        cc, ss, _ = orc.pyexp_cyl_accumulate(basis.grid, pos, m)
        assert np.abs(cs.coefs.real - cc).max() <= 1e-10 * np.abs(cc).max(), tag
        assert np.abs(cs.coefs.imag - ss).max() <= 1e-10 * np.abs(cc).max(), tag
        a_ref = orc.pyexp_cyl_accel(basis.grid, cc, ss, test)
        assert np.abs(basis.getAccel(test) - a_ref).max() <= 1e-9 * np.linalg.norm(a_ref, axis=1).max(), tag
        assert basis.force.lib is not None
    t = tabs["plain"]
    assert np.array_equal(t[3:, 1:], t[:3, 1:])
    s = tabs["shift"]
    assert np.abs(s[3:, 1:] - s[:3, 1:]).max() > 0.02 * np.abs(s[:3]).max()
    for tag in ("sech2", "double"):
        assert np.abs(tabs[tag][:3] - t[:3]).max() > 1e-3 * np.abs(t[:3]).max(), tag
    with pytest.raises(RuntimeError, match="invalid DiskType"):
        Basis.factory(base + f"  dtype: toomre\n  cachename: {tmp_path / 'eof.bad'}\n")
