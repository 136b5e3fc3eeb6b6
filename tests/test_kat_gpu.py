"""Known answers for the DEVICE path that do not go through the oracle (GPU only).

The parity suites compare the HIP path with `oracle/`; the reference holds no golden vectors for
coefficients or accelerations, so these tests hold the device path directly
to answers neither implementation produced:

* the classical multipole solution of a density made of single solid harmonics l = 1..6
  (tests/kat_multipole.py): pins factorial(l,m), the sqrt(2), the Condon-Shortley sign, -4 pi and the
  row order of src/SphericalBasis.cc:328-335, :519-566, :1555-1625 on the GPU;
* Poisson's equation in integral form for every (l, n) through pyEXP's getBasis tables;
* at the HEADLINE size (1e8 particles, lmax 10, nmax 24, numr 2000: BASELINE config 5): linearity of the
  accumulation, invariance to particle order, the used count, and Newton's theorem."""
import math
import os

import numpy as np
import pytest

from tests.conftest import make_grid
from tests.test_oracle_kat import KAT_MODES, check_multipole_errors, multipole_errors, poisson_residuals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from exp_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def _device_accel(ctx, f):
    from exp_amd.runtime import Component

    def fn(pts, coef):
        f.set_coefs(coef)
        c = Component.from_arrays(ctx, np.ones(len(pts)), pts)
        c.zero_acceleration(0)
        f.get_acceleration_and_potential(c, external=True)
        out = c.download(("acc", "pot"))
        c.close()
        return out["acc"], out["pot"]
    return fn


def test_multipole_known_answers_on_the_device(ctx):
    """tests/test_oracle_kat.py::test_multipole_known_answers_for_every_l with the HIP kernels in place of
    the oracle: k_sph_accumulate / k_sph_contract for the coefficients, the force passes (general pass
    with its r > rmax branch) for the field of each row.  Same bars."""
    from exp_amd.runtime import Component, SphereSL
    from tests.kat_multipole import MultipoleCase, row_of
    model, g = make_grid("plummer", 6, 12, 800)
    case = MultipoleCase(model, g, KAT_MODES)
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, case.mass, case.pos)
    f.determine_coefficients(c)
    coef = f.get_coefs()
    assert f.Used() == len(case.mass)
    rows = [row_of(*md) for md in KAT_MODES]
    quiet = np.delete(coef, rows, axis=0)
    assert np.abs(quiet).max() < 1e-12 * np.abs(coef).max()
    check_multipole_errors(multipole_errors(case, coef, _device_accel(ctx, f)), KAT_MODES)
    # ... and the self field on the sorted store (fast pass for cell-uniform waves) of all modes at
    # once equals the sum of the exact modes at a sample of the quadrature points
    f.set_coefs(coef)
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("pot", "acc"))
    rr = np.linalg.norm(case.pos, axis=1)
    cand = np.flatnonzero((rr > 0.2) & (rr < 3.0))
    sub = cand[np.random.default_rng(2).choice(len(cand), 40, replace=False)]
    epot, eacc = np.zeros(len(sub)), np.zeros((len(sub), 3))
    for k in range(len(KAT_MODES)):
        p, a = case.exact_mode(k, case.pos[sub])
        epot += p
        eacc += a
    assert np.abs(out["pot"][sub] - epot).max() < 2e-4 * np.abs(epot).max()
    assert np.abs(out["acc"][sub] - eacc).max() < 1e-3 * np.linalg.norm(eacc, axis=1).max()
    c.close()
    f.close()


def test_poisson_consistency_through_getbasis(ctx, tmp_path):
    """pyEXP's SphericalSL.getBasis (expui/BiorthBasis.cc:960-993) tabulates potential, density and radial
    force of every (l, n): on the device tables they satisfy the radial Poisson equation in integral
    form (tests/test_oracle_kat.py::poisson_residuals)."""
    from exp_amd.basis import Basis
    from exp_amd.models import PlummerModel
    model = PlummerModel(1.0, 1.0, 1e-3, 50.0)
    mfile = tmp_path / "plummer.model"
    model.write_table(str(mfile), 1000)
    basis = Basis.factory(f"""
id: sphereSL
parameters:
  Lmax: 4
  nmax: 8
  numr: 800
  rmapping: 1.0
  modelname: {mfile}
  cachename: {tmp_path / 'kat_cache'}
""", ctx)

    for r1, r2 in ((0.05, 0.4), (0.3, 2.0), (1.0, 10.0)):
        num = 1500
        t = basis.getBasis(math.log10(r1), math.log10(r2), num)
        rr = 10.0 ** np.linspace(math.log10(r1), math.log10(r2), num)
        P = np.array([[t[l][n]["potential"] for n in range(8)] for l in range(5)])     # [l, n, r]
        D = np.array([[t[l][n]["density"] for n in range(8)] for l in range(5)])
        F = np.array([[t[l][n]["rforce"] for n in range(8)] for l in range(5)])
        worst = 0.0
        for l in range(5):
            ip = np.trapezoid(P[l], rr, axis=1)
            lhs = -(r2 * r2 * F[l][:, -1] - r1 * r1 * F[l][:, 0]) - l * (l + 1) * ip      # rforce = -dphi/dr
            rhs = np.trapezoid(D[l] * rr * rr, rr, axis=1)
            scale = np.abs(r2 * r2 * F[l][:, -1]) + np.abs(r1 * r1 * F[l][:, 0]) + l * (l + 1) * np.abs(ip)
            worst = max(worst, float((np.abs(lhs - rhs) / scale).max()))
        assert worst < 8e-3, (r1, r2, worst)


def test_headline_size_properties(ctx):
    """BASELINE config 5 at its full size on one GPU: 1e8 truncated-NFW particles, SphericalSL lmax 10,
    nmax 24, numr 2000 (19 GB of the 288).  Particles are drawn in HBM (as bench.py does) and never
    visit the host.  Size-independent properties, no oracle:
      * linearity: the coefficients of two disjoint parts add up to those of the whole, and so do
        their used counts;
      * invariance to particle order (a device-side permutation);
      * Newton's theorem: a spherical sample of the basis' own model feels the model's force,
        median |a_r| r^2 / M(<r) = 1 (sampling noise at 1e8: ~3e-5; bar 1e-3);
      * every particle inside [rmin, rmax] is used."""
    import torch
    from bench import make_halo
    from exp_amd.models import NFWModel
    from exp_amd.runtime import Component, SphereSL
    from exp_amd.slgrid import build_slgrid
    model = NFWModel(1.0, 20.0, 6.0, 1e-3, 50.0)
    g = build_slgrid(model, 10, 24, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
    n = 100_000_000
    dev = torch.device("cuda:0")
    x, y, z, _, _, _ = make_halo(model, n, 4242, dev)
    r = torch.sqrt(x * x + y * y + z * z)
    inside = int(((r >= g.rmin) & (r <= g.rmax)).sum().item())
    mass = torch.full((n,), 1.0 / n, device=dev, dtype=torch.float64)
    f = SphereSL(ctx, g)

    def coefs(sl, scale_z=1.0, perm=None):
        xs, ys, zs, ms = x[sl], y[sl], (z[sl] * scale_z if scale_z != 1.0 else z[sl]), mass[sl]
        if perm is not None:
            xs, ys, zs = xs[perm], ys[perm], zs[perm]
        c = Component(ctx, xs.numel())
        c.upload_device(ms.contiguous(), xs.contiguous(), ys.contiguous(), zs.contiguous())
        f.determine_coefficients(c)
        out, used = f.get_coefs(), f.Used()
        return c, out, used

    # flattened copy so that every (l, m) row is exercised
    c_all, cf_all, u_all = coefs(slice(0, n), 0.8)
    c_all.close()
    cut = 37_000_000
    ca, cf_a, u_a = coefs(slice(0, cut), 0.8)
    ca.close()
    cb, cf_b, u_b = coefs(slice(cut, n), 0.8)
    cb.close()
    big = np.abs(cf_all).max()
    assert u_a + u_b == u_all
    assert np.abs(cf_a + cf_b - cf_all).max() <= 1e-10 * big
    perm = torch.randperm(cut, device=dev)
    cp, cf_p, u_p = coefs(slice(0, cut), 0.8, perm)
    cp.close()
    del perm
    assert u_p == u_a
    assert np.abs(cf_p - cf_a).max() <= 1e-10 * np.abs(cf_a).max()
    # Newton on the spherical sample
    c, cf, used = coefs(slice(0, n))
    assert used == inside
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    # every 1000th particle back to the host is enough for the median (and for the caller order)
    out = c.download(("acc", "pos"))
    sub = slice(0, n, 997)
    pos = out["pos"][sub]
    assert np.array_equal(pos[:, 0], x[sub].cpu().numpy())
    rr = np.linalg.norm(pos, axis=1)
    sel = (rr > 0.5) & (rr < 5.0)
    arad = -(out["acc"][sub][sel] * pos[sel]).sum(1) / rr[sel]
    ratio = arad * rr[sel] ** 2 / model.mass(rr[sel])
    assert abs(np.median(ratio) - 1.0) < 1e-3
    # the monopole row carries the model: c_{00,n>0} small against c_{00,0}
    assert np.abs(cf[0, 1:]).max() < 2e-3 * abs(cf[0, 0])
    c.close()
    f.close()


def test_headline_step_forms_agree_at_full_size(ctx):
    """BASELINE config 5 at its full size on one GPU, STEPPED: five `exp_amd_step_kdk` of 1e8 truncated-NFW particles
    (SphericalSL lmax 10, nmax 24, numr 2000) in the three forms of the fused step -- ordinary (key histogram, scan, scatter
    every step), APPEND (the default at this size: the force pass places every particle in the next step's cell order) and
    append with the lean payload -- must leave the same system behind.  Size-independent properties, no oracle, nothing
    but twenty numbers per run visits the host:
      * every particle is still there, once: the used count of the last accumulation and the total mass (a particle lost or
        placed twice moves the sum of 1e8 equal masses by 1e-8: bar 1e-12) agree with the ordinary run's;
      * the coefficient sets agree to 1e-11 of their largest entry, the centres of mass and velocity to 1e-12;
      * the sums of `OutLog` over the STATE the forms leave -- kinetic and potential energy, the Clausius virial, angular
        momentum (positions, velocities, accelerations and potentials of every particle: for the lean payload the
        re-evaluated ones) -- agree to 1e-10;
      * the append runs really ran without sort passes (per-kernel launch counts)."""
    import torch
    from bench import make_halo
    from exp_amd.models import NFWModel
    from exp_amd.runtime import Component, SphereSL
    from exp_amd.slgrid import build_slgrid
    if os.environ.get("EXP_AMD_APPEND_MIN") or os.environ.get("EXP_AMD_APPEND_LEAN"):
        pytest.skip("the forms are chosen by the environment")
    model = NFWModel(1.0, 20.0, 6.0, 1e-3, 50.0)
    g = build_slgrid(model, 10, 24, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
    n = 100_000_000
    dev = torch.device("cuda:0")
    x, y, z, vx, vy, vz = make_halo(model, n, 23456, dev)
    mass = torch.full((n,), 1.0 / n, device=dev, dtype=torch.float64)

    def run(append_min, lean):
        ctx.set_append_min(append_min)
        ctx.set_append_lean(lean)
        f = SphereSL(ctx, g)
        c = Component(ctx, n)
        c.upload_device(mass, x, y, z, vx, vy, vz)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
        sorts = []
        for _ in range(5):
            ctx.profile(True); ctx.profile_reset()
            f.step_kdk(c, 0.002)
            sorts.append(bool(ctx.profile_report().get("k_scatter_adv", {}).get("launches", 0)))
            ctx.profile(False)
        out = {"coef": f.get_coefs(), "used": f.Used(), "fix": c.fix_positions(0), "log": c.log_sums(), "sorts": sorts}
        c.close(); f.close()
        ctx.set_append_min(0); ctx.set_append_lean(False)
        return out

    ref = run(0, False)
    assert all(ref["sorts"]) and ref["log"]["nbodies"] == n
    big = np.abs(ref["coef"]).max()
    for form in (run(1 << 20, False), run(1 << 20, True)):
        assert form["sorts"] == [True, True, False, False, False]       # entry at the second step, then no sort passes
        assert form["used"] == ref["used"] and form["log"]["nbodies"] == n
        assert form["fix"]["mtot"] == pytest.approx(ref["fix"]["mtot"], rel=1e-12)
        assert np.abs(form["coef"] - ref["coef"]).max() <= 1e-11 * big
        for k in ("com", "cov"):
            assert np.abs(form["fix"][k] - ref["fix"][k]).max() <= 1e-12
        assert np.abs(form["fix"]["coa"] - ref["fix"]["coa"]).max() <= 1e-10 * max(1.0, np.abs(ref["log"]["eptot"]))
        for k in ("ektot", "eptot", "clausius"):
            assert form["log"][k] == pytest.approx(ref["log"][k], rel=1e-10), k
        assert np.abs(form["log"]["angm"] - ref["log"]["angm"]).max() <= 1e-10 * max(np.abs(ref["log"]["angm"]).max(), 1e-6)


def test_cylinder_poisson_consistency_through_getbasis(ctx, tmp_path):
    """pyEXP's Cylindrical.getBasis (expui/BiorthBasis.cc:1930-1974 -> EmpCylSL::get_all) on the device tables:
    potential, density, radial and vertical force of every (m, n) satisfy Poisson's equation in integral form
    (tests/test_oracle_kat.py::cyl_poisson_residual) -- the device's bilinear blend of the four table kinds, on a
    grid that is not the tables' own."""
    from exp_amd.basis import Basis
    from tests.test_oracle_kat import cyl_poisson_residual
    basis = Basis.factory(f"""
id: cylinder
parameters:
  acyl: 0.01
  hcyl: 0.001
  mmax: 2
  nmax: 5
  ncylnx: 96
  ncylny: 48
  ncylr: 1000
  lmaxfid: 24
  nmaxfid: 20
  rnum: 100
  tnum: 40
  cachename: {tmp_path / 'eof.cache.kat'}
""", ctx)
    for (R1, R2, z1, z2) in ((0.004, 0.03, -0.002, 0.002), (0.01, 0.05, -0.004, 0.001)):
        nR, nZ = 160, 120
        t = basis.getBasis(R1, R2, nR, z1, z2, nZ, True)
        R, z = np.linspace(R1, R2, nR), np.linspace(z1, z2, nZ)
        for m in range(3):
            for n in range(5):
                b = t[m][n]
                res = cyl_poisson_residual(R, z, b["potential"], b["rforce"], b["zforce"], b["density"], m)
                assert res < 2e-2, (R1, R2, m, n, res)
