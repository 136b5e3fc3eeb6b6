"""Basis orders above the unrolled kernels' range: lmax > 12 (SPH_MAX_L), mmax > 12 (CYL_MAX_M), covariance nmax > 64 / 32.
The reference takes whatever the YAML gives (src/Sphere.cc:28-96, src/Cylinder.cc:473); here those configurations run on
the any-order kernels (exp_amd/csrc/sph_gen.hip, k_cyl_moments_gen / k_cyl_force_gen in cyl.hip) -- slower, never
refused.  Bars as everywhere: coefficients 1e-10 of the largest, accelerations / potentials 1e-9 (BASELINE north_star),
counts exact; and at an order both kernel families cover, EXP_AMD_SPH_GENERIC / EXP_AMD_CYL_GENERIC = 1 must reproduce
the unrolled kernels' results to rounding."""
import numpy as np
import pytest

from tests.conftest import make_grid

pytestmark = pytest.mark.gpu
COEF_TOL, ACC_TOL = 1e-10, 1e-9


@pytest.fixture(scope="module")
def ctx():
    from exp_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def _halo(model, n, seed):
    from exp_amd.models import sample_sphere
    m, pos, vel = sample_sphere(model, n, seed=seed)
    pos[:, 2] *= 0.7
    pos[:, 0] += 0.05
    pos[:7] *= 80.0 / np.linalg.norm(pos[:7], axis=1)[:, None]        # beyond rmax: exterior continuation
    pos[7] = [0.0, 0.0, 0.3]                                           # on the polar axis
    return m, pos, vel


@pytest.mark.parametrize("kind,lmax,nmax,flags", [("plummer", 16, 6, {}), ("plummer", 13, 10, {}), ("nfw", 20, 4, {}),
                                                  ("plummer", 14, 5, dict(EVEN_M=True, NO_L1=True)),
                                                  ("plummer_log", 13, 4, {})])
def test_sphere_above_the_unrolled_range(ctx, oracle, kind, lmax, nmax, flags):
    from exp_amd.runtime import Component, SphereSL
    model, g = make_grid(kind, lmax, nmax, 300)
    m, pos, _ = _halo(model, 3000, 7 + lmax)
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax, **flags)
    c_ref, used_ref = oracle.sph_accumulate(g, prm, pos, m)
    a_ref, p_ref = oracle.sph_accel(g, prm, pos, c_ref)
    f = SphereSL(ctx, g, **flags)
    c = Component.from_arrays(ctx, m, pos)
    f.determine_coefficients(c)
    coef = f.get_coefs()
    assert f.Used() == used_ref
    assert np.abs(coef - c_ref).max() <= COEF_TOL * np.abs(c_ref).max()
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot"))
    na = np.linalg.norm(out["acc"] - a_ref, axis=1) / (np.linalg.norm(a_ref, axis=1) + 1e-300)
    assert na.max() <= ACC_TOL
    assert np.abs(out["pot"] - p_ref).max() <= ACC_TOL * np.abs(p_ref).max()
    # a fused KDK step at this order against the oracle's step
    vel = np.zeros_like(pos)
    c.upload(m, pos, vel)
    f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
    f.step_kdk(c, 0.01)
    p1, v1, a1, pt1, cf1 = oracle.sph_step(g, prm, 0.01, pos, vel, a_ref, m)
    st = c.download()
    assert np.abs(st["pos"] - p1).max() <= 1e-11
    assert np.abs(f.get_coefs() - cf1).max() <= COEF_TOL * np.abs(cf1).max()
    assert (np.linalg.norm(st["acc"] - a1, axis=1) / (np.linalg.norm(a1, axis=1) + 1e-300)).max() <= ACC_TOL
    c.close(); f.close()


def test_generic_sphere_kernels_reproduce_the_unrolled_ones(ctx, monkeypatch):
    """lmax 6: the same particles through both kernel families -- same moments, same projected table."""
    from exp_amd.runtime import Component, SphereSL
    model, g = make_grid("plummer", 6, 18, 800)
    m, pos, _ = _halo(model, 20000, 3)
    res = []
    for gen in ("0", "1"):
        monkeypatch.setenv("EXP_AMD_SPH_GENERIC", gen)
        f = SphereSL(ctx, g)
        c = Component.from_arrays(ctx, m, pos)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
        res.append((f.get_coefs().copy(), f.Used(), c.download(("acc", "pot"))))
        c.close(); f.close()
    (c0, u0, o0), (c1, u1, o1) = res
    assert u0 == u1
    assert np.abs(c0 - c1).max() <= 1e-13 * np.abs(c0).max()
    assert (np.linalg.norm(o0["acc"] - o1["acc"], axis=1) / (np.linalg.norm(o0["acc"], axis=1) + 1e-300)).max() <= 1e-11
    assert np.abs(o0["pot"] - o1["pot"]).max() <= 1e-12 * np.abs(o0["pot"]).max()


_CYL = {}


def _cyl_grid(mmax, norder):
    from exp_amd.empcyl import build_empcyl
    if (mmax, norder) not in _CYL:
        _CYL[(mmax, norder)] = build_empcyl(mmax=mmax, norder=norder, numx=32, numy=16, lmaxfid=max(20, mmax + 4), nmaxfid=12,
                                            numr=400, rnum=60, tnum=30, nodd=1)
    return _CYL[(mmax, norder)]


def _disk(n, seed, g):
    from exp_amd.models import sample_disk
    m, pos, vel = sample_disk(n, seed, a=g.ascale, h=g.hscale)
    pos[:, 0] *= 1.15
    pos[:, 1] += 0.1 * g.ascale
    return m, pos, vel


@pytest.mark.parametrize("mmax,norder", [(14, 5), (13, 3)])
def test_cylinder_above_the_unrolled_range(ctx, oracle, mmax, norder):
    from exp_amd.runtime import Component, Cylinder
    g = _cyl_grid(mmax, norder)
    m, pos, _ = _disk(6000, 50 + mmax, g)
    c_ref, s_ref, used_ref, mass_ref = oracle.cyl_accumulate(g, pos, m)
    a_ref, p_ref = oracle.cyl_accel(g, pos, c_ref, s_ref, mass_ref)
    f = Cylinder(ctx, g)
    c = Component.from_arrays(ctx, m, pos)
    f.determine_coefficients(c)
    cc, ss = f.get_coefs()
    assert f.Used() == used_ref
    assert f.cylmass == pytest.approx(mass_ref, rel=1e-12)
    scale = np.abs(c_ref).max()
    assert np.abs(cc - c_ref).max() <= COEF_TOL * scale and np.abs(ss - s_ref).max() <= COEF_TOL * scale
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot"))
    assert np.abs(out["acc"] - a_ref).max() <= ACC_TOL * np.linalg.norm(a_ref, axis=1).max()
    assert np.abs(out["pot"] - p_ref).max() <= ACC_TOL * np.abs(p_ref).max()
    c.close(); f.close()


def test_generic_cylinder_kernels_reproduce_the_unrolled_ones(ctx, monkeypatch):
    from exp_amd.runtime import Component, Cylinder
    g = _cyl_grid(4, 6)
    m, pos, vel = _disk(20000, 9, g)
    res = []
    for gen in ("0", "1"):
        monkeypatch.setenv("EXP_AMD_CYL_GENERIC", gen)
        f = Cylinder(ctx, g)
        c = Component.from_arrays(ctx, m, pos, vel)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
        f.step_kdk(c, 1e-4)
        cc, ss = f.get_coefs()
        res.append((cc.copy(), ss.copy(), f.Used(), c.download()))
        c.close(); f.close()
    (a0, b0, u0, o0), (a1, b1, u1, o1) = res
    assert u0 == u1
    assert np.abs(a0 - a1).max() <= 1e-13 * np.abs(a0).max() and np.abs(b0 - b1).max() <= 1e-13 * np.abs(a0).max()
    assert np.abs(o0["acc"] - o1["acc"]).max() <= 1e-11 * np.linalg.norm(o0["acc"], axis=1).max()
    assert np.abs(o0["pos"] - o1["pos"]).max() <= 1e-13


@pytest.mark.parametrize("lmax,nmax,mmax,norder", [(13, 4, 13, 3), (10, 24, 6, 12)])
def test_block_multistep_above_the_unrolled_range(ctx, oracle, lmax, nmax, mmax, norder):
    """The C++ step driver with both bases on the any-order kernels (lmax 13, mmax 13), two components, cross forces,
    level changes: against the n-body oracle, bars of tests/test_config4_gpu.py.  Second case: the headline's orders
    (lmax 10, nmax 24; mmax 6, nmax 12) -- unrolled kernels, but coefficient sets and projected rows large enough that
    the thin kernels' tiles need more than 64 KB of LDS."""
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    from tests import config4_util as c4
    from tests.oracle_lib import NBodyOracle
    model, g = make_grid("plummer", lmax, nmax, 300)
    cg = _cyl_grid(mmax, norder)
    inp = c4.config4_inputs(n_halo=700, n_disk=500)
    ms, dtime, dyn = 3, c4.DTIME, list(c4.DYN)
    sc = float(inp["scale"])
    prm = oracle.params(**c4.sph_window(g, sc))
    nb = NBodyOracle(oracle, ms, dtime, dyn)
    i0 = nb.add_sphere(g, prm, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
    i1 = nb.add_cylinder(cg, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
    nb.add_interaction(i0, i1); nb.add_interaction(i1, i0)
    sim = Simulation(ctx, dtime, multistep=ms, dynfrac=dyn, shiftlevl=0)
    f0 = SphereSL(ctx, g, multistep=ms, **c4.sph_window(g, sc))
    f1 = Cylinder(ctx, cg, multistep=ms)
    c0 = Component.from_arrays(ctx, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
    c1 = Component.from_arrays(ctx, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
    j0, j1 = sim.add_component(c0, f0), sim.add_component(c1, f1)
    sim.add_interaction(j0, j1); sim.add_interaction(j1, j0)
    nb.init(); sim.init()
    for k in range(3):
        if k:
            nsw = nb.step(); sim.step(1)
            assert sim.step_switches == sum(nsw)
        for j, (f, c) in enumerate(((f0, c0), (f1, c1))):
            s = nb.state[j]
            assert int((c.download_levels() != s["level"]).sum()) <= 2          # (a dtreq within rounding of a power of two)
            out = c.download()
            a = np.stack([s["a" + q] for q in "xyz"], 1)
            assert np.abs(out["pos"] - np.stack([s[q] for q in "xyz"], 1)).max() <= 1e-11
            assert np.abs(out["acc"] - a).max() <= ACC_TOL * np.linalg.norm(a, axis=1).max()
            cmax = np.abs(s["coefN"]).max()
            for M in range(ms + 1):
                gn = f.get_coefs(level=M)
                gn = gn.reshape(-1) if j == 0 else np.concatenate([x.reshape(-1) for x in gn])
                assert np.abs(gn - s["coefN"][M]).max() <= COEF_TOL * cmax
    sim.close(); c0.close(); c1.close(); f0.close(); f1.close()


def test_sphere_covariance_with_more_than_64_radial_functions(ctx, oracle):
    """sub-sample covariance (expui/BiorthBasis.cc:583-665) at nmax 70: counts exact, means / covariances 1e-10
    (the contraction kernel used to keep one thread's share of an nmax <= 64 matrix in registers)."""
    from exp_amd.runtime import Component, SphereSL
    model, g = make_grid("plummer", 1, 70, 300)
    rng = np.random.default_rng(4)
    n, sampT = 3000, 5
    pos = rng.normal(0, 0.5, (n, 3))
    pos[::60] *= 300.0
    m = rng.uniform(0.5, 1.5, n) / n
    f = SphereSL(ctx, g)
    f.cov_enable(sampT)
    c = Component.from_arrays(ctx, m, pos)
    used = f.cov_accumulate(c)
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    ref = oracle.pyexp_sph_covariance(g, prm, pos, m, sampT)
    got = f.cov_get()
    assert used == ref["used"] and np.array_equal(got["counts"], ref["counts"])
    assert np.abs(got["mean"] - ref["mean"]).max() <= 1e-10 * np.abs(ref["mean"]).max()
    assert np.abs(got["covr"] - ref["covr"]).max() <= 1e-10 * np.abs(ref["covr"]).max()
    c.close(); f.close()


def test_cylinder_covariance_with_more_than_32_radial_functions(ctx, oracle):
    """... and the cylinder's (exputil/EmpCylSL.cc:4049-4146, :4974-5015) at nmax 40, on tables extended with synthetic
    higher orders (the kernels and the oracle only consume tables), sine tables different from the cosine ones."""
    import copy
    from exp_amd.runtime import Component, Cylinder
    g0 = _cyl_grid(4, 6)
    rng = np.random.default_rng(8)
    N = 40
    g = copy.copy(g0)
    g.mmax, g.norder = 2, N
    reps = -(-N // g0.norder)
    tab = np.tile(g0.tab[:, :3], (1, 1, reps, 1, 1))[:, :, :N].copy()
    tab *= 1.0 + 0.3 * rng.standard_normal((6, 3, N, 1, 1))
    g.tab = np.ascontiguousarray(tab)
    g.dens = None
    n, sampT = 4000, 3
    m, pos, _ = _disk(n, 12, g)
    pos[::45] *= 400.0
    f = Cylinder(ctx, g)
    f.cov_enable(sampT)
    c = Component.from_arrays(ctx, m, pos)
    used = f.cov_accumulate(c)
    ref = oracle.cyl_covariance(g, pos, m, sampT)
    got = f.cov_get()
    assert used == ref["used"] and np.array_equal(got["counts"], ref["counts"])
    assert np.abs(got["mean"] - ref["mean"]).max() <= 1e-10 * np.abs(ref["mean"]).max()
    assert np.abs(got["covr"] - ref["covr"]).max() <= 1e-10 * np.abs(ref["covr"]).max()
    assert np.abs(ref["covr"].imag).max() > 0.0
    c.close(); f.close()
