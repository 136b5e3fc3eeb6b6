"""The keys of the two force methods that default to "off", against the oracle (SURVEY.md section 8, rows a6 / a9 / a10 /
a13 / a15 -- the option branches inside the cited line ranges):

* ``rtrunc`` / ``com0`` -- ``Component::freeze`` (src/Component.cc:4194-4202; call sites src/SphericalBasis.cc:468, :1159,
  :1521; src/Cylinder.cc:842, :1329, :1756);
* ``ton`` / ``toff`` / ``twid`` -- ``Component::Adiabatic`` (src/Component.cc:4214-4220; :441, :471, :1161 / :834, :1758);
* ``self_consistent: false`` (src/SphericalBasis.cc:694, :1682; src/Cylinder.cc:959, :1469, :1755);
* ``FIX_L0`` (src/SphericalBasis.cc:1689-1694);
* ``mlim`` (src/Cylinder.cc:225; exputil/EmpCylSL.cc:5317, :5465, :5602).

Bars as everywhere: coefficients 1e-10 of the largest, accelerations / potentials 1e-9.  GPU only."""
import os

import numpy as np
import pytest

from tests import config4_util as c4
from tests.golden_util import load_cyl
from tests.oracle_lib import NBodyOracle

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
    m = m * (1.0 + 0.3 * np.sin(np.arange(n)))            # unequal masses: the mass stream is read
    return m, pos, vel


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("lmax,nmax,n", [(4, 8, 6000), (6, 18, 20000), (10, 24, 5000)])
def test_sphere_freeze_and_adiabatic_against_the_oracle(ctx, oracle, lmax, nmax, n):
    """accumulation and self force with a truncation radius that freezes ~a third of the particles about a shifted
    com0 + centre, and an adiabatic factor"""
    from exp_amd.runtime import Component, SphereSL
    from tests.conftest import make_grid
    model, g = make_grid("plummer", lmax, nmax, 400 if lmax != 6 else 800)
    m, pos, _ = _halo(model, n, 11 + lmax)
    center, com0, rtrunc, adb = np.array([0.02, -0.01, 0.03]), np.array([0.1, 0.05, -0.02]), 1.1, 0.3721
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    with oracle.call_opts(adb=adb, rtrunc=rtrunc, com0=com0, fcenter=center):
        c_ref, used_ref = oracle.sph_accumulate(g, prm, pos, m, center=center)
    with oracle.call_opts(rtrunc=rtrunc, com0=com0, fcenter=center):
        a_ref, p_ref = oracle.sph_accel(g, prm, pos, c_ref, center=center)
    frozen = np.linalg.norm(pos - com0 - center, axis=1) > rtrunc
    assert 0.15 * n < frozen.sum() < 0.7 * n
    assert np.all(a_ref[frozen] == 0.0) and np.all(p_ref[frozen] == 0.0)

    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m, pos)
    c.set_center(center)
    c.set_rtrunc(rtrunc, com0)
    f.set_mass_scale(adb)
    f.determine_coefficients(c)
    assert f.Used() == used_ref
    assert _rel(f.get_coefs(), c_ref) <= COEF_TOL
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot"))
    assert np.all(out["acc"][frozen] == 0.0) and np.all(out["pot"][frozen] == 0.0)
    assert np.abs(out["acc"] - a_ref).max() <= ACC_TOL * np.linalg.norm(a_ref, axis=1).max()
    assert np.abs(out["pot"] - p_ref).max() <= ACC_TOL * np.abs(p_ref).max()
    # ... the same through one fused KDK step: frozen particles are kicked with zero acceleration, i.e. drift freely
    c2 = Component.from_arrays(ctx, m, pos, np.full_like(pos, 0.01))
    c2.set_center(center)
    c2.set_rtrunc(rtrunc, com0)
    f.determine_coefficients(c2)
    c2.zero_acceleration(0)
    f.get_acceleration_and_potential(c2)
    f.step_kdk(c2, 1e-3)
    o2 = c2.download(("pos", "vel", "acc"))
    p1 = pos + 0.01 * 1e-3
    still = frozen & (np.linalg.norm(p1 - com0 - center, axis=1) > rtrunc)
    assert still.sum() > 0.1 * n
    assert np.abs(o2["pos"][still] - p1[still]).max() <= 1e-15 and np.all(o2["vel"][still] == 0.01)
    assert np.all(o2["acc"][still] == 0.0)
    for o in (c, c2, f):
        o.close()


def test_sphere_external_target_is_frozen_by_its_own_rtrunc(ctx, oracle, plummer_small):
    """SetExternal: the TARGET's freeze (cC->freeze, src/SphericalBasis.cc:1521) with the target's own com0 + centre, while
    the positions go into the source's frame"""
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_small
    m, pos, _ = _halo(model, 4000, 5)
    mt, post, _ = _halo(model, 3000, 6)
    post = post * 1.3 + np.array([0.3, 0.0, -0.1])
    src_center, tgt_center, com0, rtrunc = np.array([0.05, 0.0, 0.0]), np.array([0.3, 0.0, -0.1]), np.array([0.0, 0.1, 0.0]), 1.5
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    c_ref, _ = oracle.sph_accumulate(g, prm, pos, m, center=src_center)
    with oracle.call_opts(rtrunc=rtrunc, com0=com0, fcenter=tgt_center):
        a_ref, p_ref = oracle.sph_accel(g, prm, post, c_ref, center=src_center)
    f = SphereSL(ctx, g)
    cs, ct = Component.from_arrays(ctx, m, pos), Component.from_arrays(ctx, mt, post)
    cs.set_center(src_center)
    ct.set_center(tgt_center)
    ct.set_rtrunc(rtrunc, com0)
    f.determine_coefficients(cs)
    ct.zero_acceleration(0)
    f.get_acceleration_and_potential(ct, external=True)
    out = ct.download(("acc", "pot"))
    frozen = np.linalg.norm(post - com0 - tgt_center, axis=1) > rtrunc
    assert 100 < frozen.sum() < 2900 and np.all(out["acc"][frozen] == 0.0)
    assert np.abs(out["acc"] - a_ref).max() <= ACC_TOL * np.linalg.norm(a_ref, axis=1).max()
    assert np.abs(out["pot"] - p_ref).max() <= ACC_TOL * np.abs(p_ref).max()
    for o in (cs, ct, f):
        o.close()


@pytest.mark.parametrize("mlim", [None, 1, 0])
def test_cylinder_freeze_adiabatic_and_mlim_against_the_oracle(ctx, oracle, mlim):
    from exp_amd.models import sample_disk
    from exp_amd.runtime import Component, Cylinder
    cg, _ = load_cyl()
    n = 20000
    m, pos, _ = sample_disk(n, 17, a=cg.ascale, h=cg.hscale, mass=1.0)
    m = m * (1.0 + 0.4 * np.cos(np.arange(n)))
    pos[:, 0] *= 1.2
    pos[::7] *= 40.0                                          # some beyond 0.75 of the table radius: taper + monopole
    center, com0, rtrunc, adb = np.array([1e-3, -5e-4, 2e-4]), np.array([2e-3, 0.0, 1e-3]), 0.025, 0.61
    with oracle.call_opts(adb=adb, rtrunc=rtrunc, com0=com0, fcenter=center, mlim=mlim):
        cc, ss, used_ref, cm_ref = oracle.cyl_accumulate(cg, pos, m, center=center)
        a_ref, p_ref = oracle.cyl_accel(cg, pos, cc, ss, cm_ref, center=center)
    frozen = np.linalg.norm(pos - com0 - center, axis=1) > rtrunc
    assert 0.1 * n < frozen.sum() < 0.8 * n
    f = Cylinder(ctx, cg, mlim=-1 if mlim is None else mlim)
    c = Component.from_arrays(ctx, m, pos)
    c.set_center(center)
    c.set_rtrunc(rtrunc, com0)
    f.set_mass_scale(adb)
    f.determine_coefficients(c)
    gc, gs = f.get_coefs()
    scale = max(np.abs(cc).max(), np.abs(ss).max())
    assert f.Used() == used_ref
    assert f.cylmass == pytest.approx(cm_ref, rel=1e-12)
    assert np.abs(gc - cc).max() <= COEF_TOL * scale and np.abs(gs - ss).max() <= COEF_TOL * scale
    if mlim is not None:
        assert np.all(gc[mlim + 1:] == 0.0) and np.all(gs[mlim + 1:] == 0.0) and np.abs(gc[:mlim + 1]).max() > 0
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot"))
    assert np.all(out["acc"][frozen] == 0.0) and np.all(out["pot"][frozen] == 0.0)
    assert np.abs(out["acc"] - a_ref).max() <= ACC_TOL * np.linalg.norm(a_ref, axis=1).max()
    assert np.abs(out["pot"] - p_ref).max() <= ACC_TOL * np.abs(p_ref).max()
    c.close(); f.close()


def test_frozen_particles_are_left_out_of_the_centre_and_log_sums(ctx):
    """``if (c->freeze(n)) continue;`` also sits in Component::fix_positions (src/Component.cc:3336) and in OutLog's sums
    (src/OutLog.cc:460): total mass, centres of mass / velocity / acceleration, angular momentum, energies and the virial
    are over the particles inside rtrunc of com0 + center.  Against numpy with the reference's test."""
    from exp_amd.runtime import Component
    rng = np.random.default_rng(23)
    n = 40000
    m = rng.uniform(0.5, 1.5, n) / n
    pos, vel, acc = rng.standard_normal((3, n, 3))
    pot = rng.standard_normal(n)
    com0, ctr, rtrunc = np.array([0.2, -0.1, 0.05]), np.array([-0.05, 0.1, 0.0]), 1.3
    c = Component.from_arrays(ctx, m, pos, vel)
    c.upload_acc(acc, pot)
    c.set_center(ctr)
    c.set_rtrunc(rtrunc, com0)
    d = pos - com0 - ctr
    keep = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]) <= rtrunc * rtrunc
    assert 0.2 * n < keep.sum() < 0.8 * n
    mk = m[keep]
    got = c.fix_positions(0)
    mt = mk.sum()
    assert abs(got["mtot"] - mt) <= 1e-13 * mt
    for key, arr in (("com", pos), ("cov", vel), ("coa", acc)):
        ref = (mk[:, None] * arr[keep]).sum(0) / mt
        assert np.abs(got[key] - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-3), key
    lg = c.log_sums()
    assert abs(lg["mtot"] - mt) <= 1e-13 * mt and lg["nbodies"] == n
    p, v, a = pos[keep], vel[keep], acc[keep]
    ref = {"com": (mk[:, None] * p).sum(0), "cov": (mk[:, None] * v).sum(0), "angm": (mk[:, None] * np.cross(p, v)).sum(0)}
    for key in ref:
        assert np.abs(lg[key] - ref[key]).max() <= 1e-12 * max(np.abs(ref[key]).max(), 1e-3), key
    assert abs(lg["ektot"] - (0.5 * mk * (v * v).sum(1)).sum()) <= 1e-12 * lg["ektot"]
    assert abs(lg["eptot"] - (0.5 * mk * pot[keep]).sum()) <= 1e-12 * max(abs(lg["eptot"]), 1e-3)
    assert abs(lg["clausius"] - (mk * (p * a).sum(1)).sum()) <= 1e-12 * max(abs(lg["clausius"]), 1e-3)
    # rtrunc back at its default: everything counts again
    c.set_rtrunc(1.0e20, com0)
    assert abs(c.fix_positions(0)["mtot"] - m.sum()) <= 1e-13 * m.sum()
    c.close()


def test_mlim_refusals(ctx):
    from exp_amd.runtime import Cylinder
    cg, _ = load_cyl()
    f = Cylinder(ctx, cg, mlim=1)
    f.set_mlim(0)                                              # lowering is fine
    with pytest.raises(RuntimeError, match="already dropped"):
        f.set_mlim(1)
    with pytest.raises(RuntimeError, match="must be >= 0"):
        f.set_mlim(-3)
    f.set_mlim(cg.mmax + 5)                                   # min(MLIM, MMAX): nothing to do
    f.close()


def test_pyexp_cylindrical_honours_mlim(oracle, tmp_path):
    """``mlim`` in the YAML of ``Basis.factory`` (expui/BiorthBasis.cc:1466, :1620: `if (mlim>=0) sl->set_mlim(mlim)`) used
    to be accepted and dropped: coefficients and accelerations against the literal pyEXP twins of the oracle with MLIM"""
    from exp_amd.basis import Basis
    from exp_amd.models import sample_disk
    cfg = f"""
---
id: cylinder
parameters:
  acyl: 0.01
  hcyl: 0.001
  mmax: 4
  nmax: 6
  mlim: 2
  ncylnx: 48
  ncylny: 24
  ncylr: 600
  lmaxfid: 16
  nmaxfid: 12
  rnum: 60
  tnum: 30
  cachename: {tmp_path / 'eof.cache.mlim'}
...
"""
    basis = Basis.factory(cfg)
    m, pos, _ = sample_disk(5000, 3, a=0.01, h=0.001)
    pos[:, 0] *= 1.25
    coefs = basis.createFromArray(m, pos, time=0.0)
    with oracle.call_opts(mlim=2):
        cc, ss, _ = oracle.pyexp_cyl_accumulate(basis.grid, pos, m)
        rng = np.random.default_rng(8)
        test = rng.normal(0, 0.03, (300, 3)) * np.array([1.0, 1.0, 0.1])
        a_ref = oracle.pyexp_cyl_accel(basis.grid, cc, ss, test)
    assert np.all(coefs.coefs[3:] == 0.0) and np.abs(coefs.coefs[:3]).max() > 0
    assert np.abs(coefs.coefs.real[:3] - cc[:3]).max() <= 1e-10 * np.abs(cc).max()
    assert np.abs(coefs.coefs.imag[:3] - ss[:3]).max() <= 1e-10 * np.abs(cc).max()
    a_got = basis.getAccel(test)
    assert np.abs(a_got - a_ref).max() <= 1e-9 * np.linalg.norm(a_ref, axis=1).max()
    # ... and it matters: the unrestricted evaluation of the same coefficients differs
    a_all = oracle.pyexp_cyl_accel(basis.grid, *oracle.pyexp_cyl_accumulate(basis.grid, pos, m)[:2], test)
    assert np.abs(a_all - a_ref).max() > 1e-6 * np.linalg.norm(a_ref, axis=1).max()


def _driver_run(ctx, inp, opts_h, opts_d, ms, dtime, nsteps):
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    g, cg = c4.grids()
    sc = float(inp["scale"])
    ch = Component.from_arrays(ctx, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
    cd = Component.from_arrays(ctx, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
    fh = SphereSL(ctx, g, multistep=ms, self_consistent=opts_h.get("self_consistent", True), FIX_L0=opts_h.get("fix_l0", False),
                  **c4.sph_window(g, sc))
    fd = Cylinder(ctx, cg, multistep=ms, self_consistent=opts_d.get("self_consistent", True),
                  mlim=opts_d["mlim"] if opts_d.get("mlim") is not None else -1)
    sim = Simulation(ctx, dtime, multistep=ms, dynfrac=c4.DYN)
    ih, idk = sim.add_component(ch, fh), sim.add_component(cd, fd)
    sim.add_interaction(ih, idk)
    sim.add_interaction(idk, ih)
    for c, o, k in ((ch, opts_h, ih), (cd, opts_d, idk)):
        if o.get("rtrunc") is not None:
            c.set_rtrunc(o["rtrunc"], o.get("com0"))
        if o.get("adiabatic") is not None:
            sim.set_adiabatic(k, *o["adiabatic"])
        if o.get("noise") is not None and k == ih:
            fh.set_noise(NOISE_MODEL, noiseN=o["noise"][0], seedN=o["noise"][1])
        if o.get("freeze_levels") or o.get("noswitch"):
            c.set_level_policy(noswitch=o.get("noswitch", False), freeze_levels=o.get("freeze_levels", False),
                               dtreset=o.get("dtreset", True))
    sim.init()
    sim.step(nsteps)
    return sim, (fh, fd), (ch, cd)


SCENARIOS = [
    (dict(rtrunc=0.35, com0=(0.004, -0.003, 0.002), fix_l0=True), dict(adiabatic=(7.5e-5, 1.0e20, 1.5e-4), mlim=1)),
    (dict(self_consistent=False), dict(rtrunc=0.03, self_consistent=False)),
    (dict(adiabatic=(1.0e-4, 4.0e-4, 1.0e-4), rtrunc=0.6), dict(rtrunc=0.05, com0=(0.001, 0.0, 0.0), adiabatic=(0.0, 1.0e20, 3.0e-4))),
    # "freezeL" (Component::FreezeLev, src/multistep.cc:158, :534): the disk keeps the levels of its first assignment, the halo moves on
    (dict(), dict(freeze_levels=True)),
    (dict(freeze_levels=True, rtrunc=0.6), dict(freeze_levels=True)),
    # "noswitch" / "dtreset" (src/multistep.cc:136-147): dtreq keeps the smallest step asked for, levels move at the end of a master step
    (dict(noswitch=True), dict()),
    (dict(noswitch=True, dtreset=False), dict(noswitch=True)),
    # NOISE (src/SphericalBasis.cc:395, :2150-2210): the halo's self calls consume their draws and keep the combined set
    # (multistep: compute_multistep_coefficients follows), its force on the disk is evaluated from the draws
    (dict(noise=(1.0, 77)), dict()),
]
NOISE_MODEL = os.path.join(os.path.dirname(__file__), "golden", "SLGridSph.model")


def _oracle_opts(oracle, g, scale, o):
    """a scenario's option dict as NBodyOracle.set_options takes it (the NOISE entry needs compute_rms_coefs of the model)"""
    o = dict(o)
    if o.get("noise") is not None:
        from exp_amd.slgrid import read_model_table
        r, d, _, _ = read_model_table(NOISE_MODEL)
        meanC, rmsC = oracle.sph_compute_rms_coefs(g, r, d, scale)
        o["noise"] = (meanC, rmsC, o["noise"][0], o["noise"][1])
    return o


@pytest.mark.parametrize("scen", range(len(SCENARIOS)))
@pytest.mark.parametrize("overlap", ["1", "0"])
def test_step_driver_with_the_option_keys_against_the_nbody_oracle(ctx, oracle, monkeypatch, scen, overlap):
    """the C++ step driver (exp_amd_sim_*: both schedules) on the disk + halo miniature with the keys switched on: levels
    bit for bit, trajectories, accelerations, combined coefficient sets, cylmass against oracle/nbody_oracle.c"""
    monkeypatch.setenv("EXP_AMD_SIM_OVERLAP", overlap)
    oh, od = SCENARIOS[scen]
    ms, dtime, nsteps = 2, 1.5e-4, 2
    inp = c4.config4_inputs(n_halo=500, n_disk=500)
    g, cg = c4.grids()
    prm = oracle.params(**c4.sph_window(g, float(inp["scale"])))
    nb = NBodyOracle(oracle, ms, dtime, c4.DYN)
    i1 = nb.add_sphere(g, prm, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
    i2 = nb.add_cylinder(cg, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
    nb.add_interaction(i1, i2)
    nb.add_interaction(i2, i1)
    nb.set_options(i1, **_oracle_opts(oracle, g, float(inp["scale"]), oh))
    nb.set_options(i2, **od)
    nb.init()
    nsw = [0, 0]
    for _ in range(nsteps):
        nsw = [a + b for a, b in zip(nsw, nb.step())]
    sim, forces, comps = _driver_run(ctx, inp, oh, od, ms, dtime, nsteps)
    assert sum(nsw) > 0 or (oh.get("freeze_levels") and od.get("freeze_levels"))
    for k, (f, c) in enumerate(zip(forces, comps)):
        if (oh, od)[k].get("freeze_levels"):
            assert nsw[k] == 0 and len(np.unique(nb.state[k]["level"])) > 1      # levels assigned once, never moved
        st = nb.state[k]
        o = c.download()
        assert np.array_equal(c.download_levels(), st["level"]), k
        p = np.stack([st[q] for q in "xyz"], 1)
        v = np.stack([st["v" + q] for q in "xyz"], 1)
        a = np.stack([st["a" + q] for q in "xyz"], 1)
        assert np.abs(o["pos"] - p).max() <= 1e-11, k
        assert np.abs(o["vel"] - v).max() <= 1e-9 * np.abs(v).max(), k
        assert np.abs(o["acc"] - a).max() <= 1e-8 * np.linalg.norm(a, axis=1).max(), k
        assert np.abs(o["pot"] - st["pot"]).max() <= 1e-8 * np.abs(st["pot"]).max(), k
        gc = f.get_coefs()
        gc = np.concatenate([np.asarray(x).reshape(-1) for x in (gc if isinstance(gc, tuple) else (gc,))])
        assert np.abs(gc - st["coef"]).max() <= 1e-10 * np.abs(st["coef"]).max(), k
    assert forces[1].cylmass == pytest.approx(nb.cylmass(1), rel=1e-12, abs=1e-300)
    if not oh.get("self_consistent", True):
        assert forces[0].coefs_frozen and forces[1].coefs_frozen
    sim.close()
    for o in list(forces) + list(comps):
        o.close()


def test_self_consistent_false_and_fix_l0_at_the_call_level(ctx, oracle, plummer_small):
    """the PotAccel calls themselves: determine_coefficients returns at once after the first completed call unless
    `initializing`; FIX_L0 saves the monopole row at the first evaluation and restores it at every later one"""
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_small
    m, pos, vel = _halo(model, 5000, 21)
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    f = SphereSL(ctx, g, self_consistent=False, FIX_L0=True)
    c = Component.from_arrays(ctx, m, pos, vel)
    assert not f.coefs_frozen
    f.determine_coefficients(c)                                # the first call always runs (firstime_coef)
    c0 = f.get_coefs()
    c_ref, _ = oracle.sph_accumulate(g, prm, pos, m)
    assert _rel(c0, c_ref) <= COEF_TOL and f.coefs_frozen
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)                        # FIX_L0: C0 saved here
    a0 = c.download(("acc",))["acc"]
    c.incr_position(0.05)
    f.determine_coefficients(c)                                # frozen: nothing happens
    assert np.array_equal(f.get_coefs(), c0)
    f.set_initializing(True)
    f.determine_coefficients(c)                                # begin_run may still re-make them
    f.set_initializing(False)
    moved = c.download(("pos",))["pos"]
    c1_ref, _ = oracle.sph_accumulate(g, prm, moved, m)
    assert _rel(f.get_coefs(), c1_ref) <= COEF_TOL and not np.array_equal(f.get_coefs(), c0)
    # the next force evaluation puts the saved monopole row back (and only that row)
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    now = f.get_coefs()
    assert np.array_equal(now[0], c0[0]) and np.array_equal(now[1:], f.get_coefs()[1:])
    want = c1_ref.copy()
    want[0] = c_ref[0]
    a_ref, _ = oracle.sph_accel(g, prm, moved, want)
    got = c.download(("acc",))["acc"]
    assert np.abs(got - a_ref).max() <= 1e-9 * np.linalg.norm(a_ref, axis=1).max()
    assert np.abs(got - a0).max() > 0
    # a fused step with the coefficients held fixed: kick, drift, force of the same set, kick
    f.step_kdk(c, 1e-3)
    assert np.array_equal(f.get_coefs(), now)
    c.close(); f.close()


def test_level_policy_keys_through_the_c_abi(ctx):
    """The component keys noswitch / freezeL / dtreset (src/Component.cc:253-255): ``freezeL`` makes the per-call
    adjust_multistep_level (exp_amd_force_adjust_multistep_level, what the C++ adaptor calls) a no-op after the first call
    (src/multistep.cc:158: firstCall = this_step == 0 and mdrft == 0) and leaves it alone on that call; with ``noswitch`` the
    calls inside a master step move nothing and the one at mdrft == Mstep does."""
    import ctypes
    from exp_amd.runtime import Component, SphereSL
    from tests.conftest import make_grid
    rng = np.random.default_rng(3)
    n, ms = 20000, 3
    _, g = make_grid("plummer", 4, 8, 400)
    m = np.full(n, 1.0 / n)
    pos = rng.standard_normal((n, 3)) * 0.5
    vel = rng.standard_normal((n, 3)) * 0.3
    f = SphereSL(ctx, g, multistep=ms)
    c = Component.from_arrays(ctx, m, pos, vel)
    c.set_level_policy(freeze_levels=True, dtreset=False)
    f.set_multistep_level(0)
    f.determine_coefficients(c)                  # (the store in this force's order, ms + 1 levels)
    c.upload_acc(rng.standard_normal((n, 3)), -np.abs(rng.standard_normal(n)) - 0.5)
    dyn = (ctypes.c_double * 5)(1e20, 0.02, 1e20, 0.05, 0.05)
    nsw = ctypes.c_longlong(-1)

    def adjust(mdrft, first_step):
        rc = c.lib.exp_amd_force_adjust_multistep_level(f.h, c.h, 0.05, dyn, 0, mdrft, first_step, ctypes.byref(nsw))
        assert rc == 0
        return int(nsw.value)

    assert adjust(0, 1) > 0                      # begin_run's call: the one assignment
    lev = c.download_levels()
    assert len(np.unique(lev)) > 1
    c.incr_velocity(0.3)                          # other velocities: every criterion changes
    assert adjust(1, 1) == 0 and adjust(8, 0) == 0
    assert np.array_equal(c.download_levels(), lev)
    c.set_level_policy(freeze_levels=False)
    assert adjust(8, 0) > 0                      # ... and with the key off the same call moves particles
    # noswitch: mdrft = 4 of Mstep = 8 examines levels >= 1 and moves nothing; mdrft = 8 assigns from the smallest dt seen
    c.set_level_policy(noswitch=True)
    assert adjust(0, 1) >= 0                     # (first call: dtreq reset, levels assigned)
    lev = c.download_levels()
    c.incr_velocity(-0.6)
    assert adjust(4, 0) == 0 and np.array_equal(c.download_levels(), lev)
    assert adjust(8, 0) > 0
    c.close(); f.close()


@pytest.mark.parametrize("ms", [2, 0])
def test_eqmotion_off_moves_nothing_and_matches_the_oracle(ctx, oracle, ms):
    """The global ``eqmotion: false`` (src/global.cc:54; src/incpos.cc:75, src/incvel.cc:93): a step of the driver evaluates
    expansions, forces and (block multistep) level proposals as the time goes on and moves nothing -- positions and velocities
    bit for bit where they started, levels, accelerations and combined coefficient sets against the n-body oracle (the disk's
    adiabatic factor makes the fields change with the time alone)."""
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    dtime, nsteps = 1.5e-4, 2
    inp = c4.config4_inputs(n_halo=400, n_disk=400)
    g, cg = c4.grids()
    sc = float(inp["scale"])
    prm = oracle.params(**c4.sph_window(g, sc))
    nb = NBodyOracle(oracle, ms, dtime, c4.DYN)
    nb.eqmotion = False
    i1 = nb.add_sphere(g, prm, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
    i2 = nb.add_cylinder(cg, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
    nb.add_interaction(i1, i2); nb.add_interaction(i2, i1)
    nb.set_options(i2, adiabatic=(1.0e-4, 1.0e20, 1.0e-4))
    nb.init()
    for _ in range(nsteps):
        nb.step()
    ch = Component.from_arrays(ctx, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
    cd = Component.from_arrays(ctx, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
    fh = SphereSL(ctx, g, multistep=ms, **c4.sph_window(g, sc))
    fd = Cylinder(ctx, cg, multistep=ms)
    sim = Simulation(ctx, dtime, multistep=ms, dynfrac=c4.DYN)
    ih, idk = sim.add_component(ch, fh), sim.add_component(cd, fd)
    sim.add_interaction(ih, idk); sim.add_interaction(idk, ih)
    sim.set_adiabatic(idk, 1.0e-4, 1.0e20, 1.0e-4)
    sim.set_eqmotion(False)
    sim.init()
    a0 = cd.download()["acc"].copy()
    sim.step(nsteps)
    for k, (f, c, pos0, vel0) in enumerate(((fh, ch, inp["halo_pos"], inp["halo_vel"]), (fd, cd, inp["disk_pos"], inp["disk_vel"]))):
        st, o = nb.state[k], c.download()
        assert np.array_equal(o["pos"], pos0) and np.array_equal(o["vel"], vel0), k            # nothing moved
        assert np.array_equal(np.stack([st[q] for q in "xyz"], 1), pos0), k
        assert np.array_equal(c.download_levels(), st["level"]), k
        a = np.stack([st["a" + q] for q in "xyz"], 1)
        assert np.abs(o["acc"] - a).max() <= 1e-8 * np.linalg.norm(a, axis=1).max(), k
        gc = f.get_coefs()
        gc = np.concatenate([np.asarray(x).reshape(-1) for x in (gc if isinstance(gc, tuple) else (gc,))])
        assert np.abs(gc - st["coef"]).max() <= 1e-10 * np.abs(st["coef"]).max(), k
    assert np.abs(cd.download()["acc"] - a0).max() > 1e-3 * np.abs(a0).max()                   # ... while the fields did change
    assert sim.time == pytest.approx(nsteps * dtime)
    for x in (sim, ch, cd, fh, fd):
        x.close()


def test_single_component_single_level_run_turns_on_adiabatically(ctx, oracle):
    """One component, multistep 0: the step driver takes its fused step (exp_amd/csrc/host.hip: exp_amd_sim_step ->
    exp_amd_step_kdk).  Component::Adiabatic is evaluated at EVERY determine_coefficients (src/SphericalBasis.cc:441,
    src/step.cc:273-293), so a component whose turn-on lies ahead of begin_run must come on during the run -- against
    the n-body oracle, and against the same run without the key."""
    from exp_amd.runtime import Component, Simulation, SphereSL
    dtime, nsteps = 1.5e-4, 4
    adb = (3.0e-4, 1.0e20, 1.5e-4)                    # mass factor 2e-5 at t = 0, 0.98 at t = 6e-4
    inp = c4.config4_inputs(n_halo=600, n_disk=10)
    g, _ = c4.grids()
    sc = float(inp["scale"])
    prm = oracle.params(**c4.sph_window(g, sc))
    nb = NBodyOracle(oracle, 0, dtime, c4.DYN)
    i1 = nb.add_sphere(g, prm, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
    nb.set_options(i1, adiabatic=adb)
    nb.init()
    for _ in range(nsteps):
        nb.step()
    outs = []
    for on in (True, False):
        ch = Component.from_arrays(ctx, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
        fh = SphereSL(ctx, g, multistep=0, **c4.sph_window(g, sc))
        sim = Simulation(ctx, dtime, multistep=0, dynfrac=c4.DYN)
        ih = sim.add_component(ch, fh)
        if on:
            sim.set_adiabatic(ih, *adb)
        sim.init()
        sim.step(nsteps)
        outs.append((ch.download(), np.asarray(fh.get_coefs()).reshape(-1).copy()))
        for x in (sim, ch, fh):
            x.close()
    (o, gc), (o_off, gc_off) = outs
    st = nb.state[0]
    a = np.stack([st["a" + q] for q in "xyz"], 1)
    p = np.stack([st[q] for q in "xyz"], 1)
    assert np.abs(o["pos"] - p).max() <= 1e-12 * np.abs(p).max()
    assert np.abs(o["acc"] - a).max() <= 1e-9 * np.linalg.norm(a, axis=1).max()
    assert np.abs(gc - st["coef"]).max() <= 1e-10 * np.abs(st["coef"]).max()
    # the factor at the last accumulation (t = 6e-4): the set is that fraction of the full-mass one, not the 2e-5 of begin_run
    from math import erf
    fac = 0.25 * (1.0 + erf((nsteps * dtime - adb[0]) / adb[2])) * 2.0
    assert 0.9 < fac < 1.0 and abs(gc[0] / gc_off[0] - fac) < 0.02


@pytest.mark.parametrize("ssfrac,nthrds", [(0.5, 1), (0.37, 1), (0.8, 4), (0.3, 3)])
def test_sphere_subset_against_the_oracle(ctx, oracle, ssfrac, nthrds):
    """``ssfrac`` (src/SphericalBasis.cc:149-152, :437-440, :459-460, :472-473): thread id of nthrds walks
    [n id / nthrds, floor(ssfrac * (n (id + 1) / nthrds))) of the level list -- the END index is scaled, so with several
    threads the later slices come out short or empty -- with every mass divided by ssfrac; Component::freeze and the
    adiabatic factor apply inside the loop.  The level list is the caller's order.  Coefficients, the in-window count and
    the forces of ALL particles from the sub-sample's set against the oracle; then the fused step with the key on."""
    from exp_amd.runtime import Component, SphereSL, do_step_single
    inp = c4.config4_inputs(n_halo=3001, n_disk=10)
    g, _ = c4.grids()
    sc = float(inp["scale"])
    win = c4.sph_window(g, sc)
    prm = oracle.params(**win)
    m, pos, vel = inp["halo_mass"], inp["halo_pos"], inp["halo_vel"]
    n = len(m)
    # which particles the reference's loops visit
    sel = np.zeros(n, bool)
    for i in range(nthrds):
        nbeg, nend = n * i // nthrds, n * (i + 1) // nthrds
        sel[nbeg:int(np.floor(ssfrac * nend))] = True
    assert 0 < sel.sum() < n
    rtr, com0, adb = 0.6 * np.abs(pos).max(), (0.002, -0.001, 0.0), 0.7
    with oracle.call_opts(adb=adb, rtrunc=rtr, com0=com0, ssfrac=ssfrac, nthrds=nthrds):
        c_ref, used = oracle.sph_accumulate(g, prm, pos, m)
    with oracle.call_opts(adb=adb, rtrunc=rtr, com0=com0):           # the same set spelled out: the selected particles, m / ssfrac
        c_sel, used_sel = oracle.sph_accumulate(g, prm, pos[sel], m[sel] / ssfrac)
    assert used == used_sel and np.abs(c_ref - c_sel).max() <= 1e-13 * np.abs(c_ref).max()
    f = SphereSL(ctx, g, **win)
    f.set_subset(ssfrac, nthrds)
    c = Component.from_arrays(ctx, m, pos, vel)
    c.set_rtrunc(rtr, com0)
    f.set_mass_scale(adb)
    f.determine_coefficients(c)
    coef = f.get_coefs()
    assert f.Used() == used
    assert np.abs(coef - c_ref).max() <= 1e-10 * np.abs(c_ref).max()
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download()
    assert np.array_equal(out["pos"], pos)                           # the store itself is left as it was
    with oracle.call_opts(rtrunc=rtr, com0=com0):
        a_ref, p_ref = oracle.sph_accel(g, prm, pos, coef)
    own = np.linalg.norm(a_ref, axis=1)
    ok = own > 0                                                     # (frozen particles are skipped by the force pass: zeros)
    assert (np.linalg.norm(out["acc"] - a_ref, axis=1)[ok] <= 1e-9 * own[ok]).all()
    # switched off again (not a sane value): the plain accumulation
    f.set_subset(0.0, nthrds)
    f.determine_coefficients(c)
    with oracle.call_opts(adb=adb, rtrunc=rtr, com0=com0):
        c_all, used_all = oracle.sph_accumulate(g, prm, pos, m)
    assert f.Used() == used_all and np.abs(f.get_coefs() - c_all).max() <= 1e-10 * np.abs(c_all).max()
    # the fused step with the key on == the call-for-call sequence (kick, drift, coefficients of the sub-sample, force, kick)
    f.set_subset(ssfrac, nthrds)
    f.set_mass_scale(1.0)
    c2 = Component.from_arrays(ctx, m, pos, vel)
    for cc in (c, c2):
        cc.set_rtrunc(1e20)
    c.upload(m, pos, vel)
    for cc in (c, c2):
        f.determine_coefficients(cc); cc.zero_acceleration(0); f.get_acceleration_and_potential(cc)
    for _ in range(3):
        f.step_kdk(c, 1.5e-4)
        do_step_single(f, c2, 1.5e-4)
    a, b = c.download(), c2.download()
    for k in ("pos", "vel", "acc"):
        assert np.abs(a[k] - b[k]).max() <= 1e-10 * np.abs(b[k]).max(), k
    for x in (c, c2, f):
        x.close()
    with pytest.raises(RuntimeError, match="multistep"):
        SphereSL(ctx, g, multistep=2, **win).set_subset(0.5, 1)


def test_sphere_noise_mode_against_the_oracle(ctx, oracle):
    """NOISE (src/SphericalBasis.cc:355, :395, :2108-2210) call by call: every force evaluation -- self, then external on another
    component -- first REPLACES the coefficient set by the next draws of the noise model (one generator per force, seeded
    once); the accumulation in between is the plain one.  Then the same through the step driver, single level (the self call
    evaluates the draws) -- against oracle/noise_oracle.cc, whose deviates are the reference's std::mt19937 +
    std::normal_distribution objects."""
    from exp_amd.runtime import Component, Simulation, SphereSL
    from exp_amd.slgrid import compute_rms_coefs
    inp = c4.config4_inputs(n_halo=700, n_disk=300)
    g, _ = c4.grids()
    sc = float(inp["scale"])
    win = c4.sph_window(g, sc)
    prm = oracle.params(**win)
    m, pos, vel = inp["halo_mass"], inp["halo_pos"], inp["halo_vel"]
    meanC, rmsC = compute_rms_coefs(g, NOISE_MODEL, sc)
    noiseN, seed = 1.0e-6, 4242
    h = oracle.noise_create(g.lmax, g.nmax, meanC, rmsC, noiseN, seed)
    f = SphereSL(ctx, g, **win)
    f.set_noise(NOISE_MODEL, noiseN=noiseN, seedN=seed)
    c = Component.from_arrays(ctx, m, pos, vel)
    t = Component.from_arrays(ctx, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
    f.determine_coefficients(c)
    c_acc, _ = oracle.sph_accumulate(g, prm, pos, m)
    assert np.abs(f.get_coefs() - c_acc).max() <= 1e-10 * np.abs(c_acc).max()      # (the accumulation knows nothing of it)
    for target, tpos, ext in ((c, pos, False), (t, inp["disk_pos"], True), (c, pos, False)):
        target.zero_acceleration(0)
        f.get_acceleration_and_potential(target, external=ext)
        cn = oracle.noise_update(h, g.lmax, g.nmax)
        assert np.abs(f.get_coefs() - cn).max() <= 1e-12 * np.abs(cn).max()
        a_ref, p_ref = oracle.sph_accel(g, prm, tpos, cn)
        o = target.download(("acc", "pot"))
        own = np.linalg.norm(a_ref, axis=1)
        assert (np.linalg.norm(o["acc"] - a_ref, axis=1) <= 1e-9 * own).all()
        assert np.abs(o["pot"] - p_ref).max() <= 1e-9 * np.abs(p_ref).max()
    oracle.lib.orc_noise_destroy(h)
    # off again: the set an accumulation leaves is the one evaluated
    f.set_noise(None)
    f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
    assert np.abs(f.get_coefs() - c_acc).max() <= 1e-10 * np.abs(c_acc).max()
    for x in (c, t, f):
        x.close()
    # the step driver, one component, single level: do_step's force evaluation is a self call that keeps its draws
    dtime, nsteps = 1.0e-4, 3
    nb = NBodyOracle(oracle, 0, dtime, c4.DYN)
    i1 = nb.add_sphere(g, prm, m, pos, vel)
    nb.set_options(i1, noise=(meanC, rmsC, 1.0, 9))
    nb.init()
    for _ in range(nsteps):
        nb.step()
    ch = Component.from_arrays(ctx, m, pos, vel)
    fh = SphereSL.from_config(ctx, g, dict(NOISE=True, noiseN=True, seedN=9, noise_model_file=NOISE_MODEL, scale=sc,
                                           rmin=win["rmin"], rmax=win["rmax"]))
    sim = Simulation(ctx, dtime, multistep=0, dynfrac=c4.DYN)
    sim.add_component(ch, fh)
    sim.init()
    sim.step(nsteps)
    st, o = nb.state[0], ch.download()
    a = np.stack([st["a" + q] for q in "xyz"], 1)
    p = np.stack([st[q] for q in "xyz"], 1)
    assert np.abs(o["pos"] - p).max() <= 1e-11 * np.abs(p).max()
    assert (np.linalg.norm(o["acc"] - a, axis=1) <= 1e-9 * np.linalg.norm(a, axis=1)).all()
    assert np.abs(np.asarray(fh.get_coefs()).reshape(-1) - st["coef"]).max() <= 1e-12 * np.abs(st["coef"]).max()
    for x in (sim, ch, fh):
        x.close()
    # several fused steps in one call: the draws are host-side effects a replayed graph would not repeat -- stepped eagerly
    outs = []
    for many in (True, False):
        cc = Component.from_arrays(ctx, m, pos, vel)
        ff = SphereSL(ctx, g, **win)
        ff.set_noise(NOISE_MODEL, noiseN=1.0, seedN=31)
        ff.determine_coefficients(cc); cc.zero_acceleration(0); ff.get_acceleration_and_potential(cc)
        if many:
            ff.step_kdk_n(cc, dtime, 6)
        else:
            for _ in range(6):
                ff.step_kdk(cc, dtime)
        outs.append((cc.download(), ff.get_coefs().copy()))
        cc.close(); ff.close()
    assert np.abs(outs[0][1] - outs[1][1]).max() <= 1e-13 * np.abs(outs[1][1]).max()
    for k in ("pos", "vel", "acc"):
        assert np.abs(outs[0][0][k] - outs[1][0][k]).max() <= 1e-10 * np.abs(outs[1][0][k]).max(), k
    with pytest.raises(ValueError, match="seedN"):
        SphereSL.from_config(ctx, g, dict(NOISE=True))
    with pytest.raises(ValueError, match="boolean"):
        SphereSL.from_config(ctx, g, dict(NOISE=True, seedN=1, noiseN=0))
