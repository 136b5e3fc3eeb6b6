"""Block multistep (src/step.cc:98-269, src/multistep.cc) and multi-component stepping on the
device against the oracle.  GPU only."""
import numpy as np
import pytest

from tests.conftest import make_grid

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from exp_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def _halo(n, seed):
    from exp_amd.models import sample_sphere
    model, g = make_grid("plummer", 4, 8, 400)
    m, pos, vel = sample_sphere(model, n, seed=seed)
    pos[:, 2] *= 0.8
    return g, m, pos, vel


def test_multistep_master_steps_match_oracle(ctx, oracle):
    from exp_amd.runtime import Component, Simulation, SphereSL
    g, m, pos, vel = _halo(4000, 31)
    ms, dtime = 3, 0.05
    dyn = [1000.0, 0.01, 0.01, 0.03, 0.05]          # src/global.cc:76-80
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    st = oracle.sph_multistep_init(g, prm, ms, dtime, dyn, 0, pos, vel, m)

    f = SphereSL(ctx, g, multistep=ms)
    c = Component.from_arrays(ctx, m, pos, vel)
    sim = Simulation(ctx, dtime, multistep=ms, dynfrac=dyn, shiftlevl=0)
    sim.add_component(c, f)
    sim.init()

    def compare(tag):
        lev = c.download_levels()
        assert np.array_equal(lev, st["level"]), (tag, int((lev != st["level"]).sum()))
        out = c.download()
        p = np.stack([st["x"], st["y"], st["z"]], 1)
        v = np.stack([st["vx"], st["vy"], st["vz"]], 1)
        a = np.stack([st["ax"], st["ay"], st["az"]], 1)
        assert np.abs(out["pos"] - p).max() <= 1e-11, tag
        assert np.abs(out["vel"] - v).max() <= 1e-10, tag
        ascale = np.linalg.norm(a, axis=1).max()
        assert np.abs(out["acc"] - a).max() <= 1e-8 * ascale, tag
        cmax = np.abs(st["coefN"]).max()
        for M in range(ms + 1):
            cn = f.get_coefs(level=M).reshape(-1)
            cl = f.get_coefs(level=M, last=True).reshape(-1)
            assert np.abs(cn - st["coefN"][M]).max() <= 1e-10 * cmax, (tag, M)
            assert np.abs(cl - st["coefL"][M]).max() <= 1e-10 * cmax, (tag, M)

    compare("init")
    assert len(np.unique(st["level"])) >= 3          # the test must actually exercise several levels
    total_switch = 0
    for k in range(2):
        total_switch += oracle.sph_multistep_step(g, prm, st)
        sim.step(1)
        compare(f"step{k}")
        assert np.abs(f.get_coefs().reshape(-1) - st["coef"]).max() <= 1e-10 * np.abs(st["coef"]).max()
    assert total_switch > 0                          # ... and level changes (multistep_update)
    assert sim.time == pytest.approx(2 * dtime)


def test_single_level_multistep_equals_multistep0(ctx):
    """SURVEY 8c KAT (viii): if nobody leaves level 0, a multistep master step is the plain KDK
    step (two components with mutual interactions, so compute_potential's loops are exercised)."""
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    from tests.test_cyl_gpu import _disk, cyl_grid
    g, m, pos, vel = _halo(3000, 5)
    cg = cyl_grid(4, 6)
    pos = pos * (3.0 * cg.ascale)                    # put the halo on the disk's scale
    vel = vel * 0.05
    dm, dpos, dvel = _disk(3000, 6, cg)
    dvel = dvel + 0.3 * np.random.default_rng(2).standard_normal(dvel.shape)   # v = 0 would force dt -> eps
    res = {}
    for ms in (0, 2):
        f1 = SphereSL(ctx, g, scale=3.0 * cg.ascale, rmin=g.rmin * 3.0 * cg.ascale,
                      rmax=g.rmax * 3.0 * cg.ascale, multistep=ms)
        f2 = Cylinder(ctx, cg, multistep=ms)
        c1 = Component.from_arrays(ctx, m, pos, vel)
        c2 = Component.from_arrays(ctx, dm, dpos, dvel)
        sim = Simulation(ctx, 1e-4, multistep=ms, dynfrac=[1e9] * 5)     # nobody wants a shorter step
        i1 = sim.add_component(c1, f1)
        i2 = sim.add_component(c2, f2)
        sim.add_interaction(i1, i2)
        sim.add_interaction(i2, i1)
        sim.init()
        sim.step(2)
        res[ms] = (c1.download(), c2.download(), f1.get_coefs(), f2.get_coefs())
        if ms:
            assert np.all(c1.download_levels() == 0) and np.all(c2.download_levels() == 0)
        for o in (sim, c1, c2, f1, f2):
            o.close()
    for k in (0, 1):
        for key in ("pos", "vel", "acc", "pot"):
            a, b = res[0][k][key], res[2][k][key]
            assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max()), (k, key)


def test_two_component_step_matches_oracle_pieces(ctx, oracle):
    """Disk + halo, multistep 0, both self-forces and both cross-forces
    (src/ComponentContainer.cc:698-716, :785-853) against the oracle's per-component pieces."""
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    from tests.test_cyl_gpu import _disk, cyl_grid
    g, m, pos, vel = _halo(3000, 15)
    cg = cyl_grid(4, 6)
    sc = 3.0 * cg.ascale
    pos, vel = pos * sc, vel * 0.05
    dm, dpos, dvel = _disk(3000, 16, cg)
    dt = 1e-4
    prm = oracle.params(scale=sc, rmin=g.rmin * sc, rmax=g.rmax * sc)

    f1 = SphereSL(ctx, g, scale=sc, rmin=g.rmin * sc, rmax=g.rmax * sc)
    f2 = Cylinder(ctx, cg)
    c1 = Component.from_arrays(ctx, m, pos, vel)
    c2 = Component.from_arrays(ctx, dm, dpos, dvel)
    sim = Simulation(ctx, dt)
    i1, i2 = sim.add_component(c1, f1), sim.add_component(c2, f2)
    sim.add_interaction(i1, i2)
    sim.add_interaction(i2, i1)
    sim.init()

    def forces(p1, p2):
        ch, _ = oracle.sph_accumulate(g, prm, p1, m)
        cc, ss, _, cmass = oracle.cyl_accumulate(cg, p2, dm)
        a1, q1 = oracle.sph_accel(g, prm, p1, ch)
        b1, r1 = oracle.cyl_accel(cg, p1, cc, ss, cmass)           # disk force on halo particles
        a2, q2 = oracle.cyl_accel(cg, p2, cc, ss, cmass)
        b2, r2 = oracle.sph_accel(g, prm, p2, ch)                  # halo force on disk particles
        return a1 + b1, q1 + r1, a2 + b2, q2 + r2

    A1, P1, A2, P2 = forces(pos, dpos)
    o1, o2 = c1.download(), c2.download()
    s1, s2 = np.linalg.norm(A1, axis=1).max(), np.linalg.norm(A2, axis=1).max()
    assert np.abs(o1["acc"] - A1).max() <= 1e-9 * s1
    assert np.abs(o2["acc"] - A2).max() <= 1e-9 * s2
    assert np.abs(o1["pot"] - P1).max() <= 1e-9 * np.abs(P1).max()
    # one KDK step
    v1 = vel + A1 * (0.5 * dt)
    v2 = dvel + A2 * (0.5 * dt)
    p1 = pos + v1 * dt
    p2 = dpos + v2 * dt
    A1n, _, A2n, _ = forces(p1, p2)
    v1 = v1 + A1n * (0.5 * dt)
    v2 = v2 + A2n * (0.5 * dt)
    sim.step(1)
    o1, o2 = c1.download(), c2.download()
    assert np.abs(o1["pos"] - p1).max() <= 1e-11 and np.abs(o2["pos"] - p2).max() <= 1e-11
    assert np.abs(o1["vel"] - v1).max() <= 1e-9 * np.abs(v1).max()
    assert np.abs(o2["vel"] - v2).max() <= 1e-9 * np.abs(v2).max()
    assert np.abs(o1["acc"] - A1n).max() <= 1e-9 * s1
    assert np.abs(o2["acc"] - A2n).max() <= 1e-9 * s2
