"""Device Orient (exp_amd/csrc/orient.hip) against the oracle's restatement of src/Orient.cc over
a run: the energy threshold (exact radix select) must be the SAME double, the number of particles
used the same integer, centre / axis / rotations equal to round-off.  GPU only."""
import os

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


def _compare(o, ref, tol=1e-12):
    st = o.state()
    assert st["Ecurr"] == ref.Ecurr                                   # bit for bit
    assert st["used"] == ref.used
    assert st["mtot"] == pytest.approx(ref.mtot, rel=1e-13)
    for k in ("center", "axis", "axis1", "center1", "center0"):
        r = np.array(getattr(ref, k)[:])
        assert np.abs(st[k] - r).max() <= tol * max(1.0, np.abs(r).max()), k
    assert np.abs(st["body"] - np.array(ref.body[:]).reshape(3, 3)).max() <= 1e-10
    assert np.abs(st["orig"] - np.array(ref.orig[:]).reshape(3, 3)).max() <= 1e-10
    for k, r in (("sigA", ref.sigA), ("sigC", ref.sigC), ("sigCz", ref.sigCz)):
        assert st[k] == pytest.approx(r, rel=1e-6, abs=1e-24), k


@pytest.mark.parametrize("ke", [False, True])
def test_orient_follows_the_oracle_over_a_run(ctx, oracle, ke):
    """A drifting, slightly rotating halo stepped with the fused KDK step (so the closing half kick
    is still pending when Orient reads the velocities): every call agrees with the oracle fed the
    downloaded state."""
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Orient, SphereSL
    model, g = make_grid("plummer", 4, 8, 400)
    n = 30000
    m, pos, vel = sample_sphere(model, n, seed=41)
    pos = pos + np.array([0.2, -0.1, 0.05])
    vel = vel + np.array([0.3, 0.1, -0.2]) + 0.2 * np.cross([0.0, 0.3, 1.0], pos)
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m, pos, vel)
    c.set_center([0.2, -0.1, 0.05])
    f.determine_coefficients(c); c.zero_acceleration(); f.get_acceleration_and_potential(c)
    keep, many, dt = 3, 2000, 0.02
    cfl = Orient.KE if ke else 0
    o = Orient(ctx, keep, many, Orient.AXIS | Orient.CENTER, cfl, dT=0.0, damping=0.7)
    ref = oracle.orient(keep, many, 3, cfl, 0.0, 0.7)
    for k in range(8):
        t = k * dt
        o.accumulate(t, c, dt)
        d = c.download(("mass", "pos", "vel", "pot"))
        oracle.orient_accumulate(ref, t, dt, d["mass"], d["pos"], d["vel"], d["pot"])
        _compare(o, ref)
        assert o.currentUsed() == many and o.currentE() == ref.Ecurr
        # the component's expansion centre follows the estimate (Component::fix_positions :3569-3582)
        c.set_center(o.currentCenter() if k else [0.2, -0.1, 0.05])
        f.step_kdk(c, dt)
    assert ref.nA == keep + 1 and ref.nC == keep + 1                 # regression branches exercised
    assert np.abs(o.transformBody() @ o.transformOrig() - np.eye(3)).max() < 1e-13
    o.close(); c.close(); f.close()


def test_orient_small_component_duplicates_and_modes(ctx, oracle):
    from exp_amd.runtime import Component, Orient
    rng = np.random.default_rng(3)
    n = 777
    m = rng.uniform(0.5, 1.5, n) / n
    pos, vel = rng.standard_normal((2, n, 3))
    pot = -1.0 / np.sqrt(0.1 + (pos ** 2).sum(axis=1))
    # +0 / -0 and exact ties AT THE TOP: both sides exclude the threshold energy.  (Exact ties BELOW it are a documented
    # deviation, DESIGN.md section 2: the reference's std::set keeps one particle per energy.)
    pot[::7] = 0.0
    pot[7::14] = -0.0
    c = Component.from_arrays(ctx, m, pos, vel)
    c.upload_acc(np.zeros((n, 3)), pot)
    # fewer particles than wanted: the threshold is the highest energy, which is excluded (:474-489)
    for many in (5000, n, n - 1, 1):
        o = Orient(ctx, 1, many, Orient.CENTER)
        ref = oracle.orient(1, many, 2)
        o.accumulate(0.0, c)
        oracle.orient_accumulate(ref, 0.0, 0.0, m, pos, vel, pot)
        st = o.state()
        assert st["Ecurr"] == ref.Ecurr and st["used"] == ref.used
        assert np.abs(st["center"] - np.array(ref.center[:])).max() <= 1e-13
        o.close()
    # spacing by dT and the linear mode
    o = Orient(ctx, 2, 50, Orient.CENTER, dT=0.5)
    o.accumulate(0.0, c); e0 = o.state()
    c.incr_position(1.0)
    o.accumulate(0.3, c)
    assert np.array_equal(o.state()["center1"], e0["center1"])          # ignored: too soon
    o.accumulate(0.5, c)
    assert not np.array_equal(o.state()["center1"], e0["center1"])
    o.close()
    o = Orient(ctx, 2, 50, Orient.CENTER)
    o.set_center(1.0, 2.0, 3.0); o.set_cenvel(0.5, 0.0, -0.5); o.set_linear()
    o.accumulate(0.0, c, 0.2)
    st = o.state()
    assert list(st["center"]) == [1.0, 2.0, 3.0] and list(st["center0"]) == [1.1, 2.0, 2.9]
    o.close()
    # refused configurations
    with pytest.raises(RuntimeError, match="keep"):
        Orient(ctx, 0, 50, Orient.CENTER)
    # EXTERNAL (energy += potext, src/Orient.cc:377) is accepted: potext only ever holds the External plug-ins' potential
    # (out of scope: zero), so the state is the one without the flag
    o1 = Orient(ctx, 2, 50, Orient.CENTER, Orient.KE | Orient.EXTERNAL)
    o2 = Orient(ctx, 2, 50, Orient.CENTER, Orient.KE)
    o1.accumulate(0.0, c, 0.2); o2.accumulate(0.0, c, 0.2)
    s1, s2 = o1.state(), o2.state()
    assert all(np.array_equal(np.asarray(s1[k]), np.asarray(s2[k])) for k in s2)
    o1.close(); o2.close()
    c.close()


def test_orient_selection_at_scale(ctx):
    """2e7 particles: the radix select returns exactly numpy's (many+1)-th smallest energy and the
    sums match a float64 numpy reduction."""
    from exp_amd.runtime import Component, Orient
    rng = np.random.default_rng(8)
    n, many = 20_000_000, 100_000
    pos = rng.standard_normal((n, 3)) * 0.5
    vel = rng.standard_normal((n, 3)) * 0.3
    m = np.full(n, 1.0 / n)
    pot = -1.0 / np.sqrt(0.05 + (pos ** 2).sum(axis=1))
    c = Component.from_arrays(ctx, m, pos, vel)
    c.upload_acc(np.zeros((1, 3)).repeat(n, axis=0), pot)
    o = Orient(ctx, 1, many, Orient.AXIS | Orient.CENTER, Orient.KE)
    o.accumulate(0.0, c)
    st = o.state()
    E = pot + 0.5 * ((vel[:, 0] * vel[:, 0] + vel[:, 1] * vel[:, 1]) + vel[:, 2] * vel[:, 2])
    Ecurr = np.partition(E, many)[many]
    sel = E < Ecurr
    assert st["Ecurr"] == Ecurr and st["used"] == int(sel.sum()) == many
    assert st["mtot"] == pytest.approx(m[sel].sum(), rel=1e-12)
    assert np.abs(st["center1"] - (m[sel, None] * pos[sel]).sum(axis=0) / m[sel].sum()).max() < 1e-12
    L = (m[sel, None] * np.cross(pos[sel], vel[sel])).sum(axis=0) / m[sel].sum()
    assert np.abs(st["axis1"] - L).max() < 1e-12
    o.close(); c.close()


def test_simulation_follows_the_orient_centre(ctx):
    """The step loop with an estimator attached (exp_amd_sim_set_orient) against the same sequence
    spelled out call by call: centre := estimator's current centre, estimator takes in the state
    (only once potentials exist), then the force -- src/ComponentContainer.cc:955-959, :1386-1389,
    src/Component.cc:3357, :3569-3582."""
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Orient, Simulation, SphereSL
    model, g = make_grid("plummer", 4, 8, 400)
    n, dt, nstep = 20000, 0.02, 5
    m, pos, vel = sample_sphere(model, n, seed=43)
    pos = pos + np.array([0.3, 0.0, -0.1])
    vel = vel + np.array([0.5, -0.25, 0.125])

    def mk():
        return SphereSL(ctx, g), Component.from_arrays(ctx, m, pos, vel), \
            Orient(ctx, 2, 1500, Orient.CENTER, Orient.KE, dT=0.0, damping=1.0)

    f, c, o = mk()
    sim = Simulation(ctx, dt)
    sim.add_component(c, f)
    sim.set_orient(0, o)
    sim.init()
    sim.step(nstep)
    got = c.download(("pos", "vel", "acc", "pot"))
    got_state = o.state()
    sim.close(); o.close(); c.close(); f.close()

    f, c, o = mk()
    centers = []

    def potential(tnow, gottapot):
        ctr = o.currentCenter()
        c.set_center(ctr if not np.isnan(ctr).any() else np.zeros(3))
        centers.append(ctr.copy())
        if gottapot:
            o.accumulate(tnow, c, dt)
        c.zero_acceleration(0)
        f.get_acceleration_and_potential(c)

    f.set_multistep_level(0)
    f.determine_coefficients(c)
    potential(0.0, False)
    t = 0.0
    for k in range(nstep):
        t += dt
        c.incr_velocity(0.5 * dt); c.incr_position(dt)
        f.determine_coefficients(c)
        potential(t, True)
        c.incr_velocity(0.5 * dt)
    ref = c.download(("pos", "vel", "acc", "pot"))
    for k in ref:       # round-off only: the coefficient sums are order-dependent at the last bit
        assert np.abs(got[k] - ref[k]).max() <= 1e-10 * np.abs(ref[k]).max(), k
    st = o.state()
    assert got_state["Ecurr"] == pytest.approx(st["Ecurr"], rel=1e-12)
    assert np.abs(got_state["center"] - st["center"]).max() < 1e-12
    # the centre did move with the halo, one call behind the estimate
    # (keep = 2 with a full history mixes in a quarter of center0 = 0, src/Orient.cc:695-699)
    assert np.abs(centers[-1] - (np.array([0.3, 0.0, -0.1]) + np.array([0.5, -0.25, 0.125]) * (t - dt))).max() < 0.1
    assert centers[-1][0] > 0.25
    assert np.array_equal(centers[0], np.zeros(3)) and np.array_equal(centers[1], np.zeros(3))
    o.close(); c.close(); f.close()


def test_pseudo_acceleration_of_the_component_frame(ctx, oracle):
    """Component::AddAcc subtracts getPseudoAccel(pos, vel) on every call (src/Component.H:914-921,
    src/Component.cc:4407-4427).  With the frame acceleration set on the component, every force pass
    -- unfused, external target, fused step, cylinder -- gives the plain force minus the oracle's
    pseudo-acceleration of the stored position and velocity, once per AddAcc call the reference's
    thread body makes (the cylinder: one per axis; the sphere: two for x and y)."""
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Cylinder, SphereSL
    from tests.test_cyl_gpu import cyl_grid, _disk
    model, g = make_grid("plummer", 4, 8, 400)
    n = 20000
    m, pos, vel = sample_sphere(model, n, seed=47)
    acc0, om, dom = np.array([0.3, -0.2, 0.1]), np.array([0.02, -0.05, 0.4]), np.array([0.01, 0.03, -0.02])
    f = SphereSL(ctx, g)

    def force(setter, fused=False, external=False):
        c = Component.from_arrays(ctx, m, pos, vel)
        src = c
        if external:                                   # coefficients from another component
            src = Component.from_arrays(ctx, m, pos, vel)
        f.determine_coefficients(src)
        setter(c)
        if fused:
            c.zero_acceleration(0)
            f.get_acceleration_and_potential(c)        # a(0) for the opening kick
            f.step_kdk(c, 0.01)
        else:
            c.zero_acceleration(0)
            f.get_acceleration_and_potential(c, external=external)
        out = c.download(("pos", "vel", "acc"))
        c.close()
        if external:
            src.close()
        return out

    # The spherical thread body hands its force to AddAcc in FIVE calls (x, y, z, then x and y again
    # for the azimuthal term when x^2 + y^2 > DSMALL, src/SphericalBasis.cc:1645-1651), and every call
    # subtracts the frame term: x and y lose it twice.  twice = that factor, as the oracle restates it.
    twice = np.array([2.0, 2.0, 1.0])
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    for external in (False, True):
        plain = force(lambda c: None, external=external)
        for cen, ax in ((True, False), (False, True), (True, True)):
            got = force(lambda c: c.set_pseudo_accel(acc0 if cen else None, om if ax else None,
                                                     dom if ax else None), external=external)
            ps = oracle.get_pseudo_accel(int(cen), int(ax), acc0, om, dom, pos, vel)
            assert np.abs(got["acc"] - (plain["acc"] - twice * ps)).max() <= 1e-12 * np.abs(plain["acc"]).max()
            if not external:
                coef, _ = oracle.sph_accumulate(g, prm, pos, m)
                ref, _ = oracle.sph_accel_pseudo(g, prm, pos, coef, ps)
                assert np.abs(got["acc"] - ref).max() <= 1e-9 * np.linalg.norm(ref, axis=1).max()
    # fused step: the closing force is the plain force at the new position minus the pseudo term of
    # the position and (half-kicked) velocity the force pass saw
    got = force(lambda c: c.set_pseudo_accel(acc0, om, dom), fused=True)
    c = Component.from_arrays(ctx, m, got["pos"], got["vel"])
    f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
    plain_acc = c.download(("acc",))["acc"]
    c.close()
    vhalf = got["vel"] - 0.005 * got["acc"]            # undo the closing half-kick
    ps = oracle.get_pseudo_accel(1, 1, acc0, om, dom, got["pos"], vhalf)
    assert np.abs(got["acc"] - (plain_acc - twice * ps)).max() <= 1e-9 * np.abs(plain_acc).max()
    f.close()
    # cylinder
    gc = cyl_grid(4, 6)
    md, pd, _ = _disk(10000, 93, gc)
    vd = 3.0 * np.cross([0.0, 0.0, 1.0], pd)
    fc = Cylinder(ctx, gc)
    outs = []
    for on in (False, True):
        c = Component.from_arrays(ctx, md, pd, vd)
        fc.determine_coefficients(c)
        if on:
            c.set_pseudo_accel(acc0, om, dom)
        c.zero_acceleration(0)
        fc.get_acceleration_and_potential(c)
        outs.append(c.download(("acc",))["acc"])
        c.close()
    ps = oracle.get_pseudo_accel(1, 1, acc0, om, dom, pd, vd)
    assert np.abs(outs[1] - (outs[0] - ps)).max() <= 1e-12 * np.abs(outs[0]).max()
    fc.close()


def test_orient_pseudo_accel_estimates_and_step_loop(ctx, oracle):
    """Orient's Naccel helper against the oracle's PseudoAccel fed the same (time, centre, axis)
    estimates, and the step loop handing the centre acceleration to the component."""
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Orient, Simulation, SphereSL
    model, g = make_grid("plummer", 4, 8, 400)
    n, dt, nacc = 20000, 0.02, 5
    m, pos, vel = sample_sphere(model, n, seed=53)
    vel = vel + np.array([0.4, -0.2, 0.1])
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m, pos, vel)
    o = Orient(ctx, 2, 2000, Orient.AXIS | Orient.CENTER, Orient.KE)
    o.set_naccel(nacc)
    f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
    rows = []
    for k in range(8):
        o.accumulate(k * dt, c, dt)
        st = o.state()
        rows.append([k * dt, *st["center1"], *st["axis1"]])
        acc, om, dom = o.currentAccel()
        if len(rows) >= nacc:
            ra, ro, rd = oracle.pseudo_accel_fit(np.array(rows[-nacc:]))
            assert np.allclose(acc, ra, rtol=0, atol=1e-9 * max(1.0, np.abs(ra).max()))
            assert np.allclose(om, ro, rtol=0, atol=1e-9) and np.allclose(dom, rd, rtol=0, atol=1e-8)
        else:
            assert not acc.any() and not om.any() and not dom.any()      # queue not full yet
        f.step_kdk(c, dt)
    # log file in the reference's 33-column layout
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        log = os.path.join(d, "halo.orient.run0")
        o.logHeader(log)
        o.logEntry(7 * dt, log, com=(0.1, 0.2, 0.3))
        lines = open(log).read().splitlines()
        assert len(lines) == 3 and lines[0].startswith("# Time") and lines[0].count("|") == 32
        row = [float(v) for v in lines[2].split()]
        st = o.state()
        assert len(row) == 33 and row[0] == pytest.approx(7 * dt) and row[2] == st["used"]
        assert row[9:12] == pytest.approx(list(st["center"]), rel=1e-5, abs=1e-12) and row[18:21] == [0.1, 0.2, 0.3]
    o.close(); c.close()
    # step loop: the estimate reaches the component (no crash, finite, small for a coasting halo)
    c = Component.from_arrays(ctx, m, pos, vel)
    o = Orient(ctx, 2, 2000, Orient.CENTER, Orient.KE)
    o.set_naccel(4)
    sim = Simulation(ctx, dt)
    sim.add_component(c, f)
    sim.set_orient(0, o)
    sim.init()
    sim.step(8)
    acc, _, _ = o.currentAccel()
    assert np.isfinite(acc).all() and acc.any()      # (4 noisy centres 0.02 apart: not a small number)
    out = c.download(("acc",))["acc"]
    assert np.isfinite(out).all()
    sim.close(); o.close(); c.close(); f.close()


def test_orient_restart_from_its_log(ctx, oracle, tmp_path):
    """The restart block of the constructor (src/Orient.cc:84-335): a run writes its log through
    Orient::logEntry, a second estimator re-reads it up to the restart time.  The device estimator
    and the oracle's restatement must (a) write the same header bytes, (b) copy the same rows into the
    fresh log and leave the old one as <log>.bak, (c) hold the same state after the restart -- the
    values are the log's 6-digit ones, so (c) is exact -- and (d) continue the run alike."""
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Orient, SphereSL
    model, g = make_grid("plummer", 4, 8, 400)
    n, dt, keep, many, nacc = 20000, 0.02, 3, 1500, 4
    m, pos, vel = sample_sphere(model, n, seed=77)
    pos = pos + np.array([0.1, 0.05, -0.02])
    vel = vel + np.array([0.2, -0.3, 0.1]) + 0.2 * np.cross([0.1, 0.2, 1.0], pos)
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m, pos, vel)
    f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
    log, rlog = str(tmp_path / "halo.orient.run0"), str(tmp_path / "ref.orient.run0")
    o = Orient(ctx, keep, many, Orient.AXIS | Orient.CENTER, Orient.KE, dT=0.0, damping=0.8)
    o.set_naccel(nacc)
    assert o.openLog(log) == 0
    ref = oracle.orient(keep, many, 3, Orient.KE, 0.0, 0.8)
    assert oracle.orient_restart(ref, rlog, True, 0.0, dt, 1)[0] == 0
    assert open(log, "rb").read() == open(rlog, "rb").read()          # (a) the two header rows
    states = []
    for k in range(9):
        t = k * dt
        o.accumulate(t, c, dt)
        o.logEntry(t, com=(0.01 * k, 0.0, -0.02))
        states.append(c.download(("mass", "pos", "vel", "pot")))
        f.step_kdk(c, dt)
    full = open(log).read().splitlines()
    assert len(full) == 2 + 9 and all(len(r.split()) == 33 for r in full[2:])
    # the oracle restarts from a copy of the same file
    open(rlog, "w").write(open(log).read())
    tnow = 6 * dt
    o2 = Orient(ctx, keep, many, Orient.AXIS | Orient.CENTER, Orient.KE, dT=0.0, damping=0.8)
    o2.set_naccel(nacc)
    rows = o2.openLog(log, restart=True, tnow=tnow, dtime=dt, Mstep=1)
    rrows, rq = oracle.orient_restart(ref, rlog, True, tnow, dt, 1, naccel=nacc)
    assert rows == rrows == 7                                           # rows 0..6, the cut is t <= tnow + 0.1 dt
    assert open(log, "rb").read() == open(rlog, "rb").read()          # (b)
    assert open(log).read().splitlines() == full[2:9]
    assert open(log + ".bak").read().splitlines() == full
    st = o2.state()
    for k in ("center", "axis", "axis1", "center1", "center0"):         # (c)
        assert np.array_equal(st[k], np.array(getattr(ref, k)[:])), k
    assert st["Ecurr"] == ref.Ecurr
    assert np.abs(st["body"] - np.array(ref.body[:]).reshape(3, 3)).max() <= 1e-15
    last = np.array(full[8].split(), dtype=float)
    assert np.array_equal(st["center"], last[9:12]) and np.array_equal(st["axis1"], last[6:9])
    # the queue holds (time, logged pseudo-acceleration, axis1) of the last `nacc` rows, as the
    # reference's does (:174-186); with queue_center1 the centre estimate instead
    tab = np.array([r.split() for r in full[2:9]], dtype=float)
    assert np.array_equal(rq, np.hstack([tab[-nacc:, 0:1], tab[-nacc:, 24:27], tab[-nacc:, 6:9]]))
    acc, om, dom = o2.currentAccel()
    ra, ro, rd = oracle.pseudo_accel_fit(rq)
    assert np.allclose(acc, ra, rtol=0, atol=1e-9 * max(1.0, np.abs(ra).max()))
    assert np.allclose(om, ro, rtol=0, atol=1e-9) and np.allclose(dom, rd, rtol=0, atol=1e-7)
    # (d) both continue from the restored histories with the run's next states
    for k in (7, 8):
        d = states[k]
        cc = Component.from_arrays(ctx, d["mass"], d["pos"], d["vel"])
        cc.upload_acc(np.zeros_like(d["pos"]), d["pot"])
        oracle.orient_accumulate(ref, k * dt, dt, d["mass"], d["pos"], d["vel"], d["pot"])
        o2.accumulate(k * dt, cc, dt)
        _compare(o2, ref)
        cc.close()
    # no restart: the old log is set aside and nothing is read back (:128 `while (in && restart)`)
    o3 = Orient(ctx, keep, many, Orient.AXIS | Orient.CENTER, Orient.KE)
    assert o3.openLog(log, restart=False, tnow=tnow, dtime=dt) == 0
    assert open(log).read() == "" and np.array_equal(o3.currentAxis(), [0.0, 0.0, 1.0])
    # queue_center1: the evident intention
    open(log, "w").write("\n".join(full) + "\n")
    o4 = Orient(ctx, keep, many, Orient.AXIS | Orient.CENTER, Orient.KE)
    o4.set_naccel(nacc)
    assert o4.openLog(log, restart=True, tnow=tnow, dtime=dt, queue_center1=True) == 7
    a4, _, _ = o4.currentAccel()
    r4, _, _ = oracle.pseudo_accel_fit(np.hstack([tab[-nacc:, 0:1], tab[-nacc:, 15:18], tab[-nacc:, 6:9]]))
    assert np.allclose(a4, r4, rtol=0, atol=1e-7 * max(1.0, np.abs(r4).max()))
    for x in (o, o2, o3, o4, c):
        x.close()
