"""BASELINE config 4 (disk + halo, SphericalSL + EmpCylSL, multistep 4, both self forces and both
cross forces, level changes in BOTH components) on the device against the n-body oracle
(oracle/nbody_oracle.c: do_step src/step.cc:98-269, ComponentContainer::compute_potential
src/ComponentContainer.cc:698-853, adjust_multistep_level src/multistep.cc:344-627, CylEXP's
multistep_update / _finish / compute_multistep_coefficients src/CylEXP.cc:56-282) and against the
frozen golden vector tests/golden/config4_small.npz.  GPU only.

Bars: levels bit-exact after begin_run and after every master step; every per-level coefficient set
(expcoefN/L, cosN/L, sinN/L) within 1e-10 of the largest coefficient; positions 1e-11 absolute,
velocities / accelerations / potentials 1e-9 relative to their largest value."""
import numpy as np
import pytest

from tests import config4_util as c4

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from exp_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def _device_run(ctx, z, multistep, dtime, dyn):
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    g, cg = c4.grids()
    f1 = SphereSL(ctx, g, multistep=multistep, **c4.sph_window(g, float(z["scale"])))
    f2 = Cylinder(ctx, cg, multistep=multistep)
    c1 = Component.from_arrays(ctx, z["halo_mass"], z["halo_pos"], z["halo_vel"])
    c2 = Component.from_arrays(ctx, z["disk_mass"], z["disk_pos"], z["disk_vel"])
    sim = Simulation(ctx, dtime, multistep=multistep, dynfrac=dyn, shiftlevl=0)
    i1, i2 = sim.add_component(c1, f1), sim.add_component(c2, f2)
    sim.add_interaction(i1, i2)
    sim.add_interaction(i2, i1)
    sim.init()
    return sim, (f1, f2), (c1, c2)


def _compare(tag, forces, comps, want, multistep):
    """`want(name, key)` returns the oracle's array."""
    for name, f, c in zip(("halo", "disk"), forces, comps):
        lev = c.download_levels()
        ref = want(name, "level")
        assert np.array_equal(lev, ref), (tag, name, int((lev != ref).sum()))
        out = c.download()
        p = np.stack([want(name, k) for k in "xyz"], 1)
        v = np.stack([want(name, "v" + k) for k in "xyz"], 1)
        a = np.stack([want(name, "a" + k) for k in "xyz"], 1)
        assert np.abs(out["pos"] - p).max() <= 1e-11, (tag, name)
        assert np.abs(out["vel"] - v).max() <= 1e-9 * np.abs(v).max(), (tag, name)
        assert np.abs(out["acc"] - a).max() <= 1e-9 * np.linalg.norm(a, axis=1).max(), (tag, name)
        assert np.abs(out["pot"] - want(name, "pot")).max() <= 1e-9 * np.abs(want(name, "pot")).max(), (tag, name)
        cN, cL = want(name, "coefN"), want(name, "coefL")
        cmax = np.abs(cN).max()
        for M in range(multistep + 1):
            if name == "halo":
                gn = f.get_coefs(level=M).reshape(-1)
                gl = f.get_coefs(level=M, last=True).reshape(-1)
            else:
                gn = np.concatenate([x.reshape(-1) for x in f.get_coefs(level=M)])
                gl = np.concatenate([x.reshape(-1) for x in f.get_coefs(level=M, last=True)])
            assert np.abs(gn - cN[M]).max() <= 1e-10 * cmax, (tag, name, M, "N")
            assert np.abs(gl - cL[M]).max() <= 1e-10 * cmax, (tag, name, M, "L")


@pytest.mark.parametrize("dense_min,list_min,thin_max", [
    (-1, 2048, 16384), (0, 2048, 16384), (100, 2048, 16384), (0, 0, 16384), (100, 0, 0), (-1, 16, 0), (-1, 2048, 0),
    (100, 2048, 40), (-1, 16, 300)])
def test_config4_against_the_nbody_oracle(ctx, oracle, dense_min, list_min, thin_max):
    """thin_max: the size of an active slot range up to which it is accumulated and evaluated straight from the basis
    tables (runtime.Context.set_thin_max; 8192 = the default; 16384 here: with dense_min -1 EVERY sub-step of this small run takes
    the direct kernels of both bases -- k_sph_acc_thin / k_sph_force_thin, k_cyl_acc_thin / k_cyl_force_thin -- for the
    self forces and both cross forces; 0: never, the moment / projected-table path alone; 40 and 300: the two mixed, by
    sub-step).  dense_min: the level population below which a level is kept unsorted (runtime.Context.
    set_dense_min): the default (-1) makes every level of this small run sparse, 0 makes all of them
    cell-sorted, 100 mixes the two paths.  list_min: the number of level changes in a sweep from which the
    differencing runs the list of movers through the accumulation kernels (set_mover_list_min; 8192 = the
    default; at 2048: per-mover atomics throughout this small run, 0: always the accumulation kernels, 16: both)."""
    ctx.set_dense_min(dense_min)
    ctx.set_mover_list_min(list_min)
    ctx.set_thin_max(thin_max)
    z = c4.load_golden()
    ms, dtime, dyn = c4.MULTISTEP, c4.DTIME, c4.DYN
    nb, _ = c4.oracle_run(oracle, z, nsteps=0)
    sim, forces, comps = _device_run(ctx, z, ms, dtime, dyn)

    def want(name, key):
        return nb.state[0 if name == "halo" else 1][key]

    _compare("init", forces, comps, want, ms)
    assert forces[1].cylmass == pytest.approx(nb.cylmass(1), rel=1e-12)
    total = [0, 0]
    for k in range(c4.NSTEPS + 1):                       # one master step more than the golden file holds
        nsw = nb.step()
        total = [a + b for a, b in zip(total, nsw)]
        sim.step(1)
        assert sim.step_switches == sum(nsw), (k, sim.step_switches, nsw)
        _compare(f"step{k}", forces, comps, want, ms)
        assert forces[1].cylmass == pytest.approx(nb.cylmass(1), rel=1e-12)
        assert forces[0].Used() == nb.used(0) and forces[1].Used() == nb.used(1)
        # the combined sets of the last force evaluation
        assert np.abs(forces[0].get_coefs().reshape(-1) - want("halo", "coef")).max() <= \
            1e-10 * np.abs(want("halo", "coef")).max()
        gc = np.concatenate([x.reshape(-1) for x in forces[1].get_coefs()])
        assert np.abs(gc - want("disk", "coef")).max() <= 1e-10 * np.abs(want("disk", "coef")).max()
    # the run really is config 4: level changes in both components, >= 4 populated levels each
    assert min(total) > 0
    for name in ("halo", "disk"):
        assert (np.bincount(want(name, "level"), minlength=ms + 1) >= 30).sum() >= 4
    assert sim.time == pytest.approx((c4.NSTEPS + 1) * dtime)
    ctx.set_dense_min(-1)
    ctx.set_mover_list_min(8192)
    ctx.set_thin_max(8192)


def test_config4_on_one_stream(ctx, oracle, monkeypatch):
    """EXP_AMD_SIM_OVERLAP=0 (read when a step driver starts a call): both components' chains on the context's one stream
    instead of one stream each -- the schedule a run with more than one rank or an Orient takes anyway.  Same bars."""
    monkeypatch.setenv("EXP_AMD_SIM_OVERLAP", "0")
    try:
        test_config4_against_the_nbody_oracle(ctx, oracle, 100, 8192, 40)
    finally:
        monkeypatch.delenv("EXP_AMD_SIM_OVERLAP")


def test_config4_against_the_golden_file(ctx):
    """No oracle at run time: the committed vector (inputs + state after begin_run and after two
    master steps) alone."""
    z = c4.load_golden()
    ms = int(z["multistep"])
    sim, forces, comps = _device_run(ctx, z, ms, float(z["dtime"]), list(z["dynfrac"]))
    _compare("init", forces, comps, lambda name, key: z[f"init_{name}_{key}"], ms)
    sim.step(int(z["nsteps"]))
    _compare("golden", forces, comps, lambda name, key: z[f"{name}_{key}"], ms)
    assert forces[1].cylmass == pytest.approx(float(z["disk_cylmass"]), rel=1e-12)
    assert sim.step_switches == int(np.sum(z["nswitch"]))


def test_config4_full_size_level_sets_add_up(ctx):
    """BASELINE config 4 at its full size (1e7 halo + 1e7 disk particles, SphericalSL lmax 6 nmax 18 and
    EmpCylSL mmax 6 nmax 12 on the 256 x 128 grid -- the helper basis of the table build reduced, table
    accuracy does not enter --, multistep 4, both cross forces), through a size-independent identity of the
    block-multistep bookkeeping instead of the oracle: at the end of a master step every level has been
    accumulated at the new time, and every particle that changed level on the way has had its contribution
    moved with it (multistep_update, src/SphericalBasis.cc:1156-1228, src/CylEXP.cc:159-188) -- so the SUM of
    the per-level coefficient sets must equal a from-scratch, single-level accumulation of the final
    positions.  Sparse levels, deferred partition, level-fused sub-steps and the differencing all have to be
    right for that.  Plus: the combined set is that sum, level populations add up, several levels are in use."""
    from exp_amd.empcyl import build_empcyl
    from exp_amd.models import NFWModel, sample_disk, sample_sphere
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    from exp_amd.slgrid import build_slgrid
    n, ms, a, h = 10_000_000, 4, 0.01, 0.001
    model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
    g = build_slgrid(model, 6, 18, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
    cg = build_empcyl(mmax=6, norder=12, numx=256, numy=128, acyl=a, hcyl=h, lmaxfid=16, nmaxfid=12,
                      numr=800, rnum=100, tnum=40)
    scale = 0.1                                   # disk scale length / halo scale radius, as in bench.py
    hm, hpos, hvel = sample_sphere(model, n, seed=23)
    hpos, hvel = hpos * scale, hvel * np.sqrt(1.0 / scale)
    dm, dpos, dvel = sample_disk(n, 29, a=a, h=h, mass=0.1)
    dvel = dvel + 0.3 * np.random.default_rng(3).standard_normal(dvel.shape)
    kw = dict(scale=scale, rmin=g.rmin * scale, rmax=g.rmax * scale)
    f1, f2 = SphereSL(ctx, g, multistep=ms, **kw), Cylinder(ctx, cg, multistep=ms)
    c1, c2 = Component.from_arrays(ctx, hm, hpos, hvel), Component.from_arrays(ctx, dm, dpos, dvel)
    del hpos, hvel, dpos, dvel
    sim = Simulation(ctx, 4e-4, multistep=ms)
    i1, i2 = sim.add_component(c1, f1), sim.add_component(c2, f2)
    sim.add_interaction(i1, i2)
    sim.add_interaction(i2, i1)
    sim.init()
    sim.step(2)
    assert sim.step_switches > 0
    for c, f, fresh in ((c1, f1, lambda: SphereSL(ctx, g, **kw)), (c2, f2, lambda: Cylinder(ctx, cg))):
        lev = c.download_levels()
        pop = np.bincount(lev, minlength=ms + 1)
        assert pop.sum() == n and lev.max() <= ms and (pop > 0).sum() >= 3, pop
        flat = lambda x: np.ravel(x) if isinstance(x, np.ndarray) else np.concatenate([np.ravel(y) for y in x])
        total = np.sum([flat(f.get_coefs(level=M)) for M in range(ms + 1)], axis=0)
        comb = flat(f.get_coefs())
        cmax = np.abs(total).max()
        assert np.abs(comb - total).max() <= 1e-12 * cmax
        out = c.download(("mass", "pos"))
        ff = fresh()
        cc = Component.from_arrays(ctx, out["mass"], out["pos"])
        ff.determine_coefficients(cc)
        ref = flat(ff.get_coefs())
        assert np.abs(total - ref).max() <= 1e-9 * cmax, (type(f).__name__, np.abs(total - ref).max() / cmax)
        # Used() of a multistep force counts the first sub-step's accumulations (src/SphericalBasis.cc:797, 861),
        # i.e. the positions one master step earlier: the same up to the few particles that crossed rmax since
        assert abs(f.Used() - ff.Used()) <= 100
        cc.close(); ff.close()
    sim.close()
    for o in (c1, c2, f1, f2):
        o.close()


def test_three_components_deterministic_multistep(ctx, oracle):
    """Three components (halo, disk, a second spherical component) with the halo's force acting on BOTH others
    and both acting back on the halo, block multistep, deterministic mode: the cross forces of ONE source on two
    targets must not run on two streams at once (they share the scratch of its force pass; the two-stream
    sub-steps are for exactly two components, exp_amd/csrc/host.hip:overlap_begin).  Against the n-body
    oracle to the usual bars, and two runs bit for bit."""
    from exp_amd.models import PlummerModel, sample_sphere
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    from tests.oracle_lib import NBodyOracle
    z = c4.load_golden()
    ms, dtime, dyn = c4.MULTISTEP, c4.DTIME, c4.DYN
    g, cg = c4.grids()
    sc = float(z["scale"])
    win = c4.sph_window(g, sc)
    bm, bpos, bvel = sample_sphere(PlummerModel(1.0, 1.0, 1e-3, 50.0), 400, seed=77, rlim=30.0)
    bm, bpos, bvel = bm * 0.3, bpos * (0.5 * sc), bvel / np.sqrt(0.5 * sc) * np.sqrt(0.3)
    parts = [(z["halo_mass"], z["halo_pos"], z["halo_vel"]), (z["disk_mass"], z["disk_pos"], z["disk_vel"]),
             (bm, bpos, bvel)]
    inter = [(0, 1), (0, 2), (1, 0), (2, 0), (2, 1)]

    nb = NBodyOracle(oracle, ms, dtime, dyn)
    nb.add_sphere(g, oracle.params(**win), *parts[0])
    nb.add_cylinder(cg, *parts[1])
    nb.add_sphere(g, oracle.params(**win), *parts[2])
    for a, b in inter:
        nb.add_interaction(a, b)
    nb.init()
    nsw = [nb.step() for _ in range(2)]
    assert sum(sum(v) for v in nsw) > 0

    def device():
        forces = [SphereSL(ctx, g, multistep=ms, **win), Cylinder(ctx, cg, multistep=ms),
                  SphereSL(ctx, g, multistep=ms, **win)]
        comps = [Component.from_arrays(ctx, *p) for p in parts]
        sim = Simulation(ctx, dtime, multistep=ms, dynfrac=dyn, shiftlevl=0)
        for c, f in zip(comps, forces):
            sim.add_component(c, f)
        for a, b in inter:
            sim.add_interaction(a, b)
        sim.init()
        sim.step(2)
        out = [(c.download(), c.download_levels()) for c in comps]
        sim.close()
        for o in comps + forces:
            o.close()
        return out

    ctx.set_deterministic(True)
    try:
        first, second = device(), device()
    finally:
        ctx.set_deterministic(False)
    for k, ((o, lev), (o2, lev2)) in enumerate(zip(first, second)):
        st = nb.state[k]
        assert np.array_equal(lev, st["level"]), k
        p = np.stack([st[q] for q in "xyz"], 1)
        a = np.stack([st["a" + q] for q in "xyz"], 1)
        assert np.abs(o["pos"] - p).max() <= 1e-11, k
        assert np.abs(o["acc"] - a).max() <= 1e-9 * np.linalg.norm(a, axis=1).max(), k
        assert np.abs(o["pot"] - st["pot"]).max() <= 1e-9 * np.abs(st["pot"]).max(), k
        for key in ("pos", "vel", "acc", "pot"):
            assert np.array_equal(o[key], o2[key]), (k, key)          # bit for bit, run to run
        assert np.array_equal(lev, lev2)


@pytest.mark.parametrize("thin_v", ["1", "2"])
def test_thin_kernels_reproduce_the_table_path(ctx, monkeypatch, thin_v):
    """The direct kernels of thin active sets (k_*_acc_thin, k_*_force_thin; "2": their any-order formulation, which the
    run-time-order kernels use -- selected here by EXP_AMD_SPH_GENERIC / EXP_AMD_CYL_GENERIC, read when a force is created)
    against the moment / projected-table path ON THE DEVICE: the same run with thin_max 0 and 16384 -- levels identical,
    per-level sets, trajectories and accelerations to 1e-12 (the force kernels project the very rows the table would hold;
    the sums differ in order only).  The bars against the oracle (1e-10 / 1e-9) are the parametrised test above."""
    if thin_v == "2":
        monkeypatch.setenv("EXP_AMD_SPH_GENERIC", "1")
        monkeypatch.setenv("EXP_AMD_CYL_GENERIC", "1")
    z = c4.load_golden()
    ms = int(z["multistep"])
    runs = []
    for tm in (0, 16384):
        ctx.set_thin_max(tm)
        sim, forces, comps = _device_run(ctx, z, ms, float(z["dtime"]), list(z["dynfrac"]))
        sim.step(2)
        out = []
        for f, c in zip(forces, comps):
            sets = [f.get_coefs(level=M) for M in range(ms + 1)]
            sets = [np.concatenate([np.asarray(x).reshape(-1) for x in (s if isinstance(s, tuple) else (s,))]) for s in sets]
            out.append((c.download_levels(), c.download(), np.stack(sets)))
        runs.append(out)
        sim.close()
        for o in list(forces) + list(comps):
            o.close()
    ctx.set_thin_max(8192)
    for (l0, d0, s0), (l1, d1, s1) in zip(*runs):
        assert np.array_equal(l0, l1)
        assert np.abs(s0 - s1).max() <= 1e-12 * np.abs(s0).max()
        assert np.abs(d0["pos"] - d1["pos"]).max() <= 1e-13
        for k in ("vel", "acc", "pot"):
            assert np.abs(d0[k] - d1[k]).max() <= 1e-11 * np.abs(d0[k]).max(), k


def test_keys_of_the_closing_sweep_are_dropped_when_the_store_is_touched(ctx):
    """The sweep that closes a master step leaves the sort keys of the next one's sub-step 0 (k_kick_adjust with the force's
    key function); they are this force's, for this smallest step and this centre, and any call that touches the store
    drops them.  Three runs from the same state: master steps one after the other; the accelerations uploaded again as they
    are (touches the store, changes nothing) between them; the centre moved and moved back between them.  Same levels, same state."""
    from exp_amd.runtime import Component
    z = c4.config4_inputs(n_halo=2000, n_disk=2000)
    outs = []
    for mode in ("plain", "touch", "center"):
        sim, fs, cs = _device_run(ctx, z, c4.MULTISTEP, c4.DTIME, c4.DYN)
        for k in range(3):
            sim.step(1)
            if mode == "touch":
                for c in cs:
                    st = c.download(("acc", "pot"))
                    c.upload_acc(st["acc"], st["pot"])
            elif mode == "center":
                for c in cs:
                    c.set_center(np.array([1e-3, -2e-3, 5e-4]))
                    c.set_center(np.zeros(3))
        outs.append([(c.download_levels(), c.download()) for c in cs])
        sim.close()
        for x in cs + fs:
            x.close()
    for other in outs[1:]:
        for (lev0, st0), (lev1, st1) in zip(outs[0], other):
            assert np.array_equal(lev0, lev1)
            for key in ("pos", "vel", "acc"):
                assert np.abs(st0[key] - st1[key]).max() <= 1e-10 * max(np.abs(st0[key]).max(), 1e-300), key
