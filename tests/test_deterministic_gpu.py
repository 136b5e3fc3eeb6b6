"""Deterministic mode (exp_amd_ctx_set_deterministic): every particle's
contribution to a coefficient sum is rounded to a fixed absolute grid first, so that all the additions
that follow -- registers, wave reductions, fp64 atomics -- are exact and the sums do not depend on the
order the hardware happens to serve them in.  Checked the hard way: the SAME particles handed over in
a different order (so that slots, waves and atomics line up differently) must give bit-identical
coefficients, and a run repeated must reproduce itself bit for bit; the answers stay within the parity
bars of the default mode.  GPU only."""
import numpy as np
import pytest

from tests.conftest import make_grid

pytestmark = pytest.mark.gpu


def _ctx(det):
    from exp_amd.runtime import Context
    c = Context(0)
    c.set_deterministic(det)
    return c


def test_sphere_sums_do_not_depend_on_particle_order(oracle):
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, SphereSL
    model, g = make_grid("plummer", 6, 18, 800)
    m, pos, vel = sample_sphere(model, 200000, seed=9)
    m = m * np.random.default_rng(1).uniform(0.5, 1.5, len(m))        # unequal masses
    perm = np.random.default_rng(2).permutation(len(m))
    ctx = _ctx(True)
    f = SphereSL(ctx, g)
    outs = []
    for order in (np.arange(len(m)), perm, perm[::-1]):
        c = Component.from_arrays(ctx, m[order], pos[order], vel[order])
        f.determine_coefficients(c)
        c0 = f.get_coefs().copy()
        c.zero_acceleration(0)
        f.get_acceleration_and_potential(c)
        for _ in range(3):
            f.step_kdk(c, 0.01)
        o = c.download()
        inv = np.argsort(order)
        outs.append((c0, f.get_coefs().copy(), o["pos"][inv], o["vel"][inv], o["acc"][inv]))
        c.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)                  # bit for bit, whatever the order
    # ... and the rounding grid costs nothing visible: the oracle at the usual bar
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    ref, _ = oracle.sph_accumulate(g, prm, pos[:20000], m[:20000])
    c = Component.from_arrays(ctx, m[:20000], pos[:20000])
    f.determine_coefficients(c)
    assert np.abs(f.get_coefs() - ref).max() <= 1e-10 * np.abs(ref).max()
    c.close(); f.close(); ctx.close()


def test_default_mode_agrees_to_rounding_only():
    """(what the mode is for: without it the same comparison holds to ~1e-15, not to the bit)"""
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, SphereSL
    model, g = make_grid("plummer", 6, 18, 800)
    m, pos, _ = sample_sphere(model, 200000, seed=9)
    perm = np.random.default_rng(2).permutation(len(m))
    ctx = _ctx(False)
    f = SphereSL(ctx, g)
    cs = []
    for order in (np.arange(len(m)), perm):
        c = Component.from_arrays(ctx, m[order], pos[order])
        f.determine_coefficients(c)
        cs.append(f.get_coefs().copy())
        c.close()
    assert np.abs(cs[0] - cs[1]).max() <= 1e-12 * np.abs(cs[0]).max()
    f.close(); ctx.close()


def test_cylinder_and_multistep_runs_reproduce_bit_for_bit():
    from tests import config4_util as c4
    from exp_amd.runtime import Component, Cylinder, Simulation, SphereSL
    z = c4.load_golden()
    g, cg = c4.grids()

    def run(order_h, order_d, dense_min, list_min=None):
        ctx = _ctx(True)
        ctx.set_dense_min(dense_min)
        if list_min is not None:
            ctx.set_mover_list_min(list_min)
        f1 = SphereSL(ctx, g, multistep=c4.MULTISTEP, **c4.sph_window(g, float(z["scale"])))
        f2 = Cylinder(ctx, cg, multistep=c4.MULTISTEP)
        c1 = Component.from_arrays(ctx, z["halo_mass"][order_h], z["halo_pos"][order_h], z["halo_vel"][order_h])
        c2 = Component.from_arrays(ctx, z["disk_mass"][order_d], z["disk_pos"][order_d], z["disk_vel"][order_d])
        sim = Simulation(ctx, c4.DTIME, multistep=c4.MULTISTEP, dynfrac=c4.DYN)
        i1, i2 = sim.add_component(c1, f1), sim.add_component(c2, f2)
        sim.add_interaction(i1, i2); sim.add_interaction(i2, i1)
        sim.init()
        sim.step(2)
        o1, o2 = c1.download(), c2.download()
        ih, idk = np.argsort(order_h), np.argsort(order_d)
        out = [o1["pos"][ih], o1["vel"][ih], o1["acc"][ih], c1.download_levels()[ih], o2["pos"][idk], o2["vel"][idk],
               o2["acc"][idk], c2.download_levels()[idk], f1.get_coefs(), *f2.get_coefs(), np.array([f2.cylmass])]
        ctx.close()
        return out

    nh, nd = len(z["halo_mass"]), len(z["disk_mass"])
    ident = (np.arange(nh), np.arange(nd))
    rng = np.random.default_rng(4)
    shuffled = (rng.permutation(nh), rng.permutation(nd))
    for dense_min in (-1, 0):                          # sparse (atomic) and cell-sorted accumulation paths
        a, b, c = run(*ident, dense_min), run(*ident, dense_min), run(*shuffled, dense_min)
        for x, y, w in zip(a, b, c):
            assert np.array_equal(x, y) and np.array_equal(x, w)
        # the other way of differencing the level changes (the list of movers through the accumulation kernels
        # instead of per-mover atomics): order-independent as well; its terms are rounded from slightly
        # different arithmetic, so it agrees with the first to rounding, not to the bit
        d, e = run(*ident, dense_min, list_min=0), run(*shuffled, dense_min, list_min=0)
        for x, y, w in zip(a, d, e):
            assert np.array_equal(y, w)
            assert np.abs(x - y).max() <= 1e-12 * max(1.0, np.abs(x).max())
    # against the frozen oracle vector, at the default mode's bars
    for name, k in (("halo", 0), ("disk", 4)):
        assert np.abs(a[k] - np.stack([z[f"{name}_x"], z[f"{name}_y"], z[f"{name}_z"]], 1)).max() <= 1e-11
        assert np.array_equal(a[k + 3], z[f"{name}_level"])
