"""Parity of the HIP spherical path against the CPU oracle (through the C ABI).  GPU only.

Tolerances (fp64): coefficients  max|c_gpu - c_cpu| / max|c_cpu| <= 1e-10  (north_star);
accelerations / potential  max_p |a_gpu - a_cpu| / (|a_cpu| + tiny) <= 1e-9.  The GPU path
re-associates the n-contraction (see exp_amd/csrc/sph_kernels.h), so agreement is to rounding,
not bitwise."""
import math

import os

import numpy as np
import pytest

from tests.conftest import make_grid

pytestmark = pytest.mark.gpu

COEF_TOL = 1e-10
ACC_TOL = 1e-9


@pytest.fixture(scope="module")
def ctx():
    from exp_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def coef_err(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def acc_err(a, b):
    na = np.linalg.norm(a - b, axis=1)
    nb = np.linalg.norm(b, axis=1)
    return (na / (nb + 1e-300)).max()


def _particles(model, n, seed, squash=0.7):
    from exp_amd.models import sample_sphere
    m, pos, vel = sample_sphere(model, n, seed=seed)
    pos[:, 2] *= squash              # break spherical symmetry so every (l,m) row is exercised
    pos[:, 0] += 0.05
    return m, pos, vel


def _run(ctx, g, m, pos, **kw):
    from exp_amd.runtime import Component, SphereSL
    f = SphereSL(ctx, g, **kw)
    c = Component.from_arrays(ctx, m, pos)
    f.determine_coefficients(c)
    coef = f.get_coefs()
    used = f.Used()
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot", "pos"))
    c.close()
    f.close()
    return coef, used, out


@pytest.mark.parametrize("kind,lmax,nmax,n", [("plummer", 6, 18, 20000), ("nfw", 6, 18, 20000),
                                              ("plummer", 10, 24, 6000), ("plummer", 2, 10, 3000),
                                              ("plummer", 0, 5, 1000), ("plummer", 12, 6, 2000),
                                              ("plummer_log", 4, 8, 3000)])
def test_coefficients_and_accel_match_oracle(ctx, oracle, kind, lmax, nmax, n):
    model, g = make_grid(kind, lmax, nmax, 800 if lmax <= 6 else 400)
    m, pos, _ = _particles(model, n, seed=100 + lmax)
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    c_ref, used_ref = oracle.sph_accumulate(g, prm, pos, m)
    a_ref, p_ref = oracle.sph_accel(g, prm, pos, c_ref)
    coef, used, out = _run(ctx, g, m, pos)
    assert used == used_ref
    assert np.array_equal(out["pos"], pos)                 # caller order preserved
    assert coef_err(coef, c_ref) <= COEF_TOL
    assert acc_err(out["acc"], a_ref) <= ACC_TOL
    assert np.abs(out["pot"] - p_ref).max() / np.abs(p_ref).max() <= ACC_TOL


def test_edge_particles(ctx, oracle, plummer_s6):
    """origin, z axis, inside rmin, outside rmax (multipole continuation), first/last cell."""
    model, g = plummer_s6
    m, pos, _ = _particles(model, 2000, seed=4)
    extra = np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.5], [0.0, 0.0, -2.0], [1e-4, 0.0, 0.0],
                      [2e-4, 1e-4, -1e-4], [60.0, 1.0, 2.0], [0.0, 70.0, 0.0], [-80.0, 5.0, -100.0],
                      [g.rmax * (1 - 1e-12), 0.0, 0.0], [0.0, g.rmin * (1 + 1e-9), 0.0],
                      [1.2e-3, 0.0, 0.0], [-3.0, 0.0, 0.0], [0.0, -3.0, 0.0]])
    pos = np.concatenate([pos, extra])
    m = np.concatenate([m, np.full(len(extra), m[0])])
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    c_ref, used_ref = oracle.sph_accumulate(g, prm, pos, m)
    a_ref, p_ref = oracle.sph_accel(g, prm, pos, c_ref)
    coef, used, out = _run(ctx, g, m, pos)
    assert used == used_ref
    assert coef_err(coef, c_ref) <= COEF_TOL
    # the particle at the origin has |a| ~ 0/0 structure: compare absolutely there
    scale = np.linalg.norm(a_ref, axis=1).max()
    assert np.abs(out["acc"] - a_ref).max() <= 1e-9 * scale
    assert np.abs(out["pot"] - p_ref).max() <= 1e-9 * np.abs(p_ref).max()


@pytest.mark.parametrize("flags", [dict(NO_L0=True), dict(NO_L1=True), dict(EVEN_L=True),
                                   dict(EVEN_M=True), dict(M0_only=True),
                                   dict(EVEN_L=True, EVEN_M=True, NO_L1=True)])
def test_flags(ctx, oracle, plummer_small, flags):
    model, g = plummer_small
    m, pos, _ = _particles(model, 3000, seed=6)
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax, **flags)
    c_ref, _ = oracle.sph_accumulate(g, prm, pos, m)
    a_ref, p_ref = oracle.sph_accel(g, prm, pos, c_ref)
    coef, _, out = _run(ctx, g, m, pos, **flags)
    assert coef_err(coef, c_ref) <= COEF_TOL
    assert acc_err(out["acc"], a_ref) <= ACC_TOL
    assert np.abs(out["pot"] - p_ref).max() <= ACC_TOL * np.abs(p_ref).max()


def test_scale_center_and_window(ctx, oracle, plummer_small):
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_small
    m, pos, _ = _particles(model, 4000, seed=7)
    ctr = np.array([0.3, -0.2, 0.1])
    scale = 1.7
    prm = oracle.params(scale=scale, rmin=0.05, rmax=20.0)
    c_ref, used_ref = oracle.sph_accumulate(g, prm, pos, m, center=ctr)
    a_ref, p_ref = oracle.sph_accel(g, prm, pos, c_ref, center=ctr)
    f = SphereSL(ctx, g, scale=scale, rmin=0.05, rmax=20.0)
    c = Component.from_arrays(ctx, m, pos)
    c.set_center(ctr)
    f.determine_coefficients(c)
    assert f.Used() == used_ref
    assert coef_err(f.get_coefs(), c_ref) <= COEF_TOL
    c.zero_acceleration()
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot"))
    assert acc_err(out["acc"], a_ref) <= ACC_TOL
    assert np.abs(out["pot"] - p_ref).max() <= ACC_TOL * np.abs(p_ref).max()


def test_empty_and_tiny_components(ctx, oracle, plummer_small):
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_small
    f = SphereSL(ctx, g)
    c0 = Component(ctx, 0)
    f.determine_coefficients(c0)
    assert np.all(f.get_coefs() == 0.0) and f.Used() == 0
    f.get_acceleration_and_potential(c0)
    # one particle
    pos = np.array([[0.3, 0.2, -0.4]])
    m = np.array([1.0])
    c1 = Component.from_arrays(ctx, m, pos)
    f.determine_coefficients(c1)
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    c_ref, _ = oracle.sph_accumulate(g, prm, pos, m)
    assert coef_err(f.get_coefs(), c_ref) <= COEF_TOL


def test_set_coefs_roundtrip_and_external_target(ctx, oracle, plummer_small):
    """set_coefs -> force on ANOTHER component's particles (SetExternal, src/PotAccel.H:215)."""
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_small
    rng = np.random.default_rng(1)
    coef = rng.standard_normal(((g.lmax + 1) ** 2, g.nmax))
    f = SphereSL(ctx, g)
    f.set_coefs(coef)
    assert np.array_equal(f.get_coefs(), coef)
    pos = rng.standard_normal((5000, 3)) * np.array([2.0, 1.0, 0.3])     # unsorted, disk-like
    tgt = Component.from_arrays(ctx, np.ones(5000), pos)
    acc0 = rng.standard_normal((5000, 3))
    pot0 = rng.standard_normal(5000)
    tgt.upload_acc(acc0, pot0)                                           # acc += semantics
    f.get_acceleration_and_potential(tgt, external=True)
    out = tgt.download(("acc", "pot"))
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    a_ref, p_ref = oracle.sph_accel(g, prm, pos, coef)
    assert acc_err(out["acc"] - acc0, a_ref) <= 1e-8
    assert np.abs(out["pot"] - pot0 - p_ref).max() <= 1e-9 * np.abs(p_ref).max()


def _polar_ladder(g, n, rng):
    """positions at polar angles from 1e-9 to 1e-2 (both poles), on the axis, at the centre, beyond rmax, and regular"""
    thetas = np.array([0.0, 1e-9, 1e-8, 1e-7, 1e-6, 3e-6, 1e-5, 1e-4, 1e-3, 3e-3, 1e-2, -1.0, -2.0])
    cat = np.arange(n) % len(thetas)
    r = np.exp(rng.uniform(np.log(g.rmin * 3), np.log(g.rmax * 0.5), n))
    th = thetas[cat] * rng.uniform(0.5, 1.5, n)
    reg = cat >= len(thetas) - 2                                          # the last two classes: anywhere on the sphere
    th[reg] = np.arccos(rng.uniform(-1.0, 1.0, reg.sum()))
    ph = rng.uniform(0.0, 2.0 * np.pi, n)
    sgn = np.where(rng.uniform(size=n) < 0.5, -1.0, 1.0)
    pos = np.stack([r * np.sin(th) * np.cos(ph), r * np.sin(th) * np.sin(ph), sgn * r * np.cos(th)], 1)
    k = np.arange(n)
    ext = k % 41 == 7
    pos[ext] *= (g.rmax * 3.0 / r[ext])[:, None]                          # exterior
    pos[k % 1999 == 11] = 0.0                                            # the centre itself
    return pos


@pytest.mark.parametrize("kind,lmax,nmax,numr", [("plummer", 4, 8, 400), ("nfw", 6, 18, 2000), ("plummer_log", 4, 8, 3000),
                                                 ("plummer", 14, 4, 300)])      # (lmax 14: the any-order kernels)
@pytest.mark.parametrize("external", [True, False])
def test_polar_axis_lanes(ctx, oracle, kind, lmax, nmax, numr, external):
    """Near the polar axis the reference's own arithmetic is ill-conditioned -- sqrt((1-x)(1+x)) and 1/(x*x-1) from the
    rounded x = z/r, the m = 0 derivative as a difference of two numbers within theta^2 of each other, the clamp of
    src/Basis.cc:81-84 -- and parity is with ITS values: lanes with sin^2(theta) < 1e-5 take the general pass, which forms
    r^2, x*x - 1 and the m = 0 recurrence operation for operation (sph_kernels.h: SPH_POLAR_FAC, leg0_lit_step).  Before,
    a lane at theta = 1e-6 was off by 1.7e-3 of its tangential force.  external: another component's particles go
    through k_sph_force_staged, whose special lanes are left on a work list for the general pass behind it; otherwise
    the component's own fast pass defers them.  Per-particle relative error, acc += semantics."""
    from exp_amd.runtime import Component, SphereSL
    model, g = make_grid(kind, lmax, nmax, numr)
    rng = np.random.default_rng(77)
    coef = rng.standard_normal(((g.lmax + 1) ** 2, g.nmax)) * 1e-2
    f = SphereSL(ctx, g)
    n = 13 * 600
    pos = _polar_ladder(g, n, rng)
    tgt = Component.from_arrays(ctx, np.ones(n), pos)
    if not external:
        f.determine_coefficients(tgt)                                    # the component's own cell order
    f.set_coefs(coef)
    acc0 = rng.standard_normal((n, 3)) * 1e-3
    pot0 = rng.standard_normal(n) * 1e-3
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    a_ref, p_ref = oracle.sph_accel(g, prm, pos, coef)
    scale = np.maximum(np.abs(a_ref).max(1), 1e-3 * np.abs(a_ref).max())
    outs = []
    for rep in range(2):                                                 # (twice: the two work-list counters alternate)
        tgt.upload_acc(acc0, pot0)
        f.get_acceleration_and_potential(tgt, external=external)
        outs.append(tgt.download(("acc", "pot")))
    out = outs[0]
    assert np.isfinite(out["acc"]).all() and np.isfinite(out["pot"]).all()
    err = np.abs(out["acc"] - acc0 - a_ref).max(1) / scale
    assert err.max() <= 1e-9, (err.argmax(), pos[err.argmax()])
    assert np.abs(out["pot"] - pot0 - p_ref).max() <= 1e-10 * np.abs(p_ref).max()
    assert np.array_equal(outs[1]["acc"], out["acc"]) and np.array_equal(outs[1]["pot"], out["pot"])
    tgt.close()
    f.close()


@pytest.mark.parametrize("theta", [0.0, 1e-9, 3e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2])
def test_polar_axis_coefficients(ctx, oracle, plummer_small, theta):
    """The same near the axis for the accumulation (src/SphericalBasis.cc:486-561): the m >= 1 functions of a particle at
    polar angle theta go with sin(theta)^m, which the reference forms from the rounded cos(theta).  A few particles at
    that angle only, so that nothing else hides their share; every coefficient row against its own size."""
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_small
    rng = np.random.default_rng(int(theta * 1e9) + 5)
    n = 5
    r = np.exp(rng.uniform(np.log(0.05), np.log(5.0), n))
    th = theta * rng.uniform(0.5, 1.5, n)
    ph = rng.uniform(0.0, 2.0 * np.pi, n)
    sgn = np.where(rng.uniform(size=n) < 0.5, -1.0, 1.0)
    pos = np.stack([r * np.sin(th) * np.cos(ph), r * np.sin(th) * np.sin(ph), sgn * r * np.cos(th)], 1)
    m = rng.uniform(0.5, 1.5, n)
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m, pos)
    f.determine_coefficients(c)
    got = f.get_coefs()
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    c_ref, used = oracle.sph_accumulate(g, prm, pos, m)
    assert f.Used() == used
    assert coef_err(got, c_ref) <= COEF_TOL
    rows = np.abs(c_ref).max(1)
    big = rows > 1e-30
    assert (np.abs(got - c_ref).max(1)[big] <= 1e-9 * rows[big]).all()
    c.close()
    f.close()


def test_external_target_after_the_source_component_is_gone(ctx, oracle, plummer_small):
    """An external target is evaluated in the frame (centre) of the component the expansion was
    built from (use_external: src/SphericalBasis.cc:1509-1520).  pyEXP-style callers build the
    coefficients from a temporary component and free it: the force must keep that frame, not a
    pointer into freed memory (it once did: garbage accelerations whenever the allocator reused it)."""
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_small
    m, pos, _ = _particles(model, 8000, seed=5)
    ctr = np.array([0.4, -0.3, 0.2])
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    c_ref, _ = oracle.sph_accumulate(g, prm, pos, m)
    tpos = np.random.default_rng(2).standard_normal((3000, 3)) + ctr
    a_ref, _ = oracle.sph_accel(g, prm, tpos, c_ref, center=ctr)
    f = SphereSL(ctx, g)
    src = Component.from_arrays(ctx, m, pos + ctr)
    src.set_center(ctr)
    f.determine_coefficients(src)
    assert coef_err(f.get_coefs(), c_ref) <= COEF_TOL
    src.close()
    junk = [Component.from_arrays(ctx, np.full(k, np.nan), np.full((k, 3), np.nan)) for k in (7, 8000, 123)]
    for rep in range(3):
        tgt = Component.from_arrays(ctx, np.ones(3000), tpos)          # its own centre is the origin
        f.get_acceleration_and_potential(tgt, external=True)
        out = tgt.download(("acc",))
        tgt.close()
        assert acc_err(out["acc"], a_ref) <= 1e-8
    for j in junk:
        j.close()
    f.close()


def test_leapfrog_pieces_bit_exact(ctx, oracle):
    """incr_position / incr_velocity are single fp64 FMAs-free updates: bit-exact vs the oracle."""
    from exp_amd.runtime import Component
    rng = np.random.default_rng(2)
    n = 10007
    pos, vel, acc = rng.standard_normal((3, n, 3))
    c = Component.from_arrays(ctx, np.ones(n), pos, vel)
    c.upload_acc(acc, np.zeros(n))
    c.incr_velocity(0.0125)
    c.incr_position(0.025)
    out = c.download(("pos", "vel"))
    v = vel + acc * 0.0125
    p = pos + v * 0.025
    assert np.array_equal(out["vel"], v)
    assert np.array_equal(out["pos"], p)


@pytest.mark.parametrize("fused", [False, True])
def test_kdk_steps_match_oracle(ctx, oracle, plummer_s6, fused):
    from exp_amd.runtime import Component, SphereSL, do_step_single
    model, g = plummer_s6
    m, pos, vel = _particles(model, 8000, seed=9)
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    c_ref, _ = oracle.sph_accumulate(g, prm, pos, m)
    acc, _ = oracle.sph_accel(g, prm, pos, c_ref)
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m, pos, vel)
    c.upload_acc(acc, np.zeros(len(m)))
    dt = 0.01
    p, v, a = pos, vel, acc
    for _ in range(3):
        p, v, a, pt, cf = oracle.sph_step(g, prm, dt, p, v, a, m)
        if fused:
            f.step_kdk(c, dt)
        else:
            do_step_single(f, c, dt)
    out = c.download()
    assert coef_err(f.get_coefs(), cf) <= COEF_TOL
    assert np.abs(out["pos"] - p).max() <= 1e-12
    assert np.abs(out["vel"] - v).max() <= 1e-10
    assert acc_err(out["acc"], a) <= 1e-8
    assert np.abs(out["pot"] - pt).max() <= 1e-9 * np.abs(pt).max()
    assert np.array_equal(out["mass"], m)


def test_fused_step_key_reuse_and_invalidation(ctx, oracle, plummer_s6):
    """exp_amd_step_kdk lets the force pass write the NEXT step's sort keys.  They may only be
    used when nothing touched the component in between: a changed dt, a moved centre, new
    accelerations or an explicit kick must all fall back to the full key pass.  Every variant
    is compared with the unfused call-for-call step sequence (and so with the oracle's step)."""
    from exp_amd.runtime import Component, SphereSL, do_step_single
    model, g = plummer_s6
    m, pos, vel = _particles(model, 30000, seed=31)
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    c_ref, _ = oracle.sph_accumulate(g, prm, pos, m)
    acc, _ = oracle.sph_accel(g, prm, pos, c_ref)

    def run(kind):
        f = SphereSL(ctx, g)
        c = Component.from_arrays(ctx, m, pos, vel)
        c.upload_acc(acc, np.zeros(len(m)))
        dts = [0.01, 0.01, 0.01, 0.004, 0.004, 0.01]
        for k, dt in enumerate(dts):
            if kind == "unfused":
                do_step_single(f, c, dt)
                continue
            f.step_kdk(c, dt)
            if kind == "touched" and k == 1:          # same state written back: keys must be dropped
                o = c.download()
                c2 = Component.from_arrays(ctx, o["mass"], o["pos"], o["vel"])
                c2.upload_acc(o["acc"], o["pot"])
                c.close()
                c = c2
            if kind == "touched" and k == 3:
                c.incr_velocity(0.0)                  # a no-op kick still invalidates
        out = c.download()
        cf = f.get_coefs()
        c.close()
        f.close()
        return out, cf

    ref, cref = run("unfused")
    for kind in ("fused", "touched"):
        out, cf = run(kind)
        assert coef_err(cf, cref) <= COEF_TOL
        assert np.abs(out["pos"] - ref["pos"]).max() <= 1e-12
        assert np.abs(out["vel"] - ref["vel"]).max() <= 1e-10
        assert acc_err(out["acc"], ref["acc"]) <= 1e-8


@pytest.mark.parametrize("violent", [False, True, "tight", "lean", "lean-tight"])
def test_append_fused_step_matches_the_ordinary_one(ctx, plummer_s6, violent):
    """The APPEND form of the fused step (exp_amd_ctx_set_append_min; exp_amd/csrc/sph.hip: fused_step_append) -- no sort
    passes: the force pass places every particle in the next step's cell order, in regions with empty slots behind their
    particles -- against the ordinary fused step on the same particles, call for call: a download in the middle (the store
    is turned into an ordinary one and the mode is entered again two steps later), a change of dt and of the centre, an
    odd particle count, a non-uniform mass.  `violent`: velocities that empty half the cells within a step -- the regions
    sized from the present populations and the tail overflow, and the step is redone the ordinary way from its source.
    "lean": the payload without acceleration and potential (exp_amd_ctx_set_append_lean): what the downloads see of them is
    re-evaluated from the coefficient set kept at the completed step."""
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_s6
    n = 300_007
    m, pos, vel = _particles(model, n, seed=41)
    m = m * np.random.default_rng(1).uniform(0.5, 1.5, n)
    if violent is True:
        vel = 8.0 * pos / np.linalg.norm(pos, axis=1)[:, None] + vel
    # "tight": regions without slack (set_append_min(-n)): every pass runs out of room and the step is redone from its source
    # (after an exit the mode stays off for 8, then 16 ... fused steps -- particles.h: app_wait -- hence the long last stretch)
    dts = [0.01] * 7 + [0.004] * 4 + [0.01] * 30

    def run(app):
        ctx.set_append_min((-1000 if violent in ("tight", "lean-tight") else 1000) if app else 0)
        ctx.set_append_lean(app and violent in ("lean", "lean-tight"))
        f = SphereSL(ctx, g)
        c = Component.from_arrays(ctx, m, pos, vel)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
        snaps = []
        for k, dt in enumerate(dts):
            f.step_kdk(c, dt)
            if k == 4:
                snaps.append(c.download())                 # densify; the next steps are ordinary until the keys are back
            if k == 11:
                c.set_center([0.01, -0.02, 0.005])
        out = c.download()
        cf, used = f.get_coefs(), f.Used()
        c.close(); f.close()
        ctx.set_append_min(0)
        ctx.set_append_lean(False)
        return out, cf, used, snaps

    ref, cref, uref, sref = run(False)
    out, cf, used, snaps = run(True)
    assert used == uref
    assert coef_err(cf, cref) <= COEF_TOL
    for a, b in ((out, ref), (snaps[0], sref[0])):
        assert np.array_equal(a["mass"], b["mass"])                   # every particle is there, once, with its own mass
        assert np.abs(a["pos"] - b["pos"]).max() <= 1e-11 * (1.0 + np.abs(b["pos"]).max())
        assert np.abs(a["vel"] - b["vel"]).max() <= 1e-9 * np.abs(b["vel"]).max()
        assert acc_err(a["acc"], b["acc"]) <= 1e-8
        assert np.abs(a["pot"] - b["pot"]).max() <= 1e-9 * np.abs(b["pot"]).max()


@pytest.mark.parametrize("then", ["other_coefficients", "centre", "force_gone", "dt"])
def test_append_state_acceleration_is_the_completed_steps(ctx, plummer_s6, then):
    """With the LEAN payload (exp_amd_ctx_set_append_lean) the placing pass of the append step stores neither acceleration
    nor potential (32 of 88 bytes no pass of the next step reads): the first call that looks at the component has them re-evaluated at the positions of the completed step from the
    coefficient set KEPT at that step (exp_amd/csrc/sph.hip: sph_app_reeval) -- whatever has happened to the force since:
    its set replaced by an accumulation of other particles (which must survive the re-evaluation), the component's centre
    moved, the force destroyed, or a next step of another length (whose owed kick is formed from those accelerations)."""
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_s6
    n = 200_003
    m, pos, vel = _particles(model, n, seed=43)
    m2, pos2, _ = _particles(model, 50_000, seed=44, squash=0.5)

    def run(app):
        ctx.set_append_min(1000 if app else 0)
        ctx.set_append_lean(app)
        f = SphereSL(ctx, g)
        c = Component.from_arrays(ctx, m, pos, vel)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
        for _ in range(4):
            f.step_kdk(c, 0.01)
        ctx.profile(True); ctx.profile_reset()
        f.step_kdk(c, 0.01)
        rep = ctx.profile_report()
        ctx.profile(False)
        assert bool(rep.get("k_scatter_adv", {}).get("launches", 0)) == (not app)        # (the mode is on: no sort passes)
        other = None
        if then == "other_coefficients":
            c2 = Component.from_arrays(ctx, m2, pos2)
            f.determine_coefficients(c2)
            out = c.download()
            other = f.get_coefs()                      # still the other particles' set
            c2.close()
        elif then == "centre":
            c.set_center([0.02, -0.01, 0.03])
            out = c.download()
        elif then == "force_gone":
            f.close(); f = None
            out = c.download()
        else:
            f.step_kdk(c, 0.004)                       # leaves the mode; the half-kick owed is a * 0.005, then -a * 0.002 ...
            out = c.download()
        c.close()
        if f is not None:
            f.close()
        ctx.set_append_min(0)
        ctx.set_append_lean(False)
        return out, other

    ref, oref = run(False)
    out, other = run(True)
    assert np.array_equal(out["mass"], ref["mass"])
    assert np.abs(out["pos"] - ref["pos"]).max() <= 1e-11 * (1.0 + np.abs(ref["pos"]).max())
    assert np.abs(out["vel"] - ref["vel"]).max() <= 1e-10 * np.abs(ref["vel"]).max()
    assert acc_err(out["acc"], ref["acc"]) <= 1e-9
    assert np.abs(out["pot"] - ref["pot"]).max() <= 1e-10 * np.abs(ref["pot"]).max()
    if oref is not None:
        assert coef_err(other, oref) <= 1e-13


def test_append_step_is_the_default_from_2_to_the_20_particles(plummer_s6):
    """A context as the library makes it -- no setter called, no environment variable: `exp_amd_step_kdk` of a single-level
    spherical component of 3e6 particles leaves the sort passes out from its third step on (the APPEND form, on by default from
    2^20 particles), and the run agrees with the same run under exp_amd_ctx_set_append_min(0): every particle there once,
    coefficients to 1e-12, positions to 1e-11, with a download (the way back to an ordinary store) in the middle."""
    if os.environ.get("EXP_AMD_APPEND_MIN") or os.environ.get("EXP_AMD_APPEND_LEAN"):
        pytest.skip("the defaults are overridden by the environment")
    from exp_amd.runtime import Component, Context, SphereSL
    model, g = plummer_s6
    n = 3_000_000
    m, pos, vel = _particles(model, n, seed=47)

    def run(off):
        cx = Context(0)
        if off:
            cx.set_append_min(0)
        f = SphereSL(cx, g)
        c = Component.from_arrays(cx, m, pos, vel)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
        sorts = []
        for k in range(6):
            cx.profile(True); cx.profile_reset()
            f.step_kdk(c, 0.005)
            sorts.append(bool(cx.profile_report().get("k_scatter_adv", {}).get("launches", 0)))
            cx.profile(False)
            if k == 3:
                mid = c.download(("pos", "mass"))
        out = c.download()
        cf = f.get_coefs()
        c.close(); f.close(); cx.close()
        return out, cf, mid, sorts

    ref, cref, mref, s0 = run(True)
    out, cf, mid, s1 = run(False)
    assert all(s0)                                         # the ordinary step sorts every step
    assert s1[:2] == [True, True] and s1[2:4] == [False, False]      # entry at the second step; no sort passes afterwards
    assert coef_err(cf, cref) <= 1e-12
    assert np.array_equal(out["mass"], ref["mass"]) and np.array_equal(mid["mass"], mref["mass"])
    assert np.abs(mid["pos"] - mref["pos"]).max() <= 1e-11 * (1.0 + np.abs(mref["pos"]).max())
    assert np.abs(out["pos"] - ref["pos"]).max() <= 1e-11 * (1.0 + np.abs(ref["pos"]).max())
    assert acc_err(out["acc"], ref["acc"]) <= 1e-8


def test_split_fused_step_matches_the_unfused_sequence(ctx, oracle, plummer_s6):
    """The split fused step (two independently sorted halves, sort passes on a second stream
    overlapping accumulate / force) against the call-for-call step: same trajectory up to the order
    of the coefficient sums -- through key reuse, a dt change, a moved centre, an intervening
    download, an intervening unfused step (one global sort, then back into the split mode) and an
    odd particle count (halves of different length, ragged last block)."""
    from exp_amd.runtime import Component, SphereSL, do_step_single
    model, g = plummer_s6
    n = 70001
    m, pos, vel = _particles(model, n, seed=37)
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    c_ref, _ = oracle.sph_accumulate(g, prm, pos, m)
    acc, _ = oracle.sph_accel(g, prm, pos, c_ref)
    dts = [0.01, 0.01, 0.01, 0.004, 0.004, 0.01, 0.01, 0.01]

    def run(split):
        ctx.set_split_min(1000 if split else 0)
        f = SphereSL(ctx, g)
        c = Component.from_arrays(ctx, m, pos, vel)
        c.upload_acc(acc, np.zeros(n))
        mid = None
        for k, dt in enumerate(dts):
            if split and k == 5:
                do_step_single(f, c, dt)              # full sort: leaves the split mode
            elif split:
                f.step_kdk(c, dt)
            else:
                do_step_single(f, c, dt)
            if k == 1:
                mid = c.download(("pos", "vel"))       # applies the pending kick, keys stay
            if k == 2:
                c.set_center([0.01, -0.02, 0.005])     # keys dropped
        out = c.download()
        cf = f.get_coefs()
        used = f.Used()
        c.close(); f.close()
        ctx.set_split_min(0)
        return out, cf, mid, used

    ref, cref, mref, uref = run(False)
    out, cf, mid, used = run(True)
    assert used == uref
    assert coef_err(cf, cref) <= COEF_TOL
    for k in ("pos", "vel"):
        assert np.abs(mid[k] - mref[k]).max() <= 1e-11, k
    assert np.abs(out["pos"] - ref["pos"]).max() <= 1e-11
    assert np.abs(out["vel"] - ref["vel"]).max() <= 1e-9
    assert acc_err(out["acc"], ref["acc"]) <= 1e-8
    assert np.array_equal(out["mass"], ref["mass"])           # the id permutation survives both halves


@pytest.mark.parametrize("kind,lmax,nmax,numr,n", [("nfw", 6, 18, 2000, 1_500_000),
                                                   ("nfw", 10, 24, 2000, 1_500_000),
                                                   ("plummer", 4, 8, 400, 400_000)])
def test_fast_pass_matches_general_pass_and_oracle(ctx, oracle, kind, lmax, nmax, numr, n):
    """At these sizes nearly every wave is cell-uniform and takes the fast force pass (scalar table
    loads, shared reciprocals).  The same particles evaluated as an EXTERNAL target (not in this
    force's cell order -> general per-lane pass) and a subset on the CPU oracle must agree; so
    must the coefficients (register accumulation over many groups per flush)."""
    from exp_amd.runtime import Component, SphereSL
    model, g = make_grid(kind, lmax, nmax, numr)
    m, pos, _ = _particles(model, n, seed=21)
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m, pos)
    f.determine_coefficients(c)                      # sorts c into f's (level, cell) order
    coef = f.get_coefs()
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)              # fast pass + deferred waves
    out = c.download(("acc", "pot", "pos"))          # original particle order
    tgt = Component.from_arrays(ctx, m, pos)         # never sorted by f: general pass only
    tgt.zero_acceleration(0)
    f.get_acceleration_and_potential(tgt, external=True)
    gen = tgt.download(("acc", "pot"))
    assert acc_err(out["acc"], gen["acc"]) <= ACC_TOL
    assert np.abs(out["pot"] - gen["pot"]).max() <= 1e-10 * np.abs(gen["pot"]).max()
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    sub = np.random.default_rng(3).choice(n, 3000, replace=False)
    a_ref, p_ref = oracle.sph_accel(g, prm, pos[sub], coef)
    assert acc_err(out["acc"][sub], a_ref) <= ACC_TOL
    assert np.abs(out["pot"][sub] - p_ref).max() <= 1e-10 * np.abs(p_ref).max()
    c_ref, used = oracle.sph_accumulate(g, prm, pos[:200_000], m[:200_000])
    c2 = Component.from_arrays(ctx, m[:200_000], pos[:200_000])
    f.determine_coefficients(c2)
    assert f.Used() == used
    assert coef_err(f.get_coefs(), c_ref) <= COEF_TOL
    for x in (c, tgt, c2):
        x.close()
    f.close()


def test_full_size_properties(ctx):
    """BASELINE config 2 size (1e7, S6): linearity of the accumulation in the particle set,
    invariance to particle order, and Newton's theorem -- size-independent properties, no oracle
    needed.  (numpy sampling + host upload: the only torch GPU op of the suite would otherwise be
    this test's RNG, and the first torch kernel on a fresh box can stall for minutes.)"""
    from exp_amd.runtime import Component, SphereSL
    from exp_amd.models import sample_sphere
    model, g = make_grid("nfw", 6, 18, 2000)
    n = 10_000_000
    _, pos, _ = sample_sphere(model, n, seed=5, velocities=False)
    mass = np.full(n, 1.0 / n)
    sq = pos.copy()
    sq[:, 2] *= 0.8                       # flattened: every (l, m) row is exercised
    f = SphereSL(ctx, g)

    def coefs(p, w):
        c = Component.from_arrays(ctx, w, p)
        f.determine_coefficients(c)
        out, used = f.get_coefs(), f.Used()
        c.close()
        return out, used

    c_all, u_all = coefs(sq, mass)
    c_a, u_a = coefs(sq[:3_500_000], mass[:3_500_000])
    c_b, u_b = coefs(sq[3_500_000:], mass[3_500_000:])
    assert u_a + u_b == u_all
    assert coef_err(c_a + c_b, c_all) <= COEF_TOL
    # order invariance: a random permutation of the same particles
    perm = np.random.default_rng(7).permutation(n)
    c_p, _ = coefs(sq[perm], mass)
    assert coef_err(c_p, c_all) <= COEF_TOL
    # Newton: the self-consistent field of a SPHERICAL sample of the basis' own model is the
    # model's force, |a_r| r^2 / M(<r) = 1 (sampling noise at 1e7: ~1e-4).  This is an analytic
    # known answer for the whole chain basis -> coefficients -> force, independent of the oracle.
    c = Component.from_arrays(ctx, mass, pos)
    f.determine_coefficients(c)
    c.zero_acceleration()
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pos"))
    assert np.array_equal(out["pos"], pos)
    rr = np.linalg.norm(pos, axis=1)
    sel = (rr > 0.5) & (rr < 5.0)
    arad = -(out["acc"][sel] * pos[sel]).sum(1) / rr[sel]
    ratio = arad * rr[sel] ** 2 / model.mass(rr[sel])
    assert abs(np.median(ratio) - 1.0) < 2e-3
    c.close()
    f.close()


def test_prekicked_velocities_leave_the_trajectory_alone(ctx, oracle, plummer_s6):
    """The fused step stores velocities with the next step's opening half-kick already applied
    (exp_amd_ctx_set_prekick, include/exp_amd.h) so that the reordering pass does not read the accelerations.
    (a) positions are bit-identical to the run that leaves the closing half-kick to the next pass, velocities
    on the way out agree to an ulp; (b) a download in the middle of a run does not move the trajectory by a
    single bit; (c) a step with another dt, or any other consumer of the velocities, takes the half-kick back
    first: the oracle's KDK sequence to the usual tolerance."""
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_s6
    from exp_amd.models import sample_sphere
    m, pos, vel = sample_sphere(model, 20011, seed=91)
    # (bit-for-bit statements need the order-independent coefficient sums: atomics alone reorder)
    ctx.set_deterministic(True)
    f = SphereSL(ctx, g)
    dt = 0.01

    def run(prekick, look_at=(), steps=6, dts=None, diag_at=()):
        from exp_amd.runtime import Orient
        ctx.set_prekick(prekick)
        c = Component.from_arrays(ctx, m, pos, vel)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
        seen = []
        ori = Orient(ctx, 4, 500, Orient.CENTER | Orient.AXIS, Orient.KE) if diag_at else None
        for k in range(steps):
            f.step_kdk(c, dts[k] if dts else dt)
            if k in look_at:
                seen.append(c.download(("pos", "vel")))
            if k in diag_at:            # read-only diagnostics between steps (they read the velocities)
                seen.append(c.fix_positions())
                ori.accumulate(0.01 * (k + 1), c)
        if ori:
            ori.close()
        out = c.download(("pos", "vel", "acc"))
        c.close()
        ctx.set_prekick(os.environ.get("EXP_AMD_PREKICK", "1") != "0")
        return out, seen

    on, _ = run(True)
    off, _ = run(False)
    assert np.array_equal(on["pos"], off["pos"]) and np.array_equal(on["acc"], off["acc"])      # (a)
    ulp = np.spacing(np.abs(off["vel"]) + np.abs(off["acc"]) * dt)
    assert np.all(np.abs(on["vel"] - off["vel"]) <= ulp)
    looked, seen = run(True, look_at=(2, 3))                                                      # (b)
    assert np.array_equal(looked["pos"], on["pos"]) and np.array_equal(looked["vel"], on["vel"])
    assert len(seen) == 2 and np.isfinite(seen[0]["vel"]).all()
    # ... and neither do fix_positions / Orient::accumulate between steps: they see v - a dt/2 formed on the way,
    # the stored state and the keys of the next step stay as they are (step / diagnostic / step == step / step)
    diag, cm = run(True, diag_at=(1, 2, 4))
    for k in ("pos", "vel", "acc"):
        assert np.array_equal(diag[k], on[k]), k
    plain, cm_off = run(False, diag_at=(1, 2, 4))             # the same diagnostics on closing-kick velocities
    for a, b in zip(cm, cm_off):
        # (block sums meet in atomics: equal to rounding, not bit for bit)
        assert np.allclose(a["cov"], b["cov"], rtol=0, atol=1e-14) and np.allclose(a["com"], b["com"], rtol=0, atol=1e-14)
    # (c) changing dt from step to step: every change takes the half-kick back and redoes it
    dts = [0.01, 0.01, 0.004, 0.004, 0.02, 0.01]
    var, _ = run(True, dts=dts)
    orc_prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    p, v, a = pos.copy(), vel.copy(), None
    _, _, a, _, _ = oracle.sph_step(g, orc_prm, 0.0, p, v, np.zeros_like(p), m)
    for d in dts:
        p, v, a, _, _ = oracle.sph_step(g, orc_prm, d, p, v, a, m)
    assert np.abs(var["pos"] - p).max() <= 1e-12
    assert np.abs(var["vel"] - v).max() <= 1e-10
    f.close()
    ctx.set_deterministic(False)


def test_extrapolation_inside_rmin_with_the_logarithmic_map(ctx, oracle, monkeypatch):
    """Found by the randomised campaign (tests/fuzz/fuzz_parity.py): with cmap = 2 a radius well inside rmin is extrapolated
    over hundreds of cells (p = (xi - xi[1]) / dxi = -244 at rmin / 6 for numr 1500), and the reference's three-term radial
    derivative (p - 1/2) H[0] - 2 p H[1] + (p + 1/2) H[2] (exputil/SLGridMP2.cc:954-989) cancels to ~1e-6 of its terms for
    l = 0 (a Plummer core: the potential is flat there, the radial force a small difference of large numbers).  Against
    50-digit arithmetic on the oracle's own tables and coefficients the reference's value is itself off by more than the
    parity tolerance there, so only ITS operation order reproduces it: lanes more than four cells outside the force
    stencil take the literal per-(l, n) evaluation (sph_dp_lit, sph_kernels.h) -- 1e-9 again; with that switched off
    (EXP_AMD_NO_LITERAL) the factored table form B + p A, algebraically equal, lands 1e-8 ... 1e-5 of the particle's force away (by basis).
    (cmap = 1 compresses r -> 0 to p > -3: no such regime.)"""
    import mpmath as mp
    from exp_amd.models import PlummerModel, sample_sphere
    from exp_amd.runtime import Component, SphereSL
    from exp_amd.slgrid import build_slgrid
    model = PlummerModel(1.0, 1.0, 1e-3, 50.0)
    g = build_slgrid(model, 2, 7, numr=1500, rmin=1e-3, rmax=49.5, cmap=2, rmap=1.0, nel=24, P=6)
    m, pos, _ = sample_sphere(model, 2000, seed=5)
    pos[:, 2] *= 0.7
    probes = np.array([[1e-4, -1e-4, 1e-4], [0.98e-3 / np.sqrt(3)] * 3, [3e-4, 0.0, 0.0], [2e-5, 1e-5, -4e-5]])
    pos[:4] = probes
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    coef, _ = oracle.sph_accumulate(g, prm, pos, m)
    a_ref, p_ref = oracle.sph_accel(g, prm, pos, coef)

    def device():
        f = SphereSL(ctx, g)
        c = Component.from_arrays(ctx, m, pos)
        f.set_coefs(coef)
        c.zero_acceleration(0)
        f.get_acceleration_and_potential(c)
        out = c.download(("acc", "pot"))
        c.close(); f.close()
        return out

    out = device()
    own = np.linalg.norm(a_ref, axis=1)
    assert (np.linalg.norm(out["acc"] - a_ref, axis=1) / own).max() <= ACC_TOL          # every particle, the probes included
    assert np.abs(out["pot"] - p_ref).max() <= ACC_TOL * np.abs(p_ref).max()
    monkeypatch.setenv("EXP_AMD_NO_LITERAL", "1")
    off = device()
    monkeypatch.delenv("EXP_AMD_NO_LITERAL")
    rel = np.linalg.norm(off["acc"] - a_ref, axis=1) / own
    assert rel[4:].max() <= ACC_TOL and rel[1] <= ACC_TOL                               # 0.98 rmin and outwards: no difference
    assert rel[[0, 2, 3]].max() > 5 * ACC_TOL and rel[[0, 2, 3]].max() < 1e-3          # deep inside: the factored form drifts
    # the reference's own value against exact arithmetic on the same tables (the l = 0 radial sum of probe 0)
    mp.mp.dps = 50
    x = np.log(float(np.linalg.norm(pos[0])))
    p = (x - g.xi[1]) / g.dxi
    assert p < -100.0
    terms = [(p - 0.5) * g.ef[0, n, 0] * g.p0[0] - 2.0 * p * g.ef[0, n, 1] * g.p0[1] + (p + 0.5) * g.ef[0, n, 2] * g.p0[2]
             for n in range(g.nmax)]
    s_ref = sum(t / np.sqrt(g.ev[0, n]) * coef[0, n] for n, t in enumerate(terms))
    s_mp = sum(((mp.mpf(p) - mp.mpf(0.5)) * mp.mpf(g.ef[0, n, 0]) * mp.mpf(g.p0[0]) - 2 * mp.mpf(p) * mp.mpf(g.ef[0, n, 1]) * mp.mpf(g.p0[1])
                + (mp.mpf(p) + mp.mpf(0.5)) * mp.mpf(g.ef[0, n, 2]) * mp.mpf(g.p0[2])) / mp.sqrt(mp.mpf(g.ev[0, n])) * mp.mpf(coef[0, n])
               for n in range(g.nmax))
    assert abs(float((mp.mpf(s_ref) - s_mp) / s_mp)) > 1e-9


def test_pyexp_extrapolation_beyond_rmax_with_the_logarithmic_map(ctx, oracle):
    """The other end: pyEXP's computeAccel has no exterior branch (expui/BiorthBasis.cc:818-926), so beyond rmax the tables
    are extrapolated outwards -- with cmap = 2 over many cells again; the same literal evaluation keeps the device on
    the literal pyEXP twin of the oracle."""
    from exp_amd.models import PlummerModel, sample_sphere
    from exp_amd.runtime import Component, SphereSL
    from exp_amd.slgrid import build_slgrid
    model = PlummerModel(1.0, 1.0, 1e-3, 50.0)
    g = build_slgrid(model, 3, 6, numr=800, rmin=1e-3, rmax=49.5, cmap=2, rmap=1.0, nel=24, P=6)
    m, pos, _ = sample_sphere(model, 1500, seed=8)
    pos[:, 0] += 0.05
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    coef, _ = oracle.pyexp_sph_accumulate(g, prm, pos, m)
    test = np.concatenate([pos[:300], np.array([[70.0, 5.0, -3.0], [0.0, 120.0, 40.0], [-300.0, 10.0, 10.0], [49.6, 0.0, 0.1]])])
    a_ref = oracle.pyexp_sph_accel(g, prm, coef, test)
    f = SphereSL(ctx, g)
    assert f.lib.exp_amd_sph_set_exterior(f.h, 0) == 0
    assert f.lib.exp_amd_sph_set_dsmall(f.h, 1.0e-18) == 0                 # Spherical::computeAccel (expui/BiorthBasis.cc:824-825)
    f.set_coefs(coef)
    c = Component.from_arrays(ctx, np.full(len(test), 1.0), test)
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c, external=True)
    acc = c.download(("acc",))["acc"]
    own = np.linalg.norm(a_ref, axis=1)
    assert (np.linalg.norm(acc - a_ref, axis=1) / own).max() <= ACC_TOL
    # getFields (Spherical::crt_eval / sph_eval) at points far inside rmin and far outside rmax: every column per point
    far = np.array([[1e-4, -1e-4, 1e-4], [3e-4, 0.0, 0.0], [2e-5, 1e-5, -4e-5], [0.1, 0.2, 0.3], [70.0, 5.0, -3.0],
                    [0.0, 120.0, 40.0], [-300.0, 10.0, 10.0]])
    x, y, z = far.T
    r = np.linalg.norm(far, axis=1)
    for ctype, args in (("cartesian", (x, y, z)), ("spherical", (r, z / r, np.arctan2(y, x)))):
        got, ref = f.fields(*args, ctype), oracle.sph_fields(g, prm, coef, *args, ctype)
        assert (np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() <= ACC_TOL, ctype
    c.close(); f.close()
