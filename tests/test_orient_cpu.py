"""The oracle's restatement of Orient (src/Orient.cc:325-747) against an independent numpy statement
of the same selection rule, and known answers for the regression (a centre moving linearly in time
is extrapolated exactly).  CPU only."""
import math

import numpy as np
import pytest


def _select_numpy(m, pos, vel, pot, center, many, ke):
    """The `many` most bound particles (strictly below the (many+1)-th lowest energy)."""
    E = pot + (0.5 * (vel ** 2).sum(axis=1) if ke else 0.0)
    srt = np.sort(E)
    Ecurr = srt[-1] if len(srt) <= many else srt[many]
    sel = E < Ecurr
    L = (m[sel, None] * np.cross(pos[sel] - center, vel[sel])).sum(axis=0)
    R = (m[sel, None] * pos[sel]).sum(axis=0)
    return Ecurr, int(sel.sum()), m[sel].sum(), L, R


@pytest.mark.parametrize("ke", [False, True])
@pytest.mark.parametrize("n,many", [(5000, 300), (200, 500), (1000, 999), (1000, 1000)])
def test_selection_matches_numpy(oracle, n, many, ke):
    rng = np.random.default_rng(n + many)
    m = rng.uniform(0.5, 1.5, n) / n
    pos, vel = rng.standard_normal((2, n, 3))
    pot = -1.0 / np.sqrt(1.0 + (pos ** 2).sum(axis=1)) + 0.01 * rng.standard_normal(n)
    o = oracle.orient(keep=1, many=many, oflags=3, cflags=2 if ke else 0)
    oracle.orient_accumulate(o, 0.0, 0.0, m, pos, vel, pot)
    Ecurr, used, mtot, L, R = _select_numpy(m, pos, vel, pot, np.zeros(3), many, ke)
    assert o.Ecurr == Ecurr and o.used == used
    assert o.mtot == pytest.approx(mtot, rel=1e-13)
    assert np.allclose(np.array(o.axis1[:]), L / mtot, rtol=0, atol=1e-13 * np.abs(L / mtot).max())
    assert np.allclose(np.array(o.center1[:]), R / mtot, rtol=0, atol=1e-13)
    # keep == 1: the centre is this call's mean position (src/Orient.cc:704-705)
    assert np.array_equal(np.array(o.center[:]), np.array(o.center1[:]))


def test_duplicate_energies_collapse_like_the_set(oracle):
    """std::set<EL3, ltEL3> orders on E alone: a second particle of equal energy is never stored
    (src/Orient.H:50-56, src/Orient.cc:399)."""
    n = 50
    m = np.full(n, 1.0 / n)
    pos, vel = np.random.default_rng(1).standard_normal((2, n, 3))
    pot = np.repeat(np.arange(n // 2, dtype=float), 2) - 100.0          # every energy twice
    o = oracle.orient(keep=1, many=10, oflags=3)
    oracle.orient_accumulate(o, 0.0, 0.0, m, pos, vel, pot)
    assert o.used == 10 and o.Ecurr == -90.0                              # ten DISTINCT energies below


def test_linear_drift_is_extrapolated_and_damped(oracle):
    """Histories + least squares (src/Orient.cc:557-716): for a cluster whose centre moves as
    c0 + u t the regression returns that line; with damping d it is evaluated at
    d t + (1 - d) t_oldest; the keep-weighting mixes in center0 (zero here)."""
    rng = np.random.default_rng(5)
    n, keep = 2000, 4
    m = np.full(n, 1.0 / n)
    base = 0.05 * rng.standard_normal((n, 3))
    vel = rng.standard_normal((n, 3)) * 0.01
    pot = -1.0 / np.sqrt(0.01 + (base ** 2).sum(axis=1))
    c0, u = np.array([0.3, -0.2, 0.1]), np.array([0.5, 0.25, -0.125])
    for damp in (1.0, 0.5):
        o = oracle.orient(keep=keep, many=n, oflags=2, damp=damp)
        times = [0.1 * k for k in range(10)]
        for t in times:
            oracle.orient_accumulate(o, t, 0.1, m, base + c0 + u * t, vel, pot)
            nC = o.nC
            assert nC == min(times.index(t) + 1, keep + 1)
            if nC > 1:
                sel_mean = np.array(o.center1[:]) - (c0 + u * t)         # cluster mean offset (constant)
                t_eval = damp * t + (1 - damp) * o.tC[0]
                factor = ((nC - keep) / keep) ** 2
                want = (c0 + u * t_eval + sel_mean) * (1 - factor)
                assert np.allclose(np.array(o.center[:]), want, rtol=0, atol=1e-11)
        assert o.sigC < 1e-20                                             # residual of an exact line


def test_axis_rotation_matrices(oracle):
    """AXIS branch (src/Orient.cc:566-611): a disc spinning about a tilted axis; once the history is
    long enough body/orig are the Slater-Euler matrices of the regressed axis, body . axis = z."""
    rng = np.random.default_rng(9)
    n, keep = 4000, 2
    nhat = np.array([math.sin(0.4) * math.cos(1.1), math.sin(0.4) * math.sin(1.1), math.cos(0.4)])
    e1 = np.cross(nhat, [0, 0, 1.0]); e1 /= np.linalg.norm(e1)
    e2 = np.cross(nhat, e1)
    R, ph = rng.uniform(0.1, 1.0, n), rng.uniform(0, 2 * np.pi, n)
    pos = R[:, None] * (np.cos(ph)[:, None] * e1 + np.sin(ph)[:, None] * e2)
    vel = 0.7 * (-np.sin(ph)[:, None] * e1 + np.cos(ph)[:, None] * e2)    # L along +-nhat
    m = np.full(n, 1.0 / n)
    pot = -1.0 / R
    o = oracle.orient(keep=keep, many=n // 2, oflags=3)
    for k in range(keep + 3):
        oracle.orient_accumulate(o, 0.1 * k, 0.1, m, pos, vel, pot)
    axis = np.array(o.axis[:])
    ahat = axis / np.linalg.norm(axis)
    assert abs(abs(ahat @ nhat) - 1.0) < 1e-12
    body, orig = np.array(o.body[:]).reshape(3, 3), np.array(o.orig[:]).reshape(3, 3)
    phi, theta = math.atan2(axis[1], axis[0]), -math.acos(ahat[2])
    assert np.allclose(body, oracle.euler_slater(phi, theta, 0.0, 0), atol=1e-15)
    assert np.allclose(orig, body.T, atol=1e-15) and np.allclose(body @ orig, np.eye(3), atol=1e-14)
    assert np.allclose(body @ ahat, [0, 0, 1.0], atol=1e-12)


def test_spacing_and_linear_mode(oracle):
    """Calls closer than deltaT are dropped (:423-427); set_linear just drifts the user centre (:431-435)."""
    n = 100
    m = np.full(n, 1.0 / n)
    pos, vel = np.random.default_rng(2).standard_normal((2, n, 3))
    pot = -np.ones(n) - np.arange(n) * 1e-3
    o = oracle.orient(keep=2, many=10, oflags=2, deltaT=0.5)
    oracle.orient_accumulate(o, 0.0, 0.1, m, pos, vel, pot)
    oracle.orient_accumulate(o, 0.3, 0.1, m, pos + 1.0, vel, pot)        # too soon: ignored
    assert o.nC == 1 and o.lasttime == 0.0
    oracle.orient_accumulate(o, 0.5, 0.1, m, pos + 1.0, vel, pot)
    assert o.nC == 2
    o = oracle.orient(keep=2, many=10, oflags=2)
    o.linear = 1
    o.center0[:] = (1.0, 2.0, 3.0)
    o.cenvel0[:] = (0.5, 0.0, -0.5)
    oracle.orient_accumulate(o, 0.0, 0.2, m, pos, vel, pot)
    assert list(o.center[:]) == [1.0, 2.0, 3.0] and list(o.center0[:]) == [1.1, 2.0, 2.9]


def test_quadls_and_pseudo_accel_known_answers(oracle):
    """QuadLS / PseudoAccel (include/QuadLS.H, include/PseudoAccel.H): exact on quadratics; a centre
    on c0 + u t + g t^2 / 2 has acceleration g; an axis turning at rate w about z gives omega = w z
    (to the order of the quadratic fit) and getPseudoAccel is 2 w x v + dw/dt x x + w x (w x x)."""
    t = np.linspace(0.3, 1.7, 9)
    a, b, c = -0.7, 2.5, 0.125
    assert np.allclose(oracle.quadls(t, a * t * t + b * t + c), [a, b, c], rtol=0, atol=1e-11)
    assert np.allclose(oracle.quadls(t, a * t * t + b * t + c), np.polyfit(t, a * t * t + b * t + c, 2), atol=1e-10)
    c0, u, g = np.array([0.1, -0.2, 0.3]), np.array([1.0, 0.5, -0.25]), np.array([0.02, -0.04, 0.06])
    w = 0.05
    rows = np.zeros((9, 7))
    rows[:, 0] = t
    rows[:, 1:4] = c0 + np.outer(t, u) + 0.5 * np.outer(t * t, g)
    tilt = 0.3
    rows[:, 4:7] = np.stack([np.sin(tilt) * np.cos(w * t), np.sin(tilt) * np.sin(w * t),
                             np.full_like(t, np.cos(tilt))], axis=1)
    acc, om, dom = oracle.pseudo_accel_fit(rows)
    assert np.allclose(acc, g, rtol=0, atol=1e-10)
    # n x dn/dt = w sin(tilt) (-cos(tilt) cos, -cos(tilt) sin, sin(tilt)) at the last time
    T = t[-1]
    want = w * np.sin(tilt) * np.array([-np.cos(tilt) * np.cos(w * T), -np.cos(tilt) * np.sin(w * T), np.sin(tilt)])
    assert np.allclose(om, want, rtol=0, atol=2e-5)
    pos, vel = np.random.default_rng(1).standard_normal((2, 5, 3))
    got = oracle.get_pseudo_accel(1, 1, acc, om, dom, pos, vel)
    ref = acc + 2 * np.cross(om, vel) + np.cross(dom, pos) + np.cross(om, np.cross(om, pos))
    assert np.allclose(got, ref, rtol=0, atol=1e-15)
    assert np.array_equal(oracle.get_pseudo_accel(0, 0, acc, om, dom, pos, vel), np.zeros((5, 3)))
