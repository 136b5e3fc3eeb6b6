"""The multi-component block-multistep oracle (oracle/nbody_oracle.c: do_step src/step.cc:67-325,
begin_run src/begin.cc:80-129, adjust_multistep_level src/multistep.cc:344-627, CylEXP's multistep
twins src/CylEXP.cc:45-282) held to what can be known without the device: it must reduce to the
single-component oracle, its level-change differencing must equal a fresh accumulation of the new
level lists, its multistep=0 step must equal the per-component pieces, and the frozen golden vector
tests/golden/config4_small.npz must not move.  CPU only."""
import numpy as np

from tests import config4_util as c4
from tests.conftest import make_grid
from tests.oracle_lib import NBodyOracle


def test_single_sphere_reduces_to_the_single_component_oracle(oracle):
    from exp_amd.models import sample_sphere
    model, g = make_grid("plummer", 4, 8, 400)
    m, pos, vel = sample_sphere(model, 1500, seed=31)
    pos[:, 2] *= 0.8
    ms, dtime = 3, 0.05
    prm = oracle.params(rmin=g.rmin, rmax=g.rmax)
    st = oracle.sph_multistep_init(g, prm, ms, dtime, c4.DYN, 0, pos, vel, m)
    nb = NBodyOracle(oracle, ms, dtime, c4.DYN)
    nb.add_sphere(g, prm, m, pos, vel)
    nb.init()
    s = nb.state[0]
    for k in range(3):
        if k:
            nsw = oracle.sph_multistep_step(g, prm, st)
            assert nb.step() == [nsw]
        assert np.array_equal(s["level"], st["level"])
        for key in ("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "pot"):
            assert np.array_equal(s[key], st[key]), (k, key)
        assert np.array_equal(s["coefN"], st["coefN"]) and np.array_equal(s["coefL"], st["coefL"])
        assert np.array_equal(s["coef"], st["coef"])
    assert len(np.unique(st["level"])) >= 3


def test_level_change_differencing_equals_fresh_accumulation(oracle):
    """After the first level assignment (all particles start on level 0 and nobody has moved), the
    differenced per-level sets must equal the accumulation of each new level list at the same
    positions: SphericalBasis::multistep_update (src/SphericalBasis.cc:1156-1228) and
    CylEXP::multistep_update (src/CylEXP.cc:159-188) against determine_coefficients_thread /
    EmpCylSL::accumulate.  (Holds because every particle here lies inside both windows.)"""
    z = c4.load_golden()
    g, cg = c4.grids()
    nb, _ = c4.oracle_run(oracle, z, pass0_only=True)
    h, d = nb.state
    prm = oracle.params(**c4.sph_window(g, float(z["scale"])))
    assert len(np.unique(h["level"])) >= 4 and len(np.unique(d["level"])) >= 4
    hp, dp = [np.stack([s["x"], s["y"], s["z"]], 1) for s in (h, d)]
    r = np.linalg.norm(hp, axis=1)
    assert r.max() < prm.rmax and r.min() >= prm.rmin
    assert np.linalg.norm(dp, axis=1).max() < cg.rtable * cg.ascale
    for M in range(c4.MULTISTEP + 1):
        sel = h["level"] == M
        ref, _ = oracle.sph_accumulate(g, prm, hp[sel], h["mass"][sel])
        assert np.abs(h["coefN"][M] - ref.reshape(-1)).max() <= 1e-12 * np.abs(h["coefN"]).max()
        sel = d["level"] == M
        cc, ss, _, _ = oracle.cyl_accumulate(cg, dp[sel], d["mass"][sel])
        ref = np.concatenate([cc.reshape(-1), ss.reshape(-1)])
        assert np.abs(d["coefN"][M] - ref).max() <= 1e-12 * np.abs(d["coefN"]).max()


def test_two_component_multistep0_equals_the_pieces(oracle):
    z = c4.load_golden()
    g, cg = c4.grids()
    prm = oracle.params(**c4.sph_window(g, float(z["scale"])))
    dt = 1e-4
    nb, _ = c4.oracle_run(oracle, z, nsteps=0, multistep=0, dtime=dt)
    h, d = nb.state

    def forces(p1, p2):
        ch, _ = oracle.sph_accumulate(g, prm, p1, z["halo_mass"])
        cc, ss, _, cmass = oracle.cyl_accumulate(cg, p2, z["disk_mass"])
        a1, q1 = oracle.sph_accel(g, prm, p1, ch)
        b1, r1 = oracle.cyl_accel(cg, p1, cc, ss, cmass)
        a2, q2 = oracle.cyl_accel(cg, p2, cc, ss, cmass)
        b2, r2 = oracle.sph_accel(g, prm, p2, ch)
        return a1 + b1, q1 + r1, a2 + b2, q2 + r2

    A1, P1, A2, P2 = forces(z["halo_pos"], z["disk_pos"])
    assert np.array_equal(np.stack([h["ax"], h["ay"], h["az"]], 1), A1) and np.array_equal(h["pot"], P1)
    # (the halo's force reaches the disk particle in two AddAcc calls per axis, so the sums associate
    # differently here: one ulp)
    assert np.abs(np.stack([d["ax"], d["ay"], d["az"]], 1) - A2).max() <= 1e-15 * np.abs(A2).max()
    assert np.abs(d["pot"] - P2).max() <= 1e-15 * np.abs(P2).max()
    nb.step()
    q1 = z["halo_pos"] + (z["halo_vel"] + A1 * (0.5 * dt)) * dt
    q2 = z["disk_pos"] + (z["disk_vel"] + A2 * (0.5 * dt)) * dt
    assert np.array_equal(np.stack([h["x"], h["y"], h["z"]], 1), q1)
    assert np.abs(np.stack([d["x"], d["y"], d["z"]], 1) - q2).max() <= 1e-18
    A1n, _, A2n, _ = forces(q1, np.stack([d["x"], d["y"], d["z"]], 1))
    assert np.abs(np.stack([d["ax"], d["ay"], d["az"]], 1) - A2n).max() <= 1e-15 * np.abs(A2n).max()
    assert np.abs(np.stack([h["ax"], h["ay"], h["az"]], 1) - A1n).max() <= 1e-15 * np.abs(A1n).max()


def test_config4_small_golden(oracle):
    """tests/golden/config4_small.npz freezes begin_run + two master steps of the small disk + halo at
    multistep 4 (inputs in the same file); the level histogram shows what it exercises."""
    z = c4.load_golden()
    nb, nsw = c4.oracle_run(oracle, z)
    snap = c4.snapshot(nb, nsw)
    assert [int(v) for v in z["nswitch"]] == nsw and min(nsw) > 0
    for name in ("halo", "disk"):
        lev = snap[name + "_level"]
        assert np.array_equal(lev, z[name + "_level"])
        assert (np.bincount(lev, minlength=5) >= 30).sum() >= 4          # >= 4 populated levels
        for key in ("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "pot", "coefN", "coefL", "coef"):
            ref = z[f"{name}_{key}"]
            assert np.abs(snap[f"{name}_{key}"] - ref).max() <= 1e-13 * np.abs(ref).max(), (name, key)
    assert snap["disk_cylmass"] == float(z["disk_cylmass"])
