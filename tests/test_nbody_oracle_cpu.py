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


def test_orient_log_restart_oracle(tmp_path):
    """The oracle's restatement of Orient's log (src/Orient.cc:84-335, :742-785) on a hand-made run:
    header layout, the row format (33 columns of width 15), the time cut, the `keep` newest history
    entries, the untouched backup and the rule that a row which ends early feeds no queue entry."""
    import numpy as np
    from tests.oracle_lib import Oracle
    orc = Oracle()
    log = str(tmp_path / "disk.orient.run1")
    o = orc.orient(3, 100, 3)
    assert orc.orient_restart(o, log, True, 0.0, 0.01, 1)[0] == 0
    head = open(log).read().splitlines()
    assert len(head) == 2 and len(head[0]) == 33 * 15 and head[0][:15] == "# Time".ljust(15)
    assert head[1].startswith("# 1------------| 2------------") and head[1].endswith("| 33-----------")
    rng = np.random.default_rng(5)
    rowsv = []
    for k in range(6):
        o.Ecurr = -1.0 - 0.1 * k
        o.used = 90 + k
        for name in ("axis", "axis1", "center", "center0", "center1"):
            v = rng.normal(size=3)
            for j in range(3):
                getattr(o, name)[j] = v[j]
        acc = rng.normal(size=3)
        orc.orient_log_entry(o, log, 0.01 * k, com=(0.1, 0.2, 0.3), accel=acc)
        rowsv.append((0.01 * k, acc))
    text = open(log).read().splitlines()
    assert all(len(r) == 33 * 15 for r in text[2:])
    with open(log, "a") as f:                       # a short (18-column) row as older logs have them
        f.write("".join(f"{v:>15.6g}" for v in [0.06, -2.0, 50] + list(range(15))) + "\n")
    o2 = orc.orient(3, 100, 3)
    rows, q = orc.orient_restart(o2, log, True, 0.06, 0.01, 2, naccel=4)
    assert rows == 7 and o2.nA == 3 and o2.nC == 3 and list(o2.tA[:3]) == [0.04, 0.05, 0.06]
    assert list(o2.center1[:]) == [12.0, 13.0, 14.0] and o2.Ecurr == -2.0
    tab = np.array([r.split() for r in text[2:]], dtype=float)
    assert q.shape == (4, 7) and np.array_equal(q[:, 0], tab[2:6, 0]) and np.array_equal(q[:, 1:4], tab[2:6, 24:27])
    assert open(log + ".bak").read().splitlines()[:8] == text and len(open(log).read().splitlines()) == 7
    # the cut: rows later than tnow + 0.1 dtime/Mstep stay behind
    o3 = orc.orient(3, 100, 2)
    rows, _ = orc.orient_restart(o3, log, True, 0.0204, 0.01, 2)
    assert rows == 3 and o3.nA == 0 and o3.nC == 3
    assert np.allclose(np.array(o3.body[:]).reshape(3, 3), np.eye(3))    # CENTER only: no rotation
