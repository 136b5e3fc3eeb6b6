"""The run log on the device path: ``exp_amd_comp_log_sums`` against the oracle's restatement of OutLog::Run's particle
loop, and the reference's own acceptance criterion (tests/Halo/check.py) read off an OUTLOG file written during a
block-multistep run of the reference's test configuration."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from exp_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def test_log_sums_match_the_oracle(ctx, oracle):
    from exp_amd.runtime import Component
    rng = np.random.default_rng(31)
    for n in (1, 63, 5000, 300001):
        m = rng.uniform(0.5, 1.5, n) / n
        pos, vel, acc = rng.normal(0, 0.4, (n, 3)), rng.normal(0, 0.5, (n, 3)) + 0.1, rng.normal(0, 1.0, (n, 3))
        pot = -rng.uniform(0.5, 2.0, n)
        c = Component.from_arrays(ctx, m, pos, vel)
        c.upload_acc(acc, pot)
        got, want = c.log_sums(), oracle.outlog_sums(m, pos, vel, acc, pot)
        assert got["nbodies"] == n
        for k in ("mtot", "ektot", "eptot", "clausius"):
            assert abs(got[k] - want[k]) <= 1e-12 * abs(want[k]), (n, k)
        scale = {"com": np.abs(m[:, None] * pos).sum(), "cov": np.abs(m[:, None] * vel).sum(),
                 "angm": (m * np.linalg.norm(pos, axis=1) * np.linalg.norm(vel, axis=1)).sum()}
        for k in ("com", "cov", "angm"):
            assert np.abs(got[k] - want[k]).max() <= 1e-12 * scale[k], (n, k)
        assert np.array_equal(c.center, np.zeros(3))
        c.set_center([0.1, -0.2, 0.3])
        assert np.array_equal(c.center, [0.1, -0.2, 0.3])
        c.close()


def test_the_references_acceptance_check_on_an_outlog_file(ctx, oracle, tmp_path):
    """tests/CMakeLists.txt expNbodyTest + expNbodyCheck2TW: the configuration of tests/Halo/config.yml (10000 bodies of
    tests/Halo/SLGridSph.model, sphereSL numr 4000 / Lmax 2 / nmax 10, dtime 0.002, multistep 4, 500 steps, `outlog` with
    nint 10), the log written by exp_amd.outlog.OutLog from the device's sums, then the criterion of tests/Halo/check.py
    applied to the file.  The last row's sums are compared with the oracle's on the downloaded state."""
    from exp_amd.models import TableModel, sample_sphere
    from exp_amd.outlog import OutLog
    from exp_amd.runtime import Component, Simulation, SphereSL
    from exp_amd.slgrid import build_slgrid
    model = TableModel(os.path.join(os.path.dirname(__file__), "golden", "SLGridSph.model"))
    g = build_slgrid(model, 2, 10, numr=4000, rmin=0.0001, rmax=1.95, cmap=1, rmap=0.0667, nel=40, P=8)
    m, pos, vel = sample_sphere(model, 10000, seed=20260101, rlim=1.95)
    f = SphereSL(ctx, g, multistep=4)
    c = Component.from_arrays(ctx, m, pos, vel)
    sim = Simulation(ctx, 0.002, multistep=4, dynfrac=[1.0e32, 0.05, 1.00, 0.03, 0.05], shiftlevl=0)
    sim.add_component(c, f)
    sim.init()
    path = tmp_path / "OUTLOG.run0"
    log = OutLog(str(path), nint=10)
    log.add_component("halo", "sphereSL", c, f)
    log.run(0, 0.0)                                           # (the reference logs the initial state from begin_run)
    row = None
    for n in range(1, 501):
        sim.step(1)
        row = log.run(n, 0.002 * n, last=(n == 500)) or row
    # the last row against the oracle on the downloaded state
    o = c.download()
    want = oracle.outlog_sums(o["mass"], o["pos"], o["vel"], o["acc"], o["pot"])
    cols = [float(x) for x in row.split("|")]
    assert cols[0] == pytest.approx(1.0, rel=1e-12) and cols[2] == 10000 and cols[38] == f.Used()
    assert cols[12] == pytest.approx(want["ektot"], rel=1e-9) and cols[14] == pytest.approx(want["clausius"], rel=1e-9)
    assert cols[13] == pytest.approx(want["eptot"], rel=1e-9)
    assert cols[16] == pytest.approx(-2.0 * want["ektot"] / want["clausius"], rel=1e-9)
    # the criterion of tests/Halo/check.py: skip the six header lines, average column 17 (2T/VC) over the rows, and
    # require (mean - 1)^2 <= 0.003
    rows = [ln.split("|") for ln in open(path).read().splitlines()[6:]]
    n = 6 + len(rows)
    mean = float(np.mean([float(r[16]) for r in rows]))
    assert len(rows) == 51
    assert (mean - 1.0) ** 2 <= 0.003
    print(f"OUTLOG 2T/VC mean over {n - 6} rows: {mean:.5f}")
    c.close(); f.close()
