"""The oracle reproduces the committed golden vectors bit for bit (CPU)."""
import numpy as np

from tests.golden_util import load_cyl, load_sph


def test_oracle_matches_sph_golden(oracle):
    g, z = load_sph()
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    coef, used = oracle.sph_accumulate(g, prm, z["pos"], z["mass"])
    assert used == int(z["used"])
    assert np.array_equal(coef, z["coef"])
    acc, pot = oracle.sph_accel(g, prm, z["pos"], coef)
    assert np.array_equal(acc, z["acc"]) and np.array_equal(pot, z["pot"])
    p1, v1, a1, pt1, c1 = oracle.sph_step(g, prm, float(z["dt"]), z["pos"], z["vel"], acc, z["mass"])
    assert np.array_equal(p1, z["step_pos"]) and np.array_equal(v1, z["step_vel"])
    assert np.array_equal(a1, z["step_acc"]) and np.array_equal(c1, z["step_coef"])


def test_oracle_matches_cyl_golden(oracle):
    g, z = load_cyl()
    cc, ss, used, mass = oracle.cyl_accumulate(g, z["pos"], z["mass"])
    assert used == int(z["used"]) and mass == float(z["cylmass"])
    assert np.array_equal(cc, z["cos"]) and np.array_equal(ss, z["sin"])
    acc, pot = oracle.cyl_accel(g, z["pos"], cc, ss, mass)
    assert np.array_equal(acc, z["acc"], equal_nan=True)
    assert np.array_equal(pot, z["pot"], equal_nan=True)


def test_arbiter_mode_agrees_with_plain_sum(oracle):
    """Kahan-compensated accumulation (arbiter) vs the reference-order plain sum: the oracle's own
    summation error is far below the 1e-10 parity budget."""
    g, z = load_sph()
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    c0, _ = oracle.sph_accumulate(g, prm, z["pos"], z["mass"])
    c1, _ = oracle.sph_accumulate(g, prm, z["pos"], z["mass"], kahan=True)
    assert np.abs(c0 - c1).max() <= 1e-14 * np.abs(c0).max()


def test_oracle_matches_field_golden(oracle):
    """getFields (sph_eval / cyl_eval / crt_eval) and fix_positions against the frozen vectors."""
    import os
    from tests.golden_util import load_sph
    g, z = load_sph()
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "sph_fields.npz"))
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    x, y, zz = f["points"].T
    R, ph, r = np.hypot(x, y), np.arctan2(y, x), np.linalg.norm(f["points"], axis=1)
    for key, args, ctype in (("crt", (x, y, zz), "cartesian"), ("cyl", (R, zz, ph), "cylindrical"),
                             ("sph", (r, zz / r, ph), "spherical")):
        got = oracle.sph_fields(g, prm, z["coef"], *args, ctype)
        assert np.abs(got - f[key]).max() <= 1e-13 * np.abs(f[key]).max()
    com = oracle.fix_positions(z["mass"], z["pos"], z["vel"], z["acc"],
                               np.zeros(len(z["mass"]), np.int32), 0, 0, np.zeros((1, 10)))
    assert np.abs(com - f["fix_positions"]).max() <= 1e-15 * np.abs(f["fix_positions"]).max()


def test_oracle_matches_extras_golden(oracle):
    """Orient, the pseudo-acceleration fit and the sub-sample covariances against the frozen vectors
    (tests/golden/extras.npz, make_golden.py extras)."""
    import os
    from tests.golden_util import load_cyl, load_sph
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "extras.npz"))
    g, z = load_sph()
    prm = oracle.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    o = oracle.orient(2, 120, 3, 2, 0.0, 0.8)
    rows, pos = [], z["pos"].copy()
    for k in range(4):
        oracle.orient_accumulate(o, 0.1 * k, 0.1, z["mass"], pos, z["vel"], z["pot"])
        got = np.array([o.Ecurr, o.used, *o.center[:], *o.axis[:], *o.center1[:], *o.axis1[:], o.sigC, o.sigA])
        assert np.allclose(got, f["orient_states"][k], rtol=1e-13, atol=1e-300), k
        rows.append([0.1 * k, *o.center1[:], *o.axis1[:]])
        pos = pos + 0.1 * z["vel"]
    assert np.allclose(np.array(o.body[:]), f["orient_body"], rtol=0, atol=1e-15)
    acc, om, dom = oracle.pseudo_accel_fit(np.array(rows))
    assert np.allclose(np.concatenate([acc, om, dom]), f["pseudo"], rtol=1e-10, atol=1e-300)
    cov = oracle.pyexp_sph_covariance(g, prm, z["pos"], z["mass"], 5)
    assert np.array_equal(cov["counts"], f["sph_cov_counts"]) and np.array_equal(cov["masses"], f["sph_cov_masses"])
    assert np.array_equal(cov["mean"], f["sph_cov_mean"]) and np.array_equal(cov["covr"], f["sph_cov_covr"])
    cg, cz = load_cyl()
    ccov = oracle.cyl_covariance(cg, cz["pos"], cz["mass"], 4)
    assert np.array_equal(ccov["counts"], f["cyl_cov_counts"])
    assert np.array_equal(ccov["mean"], f["cyl_cov_mean"]) and np.array_equal(ccov["covr"], f["cyl_cov_covr"])


def test_phase_space_files_written_by_the_reference():
    """tests/golden/psp_ref_f8.bin / _f4.bin: two-component PSP files whose bytes came out of the REFERENCE's own
    `Particle::writeBinary` + `ComponentHeader::write` (tests/golden/make_psp_golden.py, through
    oracle/_ref/libref_particle.so).  `PSPout` must read the inputs back, and `write_psp` must reproduce the files byte
    for byte -- wherever this runs, with or without the reference tree."""
    import io
    import os
    import tempfile
    from exp_amd import reader as R
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(here, "psp_ref_inputs.npz"))
    comps = []
    for k in range(2):
        c = {key[3:]: z[key] for key in z.files if key.startswith(f"c{k}_")}
        c["info"], c["name"], c["indexing"] = str(c["info"]), str(c["name"]), bool(c["indexing"])
        comps.append(c)
    for tag, real4 in (("f8", False), ("f4", True)):
        path = os.path.join(here, f"psp_ref_{tag}.bin")
        rd = R.ParticleReader.createReader("PSPout", [path])
        assert rd.GetTypes() == ["dark halo", "star disk"] and rd.CurrentTime() == 0.625
        assert [s.id for s in rd.stanzas] == ["sphereSL", "cylinder"] and [s.r_size for s in rd.stanzas] == [4 if real4 else 8] * 2
        f = (lambda x: np.asarray(x).astype(np.float32).astype(np.float64)) if real4 else np.asarray
        for c in comps:
            rd.SelectType(c["name"])
            a = rd.arrays()
            assert np.array_equal(a["mass"], f(c["mass"])) and np.array_equal(a["pos"], f(c["pos"])) and np.array_equal(a["vel"], f(c["vel"]))
            assert np.array_equal(a["pot"], f(c["pot"] + c["potext"]))
            assert np.array_equal(a["indx"], c["indx"] if c["indexing"] else np.arange(len(c["mass"])))
            if c["iattrib"].shape[1]:
                assert np.array_equal(a["iattrib"], c["iattrib"])
            assert np.array_equal(a["dattrib"], f(c["dattrib"]))
        with tempfile.TemporaryDirectory() as d:
            mine = os.path.join(d, "OUT.mine")
            R.write_psp(mine, 0.625, comps, real4)
            assert open(mine, "rb").read() == open(path, "rb").read()
        out = io.StringIO()
        rd.PrintSummary(stats=False, out=out)
        assert "Total particle number: 33" in out.getvalue() and "cparam :: {indexing: true, nlevel: 1}" in out.getvalue()
