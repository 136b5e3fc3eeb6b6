"""The HIP path against the committed golden vectors (no oracle code involved at run time)."""
import numpy as np
import pytest

from tests.golden_util import load_cyl, load_sph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from exp_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


def test_sph_golden(ctx):
    from exp_amd.runtime import Component, SphereSL
    g, z = load_sph()
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, z["mass"], z["pos"], z["vel"])
    f.determine_coefficients(c)
    assert f.Used() == int(z["used"])
    assert np.abs(f.get_coefs() - z["coef"]).max() <= 1e-10 * np.abs(z["coef"]).max()
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot"))
    scale = np.linalg.norm(z["acc"], axis=1).max()
    assert np.abs(out["acc"] - z["acc"]).max() <= 1e-9 * scale
    assert np.abs(out["pot"] - z["pot"]).max() <= 1e-9 * np.abs(z["pot"]).max()
    f.step_kdk(c, float(z["dt"]))
    o = c.download()
    assert np.abs(o["pos"] - z["step_pos"]).max() <= 1e-13
    assert np.abs(f.get_coefs() - z["step_coef"]).max() <= 1e-10 * np.abs(z["step_coef"]).max()
    assert np.abs(o["acc"] - z["step_acc"]).max() <= 1e-9 * scale


def test_cyl_golden(ctx):
    from exp_amd.runtime import Component, Cylinder
    g, z = load_cyl()
    f = Cylinder(ctx, g)
    c = Component.from_arrays(ctx, z["mass"], z["pos"])
    f.determine_coefficients(c)
    cc, ss = f.get_coefs()
    assert f.Used() == int(z["used"])
    assert f.cylmass == pytest.approx(float(z["cylmass"]), rel=1e-13)
    scale = np.abs(z["cos"]).max()
    assert np.abs(cc - z["cos"]).max() <= 1e-10 * scale
    assert np.abs(ss - z["sin"]).max() <= 1e-10 * scale
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot"))
    ok = np.isfinite(z["acc"]).all(axis=1)
    ascale = np.linalg.norm(z["acc"][ok], axis=1).max()
    assert np.abs(out["acc"][ok] - z["acc"][ok]).max() <= 1e-9 * ascale
    assert np.abs(out["pot"][ok] - z["pot"][ok]).max() <= 1e-9 * np.abs(z["pot"][ok]).max()


def test_sph_fields_and_fix_positions_golden(ctx):
    """Device getFields in the three coordinate systems and fix_positions vs the frozen vectors."""
    import os
    from exp_amd.runtime import Component, SphereSL
    g, z = load_sph()
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "sph_fields.npz"))
    f = SphereSL(ctx, g)
    f.set_coefs(z["coef"])
    f.lib.exp_amd_sph_set_exterior(f.h, 0)
    x, y, zz = gold["points"].T
    R, ph, r = np.hypot(x, y), np.arctan2(y, x), np.linalg.norm(gold["points"], axis=1)
    for key, args, ctype in (("crt", (x, y, zz), "cartesian"), ("cyl", (R, zz, ph), "cylindrical"),
                             ("sph", (r, zz / r, ph), "spherical")):
        got = f.fields(*args, ctype)
        scale = np.abs(gold[key]).max(axis=0)
        assert np.all(np.abs(got - gold[key]).max(axis=0) <= 1e-9 * scale + 1e-300)
    c = Component.from_arrays(ctx, z["mass"], z["pos"], z["vel"])
    c.upload_acc(z["acc"], z["pot"])
    got = c.fix_positions(0)
    vec = np.concatenate([[got["mtot"]], got["com"], got["cov"], got["coa"]])
    assert np.abs(vec - gold["fix_positions"]).max() <= 1e-12 * np.abs(gold["fix_positions"]).max()


def test_extras_golden(ctx):
    """Orient (selection, histories, rotations), its pseudo-acceleration fit and the sub-sample
    covariances of both bases against tests/golden/extras.npz -- no oracle code at run time."""
    import os
    from exp_amd.runtime import Component, Cylinder, Orient, SphereSL
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "extras.npz"))
    g, z = load_sph()
    pos = z["pos"].copy()
    o = Orient(ctx, 2, 120, Orient.AXIS | Orient.CENTER, Orient.KE, dT=0.0, damping=0.8)
    o.set_naccel(4)
    for k in range(4):
        c = Component.from_arrays(ctx, z["mass"], pos, z["vel"])
        c.upload_acc(np.zeros_like(pos), z["pot"])
        o.accumulate(0.1 * k, c, 0.1)
        c.close()
        st = o.state()
        got = np.array([st["Ecurr"], st["used"], *st["center"], *st["axis"], *st["center1"], *st["axis1"],
                        st["sigC"], st["sigA"]])
        ref = f["orient_states"][k]
        assert got[0] == ref[0] and got[1] == ref[1]                      # threshold energy, count: exact
        assert np.allclose(got[2:14], ref[2:14], rtol=0, atol=1e-12 * np.abs(ref[2:14]).max())
        assert np.allclose(got[14:], ref[14:], rtol=1e-6, atol=1e-24)
        pos = pos + 0.1 * z["vel"]
    assert np.abs(o.transformBody().reshape(9) - f["orient_body"]).max() <= 1e-10
    acc, om, dom = o.currentAccel()
    assert np.allclose(np.concatenate([acc, om, dom]), f["pseudo"], rtol=1e-7, atol=1e-9 * np.abs(f["pseudo"]).max())
    o.close()
    fs = SphereSL(ctx, g)
    fs.cov_enable(5)
    c = Component.from_arrays(ctx, z["mass"], z["pos"])
    fs.cov_accumulate(c, 0)
    d = fs.cov_get()
    c.close()
    assert np.array_equal(d["counts"], f["sph_cov_counts"])
    assert np.abs(d["mean"] - f["sph_cov_mean"]).max() <= 1e-10 * np.abs(f["sph_cov_mean"]).max()
    assert np.abs(d["covr"] - f["sph_cov_covr"]).max() <= 1e-10 * np.abs(f["sph_cov_covr"]).max()
    fs.close()
    cg, cz = load_cyl()
    fc = Cylinder(ctx, cg)
    fc.cov_enable(4)
    c = Component.from_arrays(ctx, cz["mass"], cz["pos"])
    fc.cov_accumulate(c)
    d = fc.cov_get()
    c.close()
    assert np.array_equal(d["counts"], f["cyl_cov_counts"])
    assert np.abs(d["mean"] - f["cyl_cov_mean"]).max() <= 1e-10 * np.abs(f["cyl_cov_mean"]).max()
    assert np.abs(d["covr"] - f["cyl_cov_covr"]).max() <= 1e-10 * np.abs(f["cyl_cov_covr"]).max()
    fc.close()
