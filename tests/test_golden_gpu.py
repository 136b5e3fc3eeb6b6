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
