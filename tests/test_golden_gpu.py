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
