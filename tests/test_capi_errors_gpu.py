"""Error paths of the C ABI through ctypes (include/exp_amd.h): every misuse returns its documented status
code, leaves a message in exp_amd_last_error and leaves the objects usable -- nothing is thrown across the
boundary, nothing crashes.  The C++ twin of these checks runs inside tests/cpp/test_potaccel.cpp.  GPU only
(contexts cannot be created without a device; the no-device failure itself is tests/test_capi_load.py)."""
import ctypes
from ctypes import byref, c_double, c_int, c_longlong, c_void_p

import numpy as np
import pytest

from tests.conftest import make_grid

pytestmark = pytest.mark.gpu

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_NODEVICE, ERR_COMM = range(6)


@pytest.fixture(scope="module")
def env():
    from exp_amd import _lib
    from exp_amd.runtime import Component, Context, SphereSL
    lib = _lib.load()
    ctx = Context(0)
    model, g = make_grid("plummer", 2, 4, 400)
    f = SphereSL(ctx, g)
    rng = np.random.default_rng(0)
    c = Component.from_arrays(ctx, np.full(100, 0.01), rng.normal(0, 1, (100, 3)))
    yield lib, ctx, f, c, g
    ctx.close()


def _msg(lib, ctx):
    m = lib.exp_amd_last_error(ctx.h)
    return m.decode() if m else ""


def test_null_handles_and_bad_arguments(env):
    lib, ctx, f, c, g = env
    ncoef = int(lib.exp_amd_force_ncoef(f.h))
    buf = np.zeros(ncoef + 8)
    p = buf.ctypes.data_as(c_void_p)
    cases = [
        ("force NULL", lambda: lib.exp_amd_force_determine_coefficients(None, c.h), ERR_ARG),
        ("comp NULL", lambda: lib.exp_amd_force_determine_coefficients(f.h, None), ERR_ARG),
        ("accel target NULL", lambda: lib.exp_amd_force_get_acceleration(f.h, None, 0), ERR_ARG),
        ("get_coefs count", lambda: lib.exp_amd_force_get_coefs(f.h, p, ncoef + 1), ERR_ARG),
        ("get_coefs NULL", lambda: lib.exp_amd_force_get_coefs(f.h, None, ncoef), ERR_ARG),
        ("set_coefs count", lambda: lib.exp_amd_force_set_coefs(f.h, p, ncoef - 1), ERR_ARG),
        ("set_level", lambda: lib.exp_amd_force_set_level(f.h, 1), ERR_ARG),
        ("set_level negative", lambda: lib.exp_amd_force_set_level(f.h, -1), ERR_ARG),
        ("level_coefs", lambda: lib.exp_amd_force_get_level_coefs(f.h, 2, 0, p, ncoef), ERR_ARG),
        ("kick level", lambda: lib.exp_amd_comp_kick(c.h, 0.1, 3), ERR_ARG),
        ("drift level", lambda: lib.exp_amd_comp_drift(c.h, 0.1, 3), ERR_ARG),
        ("zero level", lambda: lib.exp_amd_comp_zero_acc(c.h, 3), ERR_ARG),
        ("fields before density", lambda: lib.exp_amd_sph_fields(f.h, 1, p, p, p, 2, p), ERR_STATE),
        ("basis before density", lambda: lib.exp_amd_sph_basis(f.h, 1, p, p), ERR_STATE),
        ("fields coord", lambda: (lib.exp_amd_sph_set_density(f.h, np.ascontiguousarray(g.d0).ctypes.data_as(c_void_p)),
                                  lib.exp_amd_sph_fields(f.h, 1, p, p, p, 7, p))[1], ERR_ARG),
        ("cyl call on a sphere", lambda: lib.exp_amd_cyl_get_cylmass(f.h, byref(c_double())), ERR_ARG),
        ("comm rank", lambda: lib.exp_amd_comm_init_rank(ctx.h, p, 2, 2), ERR_ARG),
        ("comm world", lambda: lib.exp_amd_comm_init_rank(ctx.h, p, 0, 0), ERR_ARG),
        ("comm id NULL", lambda: lib.exp_amd_comm_init_rank(ctx.h, None, 1, 0), ERR_ARG),
        ("allreduce NULL", lambda: lib.exp_amd_comm_allreduce(ctx.h, None, 4), ERR_ARG),
        ("sim dtime", lambda: lib.exp_amd_sim_create(ctx.h, 0, 0.0, None, 0, byref(c_void_p())), ERR_ARG),
        ("sim multistep", lambda: lib.exp_amd_sim_create(ctx.h, 40, 0.1, None, 0, byref(c_void_p())), ERR_ARG),
        ("comp_create NULL", lambda: lib.exp_amd_comp_create(ctx.h, 10, None), ERR_ARG),
        ("upload_levels NULL", lambda: lib.exp_amd_comp_upload_levels(c.h, None), ERR_ARG),
        ("upload_frame NULL comp", lambda: lib.exp_amd_comp_upload_frame(None, p, p, p, p, None, None, None, 1, None, None), ERR_ARG),
        ("upload_frame NULL x", lambda: lib.exp_amd_comp_upload_frame(c.h, p, None, p, p, None, None, None, 1, None, None), ERR_ARG),
        ("upload_frame stride", lambda: lib.exp_amd_comp_upload_frame(c.h, p, p, p, p, None, None, None, 2, None, None), ERR_ARG),
        ("upload_frame columns", lambda: lib.exp_amd_comp_upload_frame(c.h, p, p, None, p, None, None, None, 1, None, None), ERR_ARG),
        ("log_sums NULL", lambda: lib.exp_amd_comp_log_sums(c.h, None), ERR_ARG),
        ("log_sums NULL comp", lambda: lib.exp_amd_comp_log_sums(None, p), ERR_ARG),
        ("get_center NULL", lambda: lib.exp_amd_comp_get_center(c.h, None), ERR_ARG),
        ("all_m on a NULL force", lambda: lib.exp_amd_sph_set_accumulate_all_m(None, 1), ERR_ARG),
    ]
    for name, call, want in cases:
        rc = call()
        assert rc == want, (name, rc, _msg(lib, ctx))
        if want != OK:
            assert (_msg(lib, ctx) or lib.exp_amd_last_global_error()), name
    # the objects survived all of that
    f.determine_coefficients(c)
    assert f.Used() > 0 and np.isfinite(f.get_coefs()).all()


def test_bad_basis_configurations_are_refused(env):
    from exp_amd._lib import CylConfig, SphConfig
    lib, ctx, f, c, g = env
    arr = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(c_void_p)
    good = dict(lmax=g.lmax, nmax=g.nmax, numr=g.numr, cmap=g.cmap, rmap=g.rmap, scale=1.0, rmin=g.rmin, rmax=g.rmax,
                xmin=g.xmin, dxi=g.dxi, NO_L0=0, NO_L1=0, EVEN_L=0, EVEN_M=0, M0_only=0, multistep=0)
    for key, val in (("lmax", -1), ("lmax", 65), ("nmax", 0), ("numr", 2), ("cmap", 5), ("multistep", 30)):
        cfg = SphConfig(**{**good, key: val})
        out = c_void_p()
        rc = lib.exp_amd_sph_create(ctx.h, byref(cfg), arr(g.xi), arr(g.p0), arr(g.ev), arr(g.ef), byref(out))
        assert rc == ERR_ARG and not out.value, (key, val, rc)
        assert key in _msg(lib, ctx) or "sph_create" in _msg(lib, ctx)
    ccfg = CylConfig(mmax=65, nmax=4, numx=8, numy=8, cmapr=1, cmapz=1, ascale=0.01, hscale=0.001, rtable=1.0,
                     xmin=0.0, dx=0.1, ymin=0.0, dy=0.1, rcylmax=20.0, EVEN_M=0, multistep=0)
    out = c_void_p()
    assert lib.exp_amd_cyl_create(ctx.h, byref(ccfg), arr(np.zeros(8)), byref(out)) == ERR_ARG and not out.value


def test_simulation_misuse(env):
    """A force of another multistep depth, interactions between unknown components."""
    from exp_amd.runtime import SphereSL
    lib, ctx, f, c, g = env
    sim = c_void_p()
    assert lib.exp_amd_sim_create(ctx.h, 2, 0.01, None, 0, byref(sim)) == OK
    idx = c_int()
    assert lib.exp_amd_sim_add_component(sim, c.h, f.h, byref(idx)) == ERR_ARG       # f has multistep 0
    assert "multistep" in _msg(lib, ctx)
    f2 = SphereSL(ctx, g, multistep=2)
    assert lib.exp_amd_sim_add_component(sim, c.h, f2.h, byref(idx)) == OK and idx.value == 0
    assert lib.exp_amd_sim_add_interaction(sim, 0, 0) == ERR_ARG
    assert lib.exp_amd_sim_add_interaction(sim, 0, 3) == ERR_ARG
    assert lib.exp_amd_force_compute_multistep_coefficients(f2.h, 99) == ERR_ARG      # mdrft beyond Mstep
    assert lib.exp_amd_sim_step(sim, -1) == ERR_ARG
    assert lib.exp_amd_sim_init(sim) == OK and lib.exp_amd_sim_step(sim, 1) == OK
    lib.exp_amd_sim_destroy(sim)
    f2.close()
