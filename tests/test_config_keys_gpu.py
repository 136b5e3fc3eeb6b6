"""Every YAML key of the two n-body force methods is either honoured or refused -- none is accepted and dropped.

Walks ``SphericalBasis::valid_keys`` (src/SphericalBasis.cc:30-52) and ``Cylinder::valid_keys`` (src/Cylinder.cc:24-80)
through ``SphereSL.from_config`` / ``Cylinder.from_config`` (exp_amd/config.py) and the pyEXP key sets through
``Basis.factory``: for each key, a value that asks for something either changes what the device computes (checked
against the oracle or against the object's state) or raises.  The CPU half pins the two key tuples to the reference's own
lists where /root/reference is present."""
import os
import re

import numpy as np
import pytest

from exp_amd.config import CYLINDER_KEYS, SPHERICALBASIS_KEYS
from tests.golden_util import load_cyl

REF = "/root/reference/src"


def _ref_keys(path, cls):
    txt = open(path).read()
    m = re.search(cls + r"::valid_keys = \{(.*?)\};", txt, re.S)
    return re.findall(r'"([^"]+)"', m.group(1))


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_key_tuples_are_the_references_lists():
    assert list(SPHERICALBASIS_KEYS) == _ref_keys(os.path.join(REF, "SphericalBasis.cc"), "SphericalBasis")
    assert list(CYLINDER_KEYS) == _ref_keys(os.path.join(REF, "Cylinder.cc"), "Cylinder")


def test_pyexp_key_sets_hold_the_option_keys():
    from exp_amd.basis import CYL_KEYS, SPH_KEYS
    assert "mlim" in CYL_KEYS and "self_consistent" in CYL_KEYS and {"NO_L0", "EVEN_M", "M0_ONLY"} <= SPH_KEYS


@pytest.fixture(scope="module")
def ctx():
    from exp_amd.runtime import Context
    c = Context(0)
    yield c
    c.close()


# key -> (a value that ASKS for something, "honoured" | "refused").  The honoured ones are checked for their effect in
# the test bodies below and in tests/test_options_gpu.py / test_sph_gpu.py / test_cyl_gpu.py.
SPH_WALK = {
    "scale": (2.0, "honoured"), "rmin": (0.4, "honoured"), "rmax": (10.0, "honoured"),
    "self_consistent": (False, "honoured"), "FIX_L0": (True, "honoured"), "NO_L0": (True, "honoured"),
    "NO_L1": (True, "honoured"), "EVEN_L": (True, "honoured"), "EVEN_M": (True, "honoured"), "M0_ONLY": (True, "honoured"),
    "NOISE": (True, "refused"), "noiseN": (1e-3, "honoured"),         # (NOISE alone: refused for want of seedN; the mode itself: below)
    "noise_model_file": ("x.model", "honoured"), "seedN": (7, "honoured"),
    "ssfrac": (0.5, "honoured"), "playback": ("PLAYBACK_FILE", "honoured"), "coefCompute": (True, "refused"),   # (alone)
    "coefMaster": (False, "honoured"), "orthocheck": (True, "honoured"), "subsampleFloat": (True, "refused"),
    "totalCovar": (True, "refused"), "fullCovar": (True, "refused"),
}


@pytest.mark.gpu
def test_every_sphericalbasis_key_is_honoured_or_refused(ctx, oracle, plummer_small, tmp_path):
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, SphereSL
    model, g = plummer_small
    assert set(SPH_WALK) == set(SPHERICALBASIS_KEYS)
    m, pos, _ = sample_sphere(model, 3000, seed=2)
    pos[:, 2] *= 0.8
    base = SphereSL.from_config(ctx, g, {})
    cb = Component.from_arrays(ctx, m, pos)
    base.determine_coefficients(cb)
    cb.zero_acceleration(0)
    base.get_acceleration_and_potential(cb)
    c_base, a_base = base.get_coefs(), cb.download(("acc",))["acc"]
    with pytest.raises(ValueError, match="unmatched"):
        SphereSL.from_config(ctx, g, {"Lmax": 4})                  # (a key of Sphere, not of SphericalBasis)
    for key in SPHERICALBASIS_KEYS:
        val, what = SPH_WALK[key]
        if key == "playback":
            continue                                               # below
        if what == "refused":
            with pytest.raises(ValueError):
                SphereSL.from_config(ctx, g, {key: val})
            continue
        f = SphereSL.from_config(ctx, g, {key: val})
        c = Component.from_arrays(ctx, m, pos)
        f.determine_coefficients(c)
        c.zero_acceleration(0)
        f.get_acceleration_and_potential(c)
        coef, acc = f.get_coefs(), c.download(("acc",))["acc"]
        if key in ("scale", "rmin", "rmax", "NO_L0", "NO_L1", "EVEN_L", "EVEN_M", "M0_ONLY"):
            prm = oracle.params(**{"scale": 1.0, "rmin": g.rmin, "rmax": g.rmax,
                                   **({key.replace("M0_ONLY", "M0_only"): val})})
            c_ref, _ = oracle.sph_accumulate(g, prm, pos, m)
            a_ref, _ = oracle.sph_accel(g, prm, pos, c_ref)
            assert np.abs(coef - c_ref).max() <= 1e-10 * np.abs(c_ref).max(), key
            assert np.abs(acc - a_ref).max() <= 1e-9 * np.linalg.norm(a_ref, axis=1).max(), key
            assert np.abs(acc - a_base).max() > 1e-6 * np.abs(a_base).max(), key          # ... and it matters
        elif key == "self_consistent":
            assert f.coefs_frozen
        elif key == "ssfrac":                  # (one thread: the first half of the caller's order, masses doubled)
            with oracle.call_opts(ssfrac=val, nthrds=1):
                c_ref, used = oracle.sph_accumulate(g, oracle.params(rmin=g.rmin, rmax=g.rmax), pos, m)
            assert f.Used() == used < len(m)
            assert np.abs(coef - c_ref).max() <= 1e-10 * np.abs(c_ref).max()
            assert np.abs(coef - c_base).max() > 1e-3 * np.abs(c_base).max()              # ... and it matters
        elif key == "FIX_L0":
            c.incr_position(0.1)
            f.determine_coefficients(c)
            c.zero_acceleration(0)
            f.get_acceleration_and_potential(c)
            assert np.array_equal(f.get_coefs()[0], coef[0]) and not np.array_equal(f.get_coefs()[1], coef[1])
        elif key == "orthocheck":
            assert 0.0 <= f.orthocheck_worst < 1e-2
        else:                                                      # read only with NOISE / which rank writes the file
            assert key in ("noiseN", "noise_model_file", "seedN", "coefMaster")
            assert np.abs(coef - c_base).max() <= 1e-12 * np.abs(c_base).max()      # (same sums, atomics in arrival order)
        c.close(); f.close()
    # NOISE with its seed: every evaluation replaces the set by draws (tests/test_options_gpu.py holds them against the oracle)
    f = SphereSL.from_config(ctx, g, {"NOISE": True, "seedN": 5, "noise_model_file": os.path.join(os.path.dirname(__file__),
                                                                                                 "golden", "SLGridSph.model")})
    c = Component.from_arrays(ctx, m, pos)
    f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
    assert np.abs(f.get_coefs() - c_base).max() > 1e-3 * np.abs(c_base).max()
    c.close(); f.close()
    with pytest.raises(ValueError, match="multistep"):              # the level lists of a multistep run have no order to reproduce
        SphereSL.from_config(ctx, g, {"ssfrac": 0.5}, multistep=2)
    SphereSL.from_config(ctx, g, {"ssfrac": 1.5}, multistep=2).close()      # (not a sane value: ignored, as the reference does)
    # playback: the key names a coefficient file; from_config takes it, start_playback opens it once dtime is known
    out = tmp_path / "outcoef.halo.test"
    with open(out, "wb") as fh:
        base.dump_coefs(fh, time=0.0)
        base.dump_coefs(fh, time=1.0)
    f = SphereSL.from_config(ctx, g, {"playback": str(out), "coefCompute": True})
    f.start_playback(0.01)
    assert f.play_back and f.play_cnew
    f.determine_coefficients(cb, 0.5)
    cb.zero_acceleration(0)
    f.get_acceleration_and_potential(cb)
    assert np.abs(cb.download(("acc",))["acc"] - a_base).max() <= 1e-9 * np.abs(a_base).max()
    for o in (f, base, cb):
        o.close()


CYL_TABLE = dict(mmax=2, nmax=3, ncylnx=16, ncylny=8)
CYL_WALK = {
    # how the tables are made: honoured by building them (grid=None) or checked against the grid handed in
    **{k: (None, "table") for k in ("rcylmin", "rcylmax", "acyl", "hcyl", "nmaxfid", "lmaxfid", "mmax", "ncylnx", "ncylny",
                                    "ncylr", "nmax", "ncylodd", "rnum", "pnum", "tnum", "ashift", "cmap", "cmapr", "cmapz")},
    "mlim": (1, "honoured"), "EVEN_M": (True, "honoured"), "self_consistent": (False, "honoured"),
    "playback": ("PLAYBACK_FILE", "honoured"), "coefCompute": (True, "refused"), "coefMaster": (False, "honoured"),
    "vflag": (3, "honoured"), "density": (True, "honoured"), "override": (True, "honoured"), "try_cache": (False, "honoured"),
    "sech2": (True, "honoured"), "expcond": (True, "honoured"), "precond": (True, "honoured"),
    "tk_type": ("Hall", "refused"), "bias": (2.0, "refused"), "hexp": (2.0, "refused"), "snr": (3.0, "refused"),
    "evcut": (0.5, "refused"), "ncylrecomp": (10, "refused"), "npca": (50, "honoured"), "npca0": (10, "honoured"),
    "nvtk": (5, "refused"), "cachename": ("eof.cache", "refused"), "eof_file": ("eof.cache", "refused"),
    "samplesz": (4, "refused"), "logr": (True, "refused"), "pcavar": (True, "refused"), "pcaeof": (True, "refused"),
    "pcavtk": (True, "refused"), "pcadiag": (True, "refused"), "subsamp": (True, "refused"), "nint": (5, "refused"),
    "mtype": ("Gaussian", "refused"), "ppower": (5.0, "refused"), "pyname": ("dens.py", "refused"),
    "dumpbasis": (True, "refused"), "fullCovar": (True, "refused"), "totalCovar": (True, "refused"),
}


@pytest.mark.gpu
def test_every_cylinder_key_is_honoured_or_refused(ctx, oracle):
    from exp_amd.models import sample_disk
    from exp_amd.runtime import Component, Cylinder
    assert set(CYL_WALK) == set(CYLINDER_KEYS)
    cg, _ = load_cyl()
    m, pos, _ = sample_disk(4000, 9, a=cg.ascale, h=cg.hscale, mass=1.0)
    pos[:, 0] *= 1.3
    with pytest.raises(ValueError, match="unmatched"):
        Cylinder.from_config(ctx, {"Lmax": 4}, grid=cg)
    table_ok = dict(mmax=cg.mmax, nmax=cg.norder, ncylnx=cg.numx, ncylny=cg.numy, acyl=cg.ascale, hcyl=cg.hscale,
                    rcylmin=cg.rmin, rcylmax=cg.rmax, cmapr=cg.cmapr, cmap=cg.cmapr, cmapz=cg.cmapz)
    for key in CYLINDER_KEYS:
        val, what = CYL_WALK[key]
        if key == "playback":
            continue
        if what == "table":
            if key in table_ok:
                f = Cylinder.from_config(ctx, {key: table_ok[key]}, grid=cg)       # agrees with the tables: taken
                f.close()
                wrong = table_ok[key] * 2 if key not in ("cmapr", "cmap", "cmapz") else table_ok[key] + 1
                with pytest.raises(ValueError, match="contradicts"):
                    Cylinder.from_config(ctx, {key: wrong}, grid=cg)
            else:                                                  # how the tables were MADE: not recorded by a grid
                with pytest.raises(ValueError, match="MADE"):
                    Cylinder.from_config(ctx, {key: 12}, grid=cg)
            continue
        if what == "refused":
            with pytest.raises(ValueError):
                Cylinder.from_config(ctx, {key: val}, grid=None if key in ("cachename", "eof_file") else cg)
            continue
        f = Cylinder.from_config(ctx, {key: val}, grid=cg)
        c = Component.from_arrays(ctx, m, pos)
        f.determine_coefficients(c)
        gc, gs = f.get_coefs()
        if key == "mlim":
            with oracle.call_opts(mlim=val):
                cc, ss, _, _ = oracle.cyl_accumulate(cg, pos, m)
            assert np.abs(gc - cc).max() <= 1e-10 * np.abs(cc).max() and np.all(gc[val + 1:] == 0.0)
        elif key == "EVEN_M":
            cc, ss, _, _ = oracle.cyl_accumulate(cg, pos, m, EVEN_M=True)
            assert np.abs(gc[::2] - cc[::2]).max() <= 1e-10 * np.abs(cc).max()
        elif key == "self_consistent":
            assert f.coefs_frozen
        else:                  # cache policy, deprecated no-ops, verbosity, the defaults of the conditioning: no effect
            cc, ss, _, _ = oracle.cyl_accumulate(cg, pos, m)
            assert np.abs(gc - cc).max() <= 1e-10 * np.abs(cc).max(), key
        c.close(); f.close()
    # the defaults of features that are not built ask for nothing and are taken
    f = Cylinder.from_config(ctx, dict(pcavar=False, logr=False, nint=0, ncylrecomp=-1, precond=True, mtype="Exponential",
                                       bias=1.0, tk_type="Null", samplesz=1, nvtk=1, evcut=-1.0), grid=cg)
    f.close()
    # `expcond` is the deprecated spelling and a later `precond` overrides it (src/Cylinder.cc:492-493): this pair is valid ...
    f = Cylinder.from_config(ctx, dict(expcond=False, precond=True, npca=50), grid=cg)
    f.close()
    # ... these ask for the EOF pass over the particles (src/Cylinder.cc:960-988): a cache that reads -- grid= here -- wins as
    # it does there (`eof = cache_ok ? 0 : 1`); without one the particles must come along (condition_on=, tests/test_cyl_gpu.py)
    small = dict(mmax=2, nmax=4, ncylnx=24, ncylny=12, ncylr=400, lmaxfid=10, nmaxfid=8, ncylodd=1, acyl=0.01, hcyl=0.001)
    for eofp in (dict(precond=False), dict(expcond=False), dict(expcond=True, precond=False)):
        f = Cylinder.from_config(ctx, eofp, grid=cg)
        f.close()
        with pytest.raises(ValueError, match="PARTICLES"):
            Cylinder.from_config(ctx, dict(small, **eofp))
    with pytest.raises(ValueError, match="condition_on"):
        Cylinder.from_config(ctx, dict(small), condition_on=(m, pos))
    # grid=None: the tables are built from the keys (small orders so that it takes seconds)
    f = Cylinder.from_config(ctx, dict(mmax=2, nmax=4, ncylnx=24, ncylny=12, ncylr=400, lmaxfid=10, nmaxfid=8, rnum=40,
                                       tnum=20, ncylodd=1, acyl=0.01, hcyl=0.001, mlim=1))
    assert (f.mmax, f.nmax, f.mlim) == (2, 4, 1) and f.grid.numx == 24
    f.close()


# ---- Component::valid_keys_parm (src/Component.cc:40-95) through exp_amd.config.configure_component ------------------------------
def _ref_component_keys():
    txt = open(os.path.join(REF, "Component.cc")).read()
    m = re.search(r"Component::valid_keys_parm =\s*\{(.*?)\};", txt, re.S)
    seen, out = set(), []
    for k in re.findall(r'"([^"]+)"', m.group(1)):          # ("ctr_name" stands twice in the reference's list)
        if k not in seen:
            seen.add(k); out.append(k)
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")
def test_component_key_tuple_is_the_references_list():
    from exp_amd.config import COMPONENT_KEYS
    assert list(COMPONENT_KEYS) == _ref_component_keys()


# key -> (a value that ASKS for something, "honoured" | "refused")
COMP_WALK = {
    "name": ("halo", "honoured"), "parameters": ({}, "honoured"), "bodyfile": ("halo.bods", "honoured"), "force": ({}, "honoured"),
    "EJ": (2, "honoured"), "nEJkeep": (50, "honoured"), "nEJwant": (300, "honoured"), "nEJaccel": (4, "honoured"),
    "EJkinE": (False, "honoured"), "EJext": (True, "honoured"), "EJdiag": (True, "refused"), "EJdryrun": (True, "honoured"),
    "EJx0": (0.01, "honoured"), "EJy0": (0.02, "honoured"), "EJz0": (-0.01, "honoured"), "EJu0": (0.1, "honoured"),
    "EJv0": (0.1, "honoured"), "EJw0": (0.1, "honoured"), "EJdT": (0.05, "honoured"), "EJlinear": (True, "honoured"),
    "EJdamp": (0.5, "honoured"), "binary": (True, "honoured"), "adiabatic": (True, "refused"), "ton": (0.1, "honoured"),
    "toff": (5.0, "honoured"), "twid": (0.2, "honoured"), "rtrunc": (1.5, "honoured"), "rcom": (2.0, "honoured"),
    "consp": (True, "honoured"), "tidal": (0, "honoured"), "comlog": (True, "refused"), "bunch": (1000, "honoured"),
    "timers": (True, "refused"), "com": (True, "refused"), "indexing": (True, "honoured"), "aindex": (True, "refused"),
    "magic": (True, "honoured"), "nlevel": (10, "honoured"), "keypos": (0, "refused"), "pbufsiz": (1000, "honoured"),
    "blocking": (True, "honoured"), "ctr_name": ("disk", "refused"),            # (refused without `names`; honoured with: below)
    "buffered": (False, "honoured"),
    "noswitch": (True, "honoured"), "freezeL": (True, "honoured"), "dtreset": (False, "honoured"),
    "H5compress": (5, "honoured"), "H5shuffle": (True, "honoured"), "H5chunk": (4096, "honoured"),
}


@pytest.mark.gpu
def test_every_component_key_is_honoured_or_refused(ctx):
    """``Component::configure`` + the EJ block of ``Component::initialize`` (src/Component.cc:985-1075, :1323-1370) through
    ``exp_amd.config.configure_component``: each key alone, with a value that asks for something.  The keys that reach the
    device are checked for having arrived (the store's switches, the estimator, the driver's adiabatic factor); what they then
    DO is tests/test_options_gpu.py's (rtrunc, ton / toff / twid, noswitch / freezeL), test_multistep_gpu.py's (tidal / rcom) and
    test_orient_gpu.py's (EJ*)."""
    from exp_amd.config import COMPONENT_KEYS, configure_component
    from exp_amd.runtime import Component, Simulation, SphereSL
    from tests.conftest import make_grid
    assert set(COMP_WALK) == set(COMPONENT_KEYS)
    _, g = make_grid("plummer", 4, 8, 400)
    rng = np.random.default_rng(1)
    n = 3000
    m, pos, vel = np.full(n, 1.0 / n), rng.standard_normal((n, 3)), 0.3 * rng.standard_normal((n, 3))

    def fresh():
        sim = Simulation(ctx, 0.01, multistep=2)
        c = Component.from_arrays(ctx, m, pos, vel)
        f = SphereSL(ctx, g, multistep=2)
        return sim, c, f, sim.add_component(c, f)

    for key, (val, kind) in COMP_WALK.items():
        sim, c, f, k = fresh()
        conf = {key: val}
        if key.startswith("EJ") or key.startswith("nEJ"):
            conf.setdefault("EJ", 2)                          # (the EJ keys are only read with EJ != 0)
        if key == "ctr_name":                                  # honoured when the name is known (below), refused otherwise
            # (the source BEFORE the follower in the list: the follower then sees this call's centre -- one further down the list
            # would hand on the centre of the call before, as in the reference's loop over its components)
            sim2 = Simulation(ctx, 0.01, multistep=2)
            c2, f2 = Component.from_arrays(ctx, m, pos + 0.25, vel), SphereSL(ctx, g, multistep=2)
            c3, f3 = Component.from_arrays(ctx, m, pos, vel), SphereSL(ctx, g, multistep=2)
            k2, k3 = sim2.add_component(c2, f2), sim2.add_component(c3, f3)
            configure_component(sim2, k2, c2, {"EJ": 2, "nEJkeep": 10, "nEJwant": 200})
            configure_component(sim2, k3, c3, conf, names={"disk": k2})
            sim2.init(); sim2.step(3)
            assert np.array_equal(c3.center, c2.center) and np.abs(c2.center).max() > 0.0     # (the estimator's, handed on)
            for x in (sim2, c2, f2, c3, f3):
                x.close()
        if kind == "refused":
            with pytest.raises(ValueError, match=key):
                configure_component(sim, k, c, conf)
        else:
            o = configure_component(sim, k, c, conf)
            if "EJ" in conf:
                assert o is not None and np.allclose(o.currentCenter(), [conf.get("EJx0", 0.0), conf.get("EJy0", 0.0), conf.get("EJz0", 0.0)])
            else:
                assert o is None
            if key == "tidal":
                assert not c.escaped().any()                   # consp is on: the flags exist
            if key == "rcom" or key == "consp":
                with pytest.raises(RuntimeError):
                    c.escaped()                                # (without `tidal` nothing is tested, src/Component.cc:3317)
        sim.close(); c.close(); f.close()
    with pytest.raises(ValueError, match="unmatched"):
        sim, c, f, k = fresh()
        configure_component(sim, k, c, {"rtrunk": 1.0})
    # the keys together, as a configuration file has them: one master step runs
    sim, c, f, k = fresh()
    o = configure_component(sim, k, c, {"EJ": 2, "nEJkeep": 20, "nEJwant": 200, "EJdamp": 0.8, "rtrunc": 3.0, "tidal": 0, "rcom": 2.5,
                                        "ton": -1.0, "twid": 0.5, "noswitch": True, "dtreset": True, "indexing": True})
    sim.init()
    sim.step(1)
    assert o is not None and np.isfinite(c.fix_positions(0)["com"]).all() and c.escaped().sum() > 0
    sim.close(); c.close(); f.close()
