"""The YAML keys of EXP's n-body force methods -- and of the Component that owns one (``configure_component``, at the end) --
one by one.

``SphereSL.from_config`` / ``Cylinder.from_config`` (exp_amd/runtime.py) take the ``parameters`` block a component's
force has in an EXP configuration file.  Every key of ``SphericalBasis::valid_keys`` (src/SphericalBasis.cc:30-52) and
``Cylinder::valid_keys`` (src/Cylinder.cc:24-80) is either HONOURED -- it reaches the device path, or it is checked
against the tables that were handed in -- or REFUSED with a ``ValueError`` that says what is missing; a key outside the
two sets is refused as the reference refuses it (``unmatched()``, src/PotAccel.H:323).  Nothing is accepted and dropped:
``tests/test_config_keys_gpu.py`` walks both sets.

A key whose value is the reference's default for a feature that is not built here (``NOISE: false``, ``pcavar: false`` ...) asks for nothing and is honoured as such; only asking for the feature is refused.
"""
from __future__ import annotations

from typing import Optional

# ---- src/SphericalBasis.cc:30-52 ---------------------------------------------------------------------------------
SPHERICALBASIS_KEYS = ("scale", "rmin", "rmax", "self_consistent", "FIX_L0", "NO_L0", "NO_L1", "EVEN_L", "EVEN_M",
                       "M0_ONLY", "NOISE", "noiseN", "noise_model_file", "seedN", "ssfrac", "playback", "coefCompute",
                       "coefMaster", "orthocheck", "subsampleFloat", "totalCovar", "fullCovar")

# ---- src/Cylinder.cc:24-80 ---------------------------------------------------------------------------------------
CYLINDER_KEYS = ("tk_type", "rcylmin", "rcylmax", "acyl", "bias", "hcyl", "sech2", "hexp", "snr", "evcut", "nmaxfid",
                 "lmaxfid", "mmax", "mlim", "ncylnx", "ncylny", "ncylr", "nmax", "ncylodd", "ncylrecomp", "npca", "npca0",
                 "nvtk", "cachename", "eof_file", "override", "samplesz", "rnum", "pnum", "tnum", "ashift", "expcond",
                 "precond", "logr", "pcavar", "pcaeof", "pcavtk", "pcadiag", "subsamp", "nint", "try_cache", "density",
                 "EVEN_M", "cmap", "cmapr", "cmapz", "vflag", "mtype", "ppower", "self_consistent", "playback",
                 "coefCompute", "coefMaster", "pyname", "dumpbasis", "fullCovar", "totalCovar")


def _bool(v) -> bool:
    if isinstance(v, str):
        t = v.strip().lower()
        if t in ("true", "yes", "on", "y", "1"):
            return True
        if t in ("false", "no", "off", "n", "0"):
            return False
        raise ValueError(f"<{v}> does not convert to a boolean")
    return bool(v)


def _refuse(cls_name: str, key: str, value, why: str):
    return ValueError(f"{cls_name}: key '{key}: {value}' is not supported by this build -- {why}")


def sphere_from_config(cls, ctx, grid, conf: dict, multistep: int = 0, nthrds: int = 1):
    """``nthrds``: the global thread count of the run (src/global.cc: nthrds) -- only ``ssfrac`` depends on it."""
    conf = dict(conf or {})
    bad = sorted(set(conf) - set(SPHERICALBASIS_KEYS))
    if bad:
        raise ValueError(f"SphericalBasis: unmatched parameter(s) {bad} (src/SphericalBasis.cc:30-52)")
    name = "SphericalBasis"
    g = conf.get
    # ---- refused when they ask for something (defaults of src/SphericalBasis.cc:66-90) ------------------------------
    noise = None
    if "NOISE" in conf and _bool(conf["NOISE"]):
        # update_noise (src/SphericalBasis.cc:2150-2210) with compute_rms_coefs (:2108-2147) of `noise_model_file`.  Two quirks of
        # the reference, kept: `noiseN` is read `.as<bool>()` (:143) -- any true value is 1.0, a false one 0.0 (a division by
        # zero there), the default 1e-6 only when the key is absent -- and `seedN` has no default at all (an uninitialised
        # member, src/SphericalBasis.H:339): it must be given here
        if "seedN" not in conf:
            raise _refuse(name, "NOISE", conf["NOISE"], "NOISE needs `seedN`: the reference seeds its generator with an "
                          "uninitialised member when the key is absent (src/SphericalBasis.H:339, src/SphericalBasis.cc:147)")
        noise = (str(g("noise_model_file", "SLGridSph.model")),
                 (1.0 if _bool(conf["noiseN"]) else 0.0) if "noiseN" in conf else 1.0e-6, int(conf["seedN"]))
        if noise[1] == 0.0:
            raise _refuse(name, "noiseN", conf["noiseN"], "read as a boolean by the reference (src/SphericalBasis.cc:143): "
                          "false is a division by zero in update_noise (:2197)")
    ss = float(g("ssfrac", 0.0))
    if 0.0 < ss < 1.0 and multistep > 0:
        # `subset`: thread id walks [n id / nthrds, floor(ssfrac n (id + 1) / nthrds)) of every LEVEL list (src/SphericalBasis.cc:
        # 435-460); with block multistep those lists are in the order of the run's level changes -- nothing to reproduce
        raise _refuse(name, "ssfrac", ss, "with block multistep the sub-sample is a slice of level lists whose order is the "
                      "history of the level changes (src/SphericalBasis.cc:435-460); single-level runs take it")
    for key in ("subsampleFloat", "totalCovar", "fullCovar"):
        if key in conf and _bool(conf[key]):
            raise _refuse(name, key, conf[key], "the n-body sub-sample covariance (nint, src/SphericalBasis.cc:700-720) is "
                          "not driven from here; SphereSL.cov_enable / cov_accumulate give the pyEXP form of it")
    # ---- honoured ------------------------------------------------------------------------------------------------------
    # (noiseN, noise_model_file, seedN are only read when NOISE is on: above; coefMaster says which rank writes the coefficient
    # file: the one process that calls dump_coefs here)
    kw = dict(scale=float(g("scale", 1.0)),
              NO_L0=_bool(g("NO_L0", False)), NO_L1=_bool(g("NO_L1", False)), EVEN_L=_bool(g("EVEN_L", False)),
              EVEN_M=_bool(g("EVEN_M", False)), M0_only=_bool(g("M0_ONLY", False)),
              self_consistent=_bool(g("self_consistent", True)), FIX_L0=_bool(g("FIX_L0", False)), multistep=multistep)
    # Sphere::Sphere takes the window from the SL grid (src/Sphere.cc:65-67); the keys of the base class narrow it
    rmin = max(float(g("rmin", 0.0)), grid.rmin) if "rmin" in conf else grid.rmin
    rmax = min(float(g("rmax", grid.rmax)), grid.rmax) if "rmax" in conf else grid.rmax
    f = cls(ctx, grid, rmin=rmin, rmax=rmax, **kw)
    if noise is not None:
        f.set_noise(noise[0], noiseN=noise[1], seedN=noise[2])
    if 0.0 < ss < 1.0:
        # (the level list of a single-level run is the caller's particle order here; the reference's is the iteration order
        # of its particle map -- exp_amd_sph_set_subset, include/exp_amd.h)
        f.set_subset(ss, nthrds)
    if "orthocheck" in conf and _bool(conf["orthocheck"]):
        # SphericalBasis::orthoTest (src/SphericalBasis.cc:2109-2150): the worst deviation of the biorthogonality matrix
        from .slgrid import orthocheck_max
        f.orthocheck_worst = orthocheck_max(grid)
    if "playback" in conf:
        f._pending_playback = (str(conf["playback"]), _bool(g("coefCompute", False)))    # needs the run's dtime: set_playback
    elif "coefCompute" in conf and _bool(conf["coefCompute"]):
        raise ValueError("SphericalBasis: coefCompute without playback (src/SphericalBasis.cc:155-213)")
    return f


# keys that describe how the EOF tables are made -> (argument of exp_amd.empcyl.build_empcyl, attribute of EmpCylGrid or
# None when the grid does not record it, reference default src/Cylinder.cc:104-135)
_CYL_TABLE_KEYS = {
    "mmax": ("mmax", "mmax", 6), "nmax": ("norder", "norder", 18), "ncylnx": ("numx", "numx", 256),
    "ncylny": ("numy", "numy", 128), "acyl": ("acyl", "ascale", 0.01), "hcyl": ("hcyl", "hscale", 0.002),
    "rcylmin": ("rcylmin", "rmin", 0.001), "rcylmax": ("rcylmax", "rmax", 20.0), "cmapr": ("cmapr", "cmapr", 1),
    "cmapz": ("cmapz", "cmapz", 1), "lmaxfid": ("lmaxfid", None, 128), "nmaxfid": ("nmaxfid", None, 64),
    "ncylr": ("numr", None, 2000), "rnum": ("rnum", None, 200), "pnum": ("pnum", None, 1), "tnum": ("tnum", None, 80),
    "ashift": ("ashift", None, 0.0), "ncylodd": ("nodd", None, None),
}


def _conditioning_particles(condition_on):
    """(mass, pos in the basis' frame) of what ``precond: false`` conditions on: a ``Component`` -- every body whatever its
    level, ``Pos(Local | Centered)``, frozen ones left out (src/Cylinder.cc:764-806) -- or a (mass, pos) pair already in
    that frame."""
    import numpy as np
    if hasattr(condition_on, "download"):
        d = condition_on.download(("mass", "pos"))
        mass, pos = d["mass"], d["pos"] - condition_on.center[None, :]
        rt = getattr(condition_on, "rtrunc", None)
        if rt is not None and rt < 1.0e20:                  # Component::freeze (src/Component.cc:4194-4202)
            c0 = getattr(condition_on, "com0", None)
            keep = np.linalg.norm(pos if c0 is None else pos - c0[None, :], axis=1) <= rt
            mass, pos = mass[keep], pos[keep]
        return mass, pos
    mass, pos = condition_on
    return np.asarray(mass, dtype=np.float64), np.asarray(pos, dtype=np.float64)


def cylinder_from_config(cls, ctx, conf: dict, multistep: int = 0, grid=None, condition_on=None):
    conf = dict(conf or {})
    bad = sorted(set(conf) - set(CYLINDER_KEYS))
    if bad:
        raise ValueError(f"Cylinder: unmatched parameter(s) {bad} (src/Cylinder.cc:24-80)")
    name = "Cylinder"
    g = conf.get
    if "cmap" in conf and "cmapr" not in conf:           # `cmap` is the older spelling of cmapr (src/Cylinder.cc:503-504)
        conf["cmapr"] = conf["cmap"]
    # ---- refused when they ask for something ------------------------------------------------------------------------
    for key, why in (("pcavar", "Hall smoothing / PCA of the coefficients (EmpCylSL::pca_hall)"),
                     ("pcaeof", "the PCA rotation of the EOF basis"), ("pcavtk", "VTK output of the PCA"),
                     ("pcadiag", "PCA diagnostics output"), ("subsamp", "sub-sampled covariance inside the step loop"),
                     ("fullCovar", "the n-body covariance accumulation"), ("totalCovar", "the n-body covariance accumulation"),
                     ("logr", "the logarithmic radial grid of the helper model (EmpCylSL::logarithmic)"),
                     ("dumpbasis", "the basis dump files (EmpCylSL::dump_basis)")):
        if key in conf and _bool(conf[key]):
            raise _refuse(name, key, conf[key], why + " is not built here")
    if int(g("nint", 0)) != 0:
        raise _refuse(name, "nint", conf["nint"], "the periodic sub-sample covariance of the step loop is not driven from here")
    # npca / npca0: the reference acts on them only under pcavar / pcaeof (`compute`, src/Cylinder.cc:1025-1027; pca_hall
    # does nothing without it, exputil/EmpCylSL.cc:4582; set_trimmed asks for pcavar, src/Cylinder.cc:1133) -- both refused
    # above when switched on -- so any value is inert here, as there
    for key in ("npca", "npca0"):
        if key in conf:
            int(conf[key])
    if "ncylrecomp" in conf and int(conf["ncylrecomp"]) >= 0:
        raise _refuse(name, "ncylrecomp", conf["ncylrecomp"], "re-making the EOF basis from the particles during a run "
                      "(src/Cylinder.cc:1133-1190) is not built here")
    # `expcond` is the deprecated spelling; a later `precond` overrides it (src/Cylinder.cc:492-493, :520-526)
    precond, pkey = True, None
    for key in ("expcond", "precond"):
        if key in conf:
            precond, pkey = _bool(conf[key]), key
    # precond: false -- the basis is conditioned on the PARTICLES at the first evaluation (`eof = 1`, src/Cylinder.cc:981-988,
    # determine_coefficients_eof :1202-1249).  A force method is made before it sees a component here, so the component (or
    # a (mass, pos) pair in the basis' frame) comes in as `condition_on`; without one there is nothing to condition on
    if not precond and grid is None and condition_on is None:
        raise _refuse(name, pkey, conf[pkey], "conditioning the basis on the PARTICLES (determine_coefficients_eof, "
                      "src/Cylinder.cc:1202-1249) needs them: Cylinder.from_config(..., condition_on=component)")
    if precond and condition_on is not None:
        raise ValueError("Cylinder: condition_on is given but the keys ask for the analytic conditioning (precond: true, the "
                         "default, src/Cylinder.cc:131): say precond: false")
    if "pyname" in conf:
        raise _refuse(name, "pyname", conf["pyname"], "a Python target density: pass a callable to "
                      "exp_amd.empcyl.build_empcyl(dens=...) and hand the grid in")
    if "mtype" in conf and str(conf["mtype"]).lower() not in ("exponential",):
        raise _refuse(name, "mtype", conf["mtype"], "the deprojected target models (src/Cylinder.cc:243-290)")
    for key, dflt in (("bias", 1.0), ("hexp", 1.0), ("snr", 1.0), ("ppower", 4.0)):
        if key in conf and float(conf[key]) != dflt:
            raise _refuse(name, key, conf[key], f"only the default {dflt} is built")
    if "evcut" in conf and float(conf["evcut"]) >= 0.0:
        raise _refuse(name, "evcut", conf["evcut"], "the eigenvalue cut of the Hall smoothing")
    if "samplesz" in conf and int(conf["samplesz"]) != 1:
        raise _refuse(name, "samplesz", conf["samplesz"], "sub-sample partitions of the covariance")
    if "tk_type" in conf and str(conf["tk_type"]).lower() not in ("null", "none"):
        raise _refuse(name, "tk_type", conf["tk_type"], "Hall truncation of the coefficients (EmpCylSL::setTK)")
    if "nvtk" in conf and int(conf["nvtk"]) != 1:
        raise _refuse(name, "nvtk", conf["nvtk"], "VTK output frequency of the PCA")
    for key in ("cachename", "eof_file"):
        if key in conf and grid is None:
            raise _refuse(name, key, conf[key], "read the cache with exp_amd.h5cache.read_empcyl_cache (EXP's HDF5 layout) "
                          "and pass it as grid=")
    # (sech2 selects sech^2(z/2h) against sech^2(z/h) for the conditioning density, src/Cylinder.cc:315-322: the table
    # builder conditions on sech^2(z/2h), the reference's current default ...)
    if "sech2" in conf and not _bool(conf["sech2"]) and grid is None:
        raise _refuse(name, "sech2", conf["sech2"], "the builder conditions on sech^2(z/(2h)) (src/Cylinder.cc:315-322)")
    # ---- the tables: built from the keys, or checked against the grid handed in -------------------------------------------
    if grid is None:
        from .empcyl import build_empcyl
        kw = {}
        for key, (arg, _attr, dflt) in _CYL_TABLE_KEYS.items():
            if key in conf:
                kw[arg] = type(dflt)(conf[key]) if dflt is not None else int(conf[key])
            elif dflt is not None:
                kw[arg] = dflt
        if "nodd" not in kw:
            kw["nodd"] = kw["norder"] // 4                 # `ncylodd = nmax/4` (src/Cylinder.cc:549-551)
        if not precond:
            kw["particles"] = _conditioning_particles(condition_on)
        grid = build_empcyl(**kw)
    else:
        for key, (_arg, attr, _dflt) in _CYL_TABLE_KEYS.items():
            if key not in conf:
                continue
            if attr is None:
                raise ValueError(f"Cylinder: key '{key}' describes how the EOF tables are MADE and the grid handed in does "
                                 "not record it: drop the key, or let from_config build the tables (grid=None)")
            have = getattr(grid, attr)
            if abs(float(have) - float(conf[key])) > 1e-12 * max(1.0, abs(float(have))):
                raise ValueError(f"Cylinder: key '{key}: {conf[key]}' contradicts the tables handed in ({attr} = {have})")
    # ---- honoured at run time ------------------------------------------------------------------------------------------
    # (try_cache / override / density / vflag / coefMaster: cache policy, deprecated no-ops, verbosity and the rank that
    # writes the coefficient file -- none reaches the hot path in the reference either)
    mlim = int(g("mlim", -1))
    f = cls(ctx, grid, rcylmax=float(g("rcylmax", grid.rmax)), EVEN_M=_bool(g("EVEN_M", False)), multistep=multistep,
            self_consistent=_bool(g("self_consistent", True)), mlim=mlim)
    if "playback" in conf:
        f._pending_playback = (str(conf["playback"]), _bool(g("coefCompute", False)))
    elif "coefCompute" in conf and _bool(conf["coefCompute"]):
        raise ValueError("Cylinder: coefCompute without playback (src/Cylinder.cc:560-618)")
    return f


# ---- src/Component.cc:40-95: Component::valid_keys_parm -------------------------------------------------------------------
COMPONENT_KEYS = ("name", "parameters", "bodyfile", "force", "EJ", "nEJkeep", "nEJwant", "nEJaccel", "EJkinE", "EJext", "EJdiag",
                  "EJdryrun", "EJx0", "EJy0", "EJz0", "EJu0", "EJv0", "EJw0", "EJdT", "EJlinear", "EJdamp", "binary", "adiabatic",
                  "ton", "toff", "twid", "rtrunc", "rcom", "consp", "tidal", "comlog", "bunch", "timers", "com", "indexing",
                  "aindex", "magic", "nlevel", "keypos", "pbufsiz", "blocking", "ctr_name", "buffered", "noswitch", "freezeL",
                  "dtreset", "H5compress", "H5shuffle", "H5chunk")


def configure_component(sim, index: int, comp, conf: dict, com0=None, centerlevl: int = -1, logfile: Optional[str] = None,
                        names: Optional[dict] = None):
    """``Component::configure`` and the EJ block of ``Component::initialize`` (src/Component.cc:985-1075, :1323-1370) for the
    ``parameters`` block of one component of an EXP configuration file: every key of ``Component::valid_keys_parm`` (:40-95)
    is HONOURED -- it reaches the device store, the step driver or the orientation estimator -- or REFUSED by name with what is
    missing; a key outside the set is refused as the reference refuses it.  ``sim`` is the ``Simulation`` the component was
    added to as number ``index`` (``sim.add_component``), ``comp`` its ``Component``; call before ``sim.init()``.  ``com0``: the
    centre ``rtrunc`` / ``rcom`` are measured from (zeros, as ``com_system`` is off here); ``names``: component name -> its number
    in ``sim``, for ``ctr_name``.  Returns the ``Orient`` it made
    (EJ != 0) or None.  ``tests/test_config_keys_gpu.py`` walks the set."""
    from .runtime import Orient
    conf = dict(conf or {})
    bad = sorted(set(conf) - set(COMPONENT_KEYS))
    if bad:
        raise ValueError(f"Component: unmatched parameter(s) {bad} (src/Component.cc:40-95)")
    name = "Component"
    g = conf.get
    # ---- refused when they ask for something that is not built --------------------------------------------------------------
    if "com" in conf and _bool(conf["com"]):
        raise _refuse(name, "com", conf["com"], "the centre-of-mass coordinate system (com_system: positions local to a moving "
                      "frame, incr_com_position / incr_com_velocity, src/Component.cc:3555-3590) is not built")
    for key, why in (("comlog", "the centre-of-mass log file"), ("timers", "the per-component timing report"),
                     ("aindex", "re-indexing the bodies from an attribute column"), ("EJdiag", "the estimator's diagnostic output")):
        if key in conf and _bool(conf[key]):
            raise _refuse(name, key, conf[key], why + " is not built")
    if "keypos" in conf and int(conf["keypos"]) >= 0:
        raise _refuse(name, "keypos", conf["keypos"], "species keys in an integer attribute belong to the collision modules")
    if "ctr_name" in conf and str(conf["ctr_name"]) != "":
        # Component::find_ctr_component (src/Component.cc:284-310): the centre follows the component of that name
        if not names or str(conf["ctr_name"]) not in names:
            raise _refuse(name, "ctr_name", conf["ctr_name"], "no component of that name among `names` (name -> its number in "
                          "the Simulation); the reference stops as well when it finds none (src/Component.cc:312-320)")
        sim.set_center_from(index, int(names[str(conf["ctr_name"])]))
    # (name, parameters, bodyfile, force: the structure of the file, read by whoever builds the components; binary, indexing,
    # magic, pbufsiz, blocking, buffered, H5compress, H5shuffle, H5chunk: how phase-space files are read and written --
    # exp_amd.reader / write_psp take them as arguments; bunch: the CUDA path's batch size; nlevel: how often the level
    # populations are reported.  None of them reaches the path in the reference either.)
    # ---- honoured: the store ---------------------------------------------------------------------------------------------------
    if "rtrunc" in conf or com0 is not None:
        comp.set_rtrunc(float(g("rtrunc", 1.0e20)), com0)
    if "tidal" in conf or ("consp" in conf and _bool(conf["consp"])):
        # (`tidal` is what switches consp on, src/Component.cc:998-1000; a bare `consp: true` leaves tidal at -1 and the thread
        # body then tests nothing, :3317)
        if "tidal" in conf and int(conf["tidal"]) >= 0:
            comp.set_consp(float(g("rcom", 1.0e20)))
    if any(k in conf for k in ("noswitch", "freezeL", "dtreset")):
        comp.set_level_policy(noswitch=_bool(g("noswitch", False)), freeze_levels=_bool(g("freezeL", False)),
                              dtreset=_bool(g("dtreset", True)))
    # ---- honoured: the step driver ---------------------------------------------------------------------------------------------
    if any(k in conf for k in ("ton", "toff", "twid")):      # each of the three switches `adiabatic` on (src/Component.cc:1040-1055)
        sim.set_adiabatic(index, float(g("ton", -1.0e20)), float(g("toff", 1.0e20)), float(g("twid", 0.1)))
    elif "adiabatic" in conf and _bool(conf["adiabatic"]):
        raise ValueError("Component: 'adiabatic' is set by ton / toff / twid (src/Component.cc:1040-1055), not by itself")
    # ---- honoured: the orientation estimator (src/Component.cc:1323-1370) -------------------------------------------------------
    ej = int(g("EJ", 0))
    if not ej:
        return None
    ctl = (Orient.KE if _bool(g("EJkinE", True)) else 0) | (Orient.EXTERNAL if _bool(g("EJext", False)) else 0)
    o = Orient(comp.ctx, int(g("nEJkeep", 100)), int(g("nEJwant", 500)), ej, ctl, float(g("EJdT", 0.0)), float(g("EJdamp", 1.0)))
    if int(g("nEJaccel", 0)) > 0:
        o.set_naccel(int(conf["nEJaccel"]))
    if _bool(g("EJlinear", False)):
        o.set_linear()
    o.set_center(float(g("EJx0", 0.0)), float(g("EJy0", 0.0)), float(g("EJz0", 0.0)))
    o.set_cenvel(float(g("EJu0", 0.0)), float(g("EJv0", 0.0)), float(g("EJw0", 0.0)))
    comp.set_center([float(g("EJx0", 0.0)), float(g("EJy0", 0.0)), float(g("EJz0", 0.0))])
    if logfile:
        o.openLog(logfile)
    sim.set_orient(index, o, dryrun=_bool(g("EJdryrun", False)), centerlevl=centerlevl)
    return o
