"""The YAML keys of EXP's n-body force methods, one by one.

``SphereSL.from_config`` / ``Cylinder.from_config`` (exp_amd/runtime.py) take the ``parameters`` block a component's
force has in an EXP configuration file.  Every key of ``SphericalBasis::valid_keys`` (src/SphericalBasis.cc:30-52) and
``Cylinder::valid_keys`` (src/Cylinder.cc:24-80) is either HONOURED -- it reaches the device path, or it is checked
against the tables that were handed in -- or REFUSED with a ``ValueError`` that says what is missing; a key outside the
two sets is refused as the reference refuses it (``unmatched()``, src/PotAccel.H:323).  Nothing is accepted and dropped:
``tests/test_config_keys_gpu.py`` walks both sets.

A key whose value is the reference's default for a feature that is not built here (``NOISE: false``, ``pcavar: false``,
``ssfrac: 0`` ...) asks for nothing and is honoured as such; only asking for the feature is refused.
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


def sphere_from_config(cls, ctx, grid, conf: dict, multistep: int = 0):
    conf = dict(conf or {})
    bad = sorted(set(conf) - set(SPHERICALBASIS_KEYS))
    if bad:
        raise ValueError(f"SphericalBasis: unmatched parameter(s) {bad} (src/SphericalBasis.cc:30-52)")
    name = "SphericalBasis"
    g = conf.get
    # ---- refused when they ask for something (defaults of src/SphericalBasis.cc:66-90) ------------------------------
    if "NOISE" in conf and _bool(conf["NOISE"]):
        raise _refuse(name, "NOISE", conf["NOISE"], "the noise-model coefficients (src/SphericalBasis.cc:907-1000) are not "
                      "on the device path")
    ss = float(g("ssfrac", 0.0))
    if 0.0 < ss < 1.0:
        # `subset`: the first floor(ssfrac * n) entries of every thread's slice of the level list (src/SphericalBasis.cc:
        # 459-460) -- which particles those are follows the iteration order of the reference's hash map of particles
        raise _refuse(name, "ssfrac", ss, "the sub-sample is whatever order EXP's particle map iterates in "
                      "(src/SphericalBasis.cc:459-460); there is no such order to reproduce")
    for key in ("subsampleFloat", "totalCovar", "fullCovar"):
        if key in conf and _bool(conf[key]):
            raise _refuse(name, key, conf[key], "the n-body sub-sample covariance (nint, src/SphericalBasis.cc:700-720) is "
                          "not driven from here; SphereSL.cov_enable / cov_accumulate give the pyEXP form of it")
    # ---- honoured ------------------------------------------------------------------------------------------------------
    # (noiseN, noise_model_file, seedN are only read when NOISE is on; coefMaster says which rank writes the coefficient
    # file: the one process that calls dump_coefs here)
    kw = dict(scale=float(g("scale", 1.0)),
              NO_L0=_bool(g("NO_L0", False)), NO_L1=_bool(g("NO_L1", False)), EVEN_L=_bool(g("EVEN_L", False)),
              EVEN_M=_bool(g("EVEN_M", False)), M0_only=_bool(g("M0_ONLY", False)),
              self_consistent=_bool(g("self_consistent", True)), FIX_L0=_bool(g("FIX_L0", False)), multistep=multistep)
    # Sphere::Sphere takes the window from the SL grid (src/Sphere.cc:65-67); the keys of the base class narrow it
    rmin = max(float(g("rmin", 0.0)), grid.rmin) if "rmin" in conf else grid.rmin
    rmax = min(float(g("rmax", grid.rmax)), grid.rmax) if "rmax" in conf else grid.rmax
    f = cls(ctx, grid, rmin=rmin, rmax=rmax, **kw)
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


def cylinder_from_config(cls, ctx, conf: dict, multistep: int = 0, grid=None):
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
    for key in ("npca", "npca0"):
        if key in conf and int(conf[key]) < 2 ** 31 - 1:
            raise _refuse(name, key, conf[key], "PCA / Hall smoothing (src/Cylinder.cc:1120-1131) is not built here")
    if "ncylrecomp" in conf and int(conf["ncylrecomp"]) >= 0:
        raise _refuse(name, "ncylrecomp", conf["ncylrecomp"], "re-making the EOF basis from the particles during a run "
                      "(src/Cylinder.cc:1133-1190) is not built here")
    for key in ("precond", "expcond"):
        if key in conf and not _bool(conf[key]):
            raise _refuse(name, key, conf[key], "conditioning the basis on the PARTICLES (determine_coefficients_eof, "
                          "src/Cylinder.cc:1018-1080) is not built here: the tables are conditioned on the analytic disk")
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
