"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  Test infrastructure:
imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
from numpy.polynomial import legendre as npleg

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)


def _dp(a):
    return a.ctypes.data_as(c_double_p)


class _SLGrid(ctypes.Structure):
    _fields_ = [("lmax", ctypes.c_int), ("nmax", ctypes.c_int), ("numr", ctypes.c_int),
                ("cmap", ctypes.c_int), ("rmin", ctypes.c_double), ("rmax", ctypes.c_double),
                ("rmap", ctypes.c_double), ("xmin", ctypes.c_double), ("xmax", ctypes.c_double),
                ("dxi", ctypes.c_double), ("xi", c_double_p), ("p0", c_double_p),
                ("d0", c_double_p), ("ev", c_double_p), ("ef", c_double_p)]


class _SphParams(ctypes.Structure):
    _fields_ = [("scale", ctypes.c_double), ("rmin", ctypes.c_double), ("rmax", ctypes.c_double),
                ("NO_L0", ctypes.c_int), ("NO_L1", ctypes.c_int), ("EVEN_L", ctypes.c_int),
                ("EVEN_M", ctypes.c_int), ("M0_only", ctypes.c_int), ("N1", ctypes.c_int),
                ("N2", ctypes.c_int)]


class _CylGrid(ctypes.Structure):
    _fields_ = [("mmax", ctypes.c_int), ("norder", ctypes.c_int), ("numx", ctypes.c_int),
                ("numy", ctypes.c_int), ("cmapr", ctypes.c_int), ("cmapz", ctypes.c_int),
                ("EVEN_M", ctypes.c_int), ("ascale", ctypes.c_double), ("hscale", ctypes.c_double),
                ("rtable", ctypes.c_double), ("xmin", ctypes.c_double), ("dx", ctypes.c_double),
                ("ymin", ctypes.c_double), ("dy", ctypes.c_double), ("rcylmax", ctypes.c_double),
                ("acyl", ctypes.c_double), ("tab", c_double_p)]


def build_oracle() -> str:
    so = os.path.join(_ORACLE_DIR, "_build", "liboracle.so")
    srcs = [os.path.join(_ORACLE_DIR, f) for f in os.listdir(_ORACLE_DIR)
            if f.endswith((".c", ".h"))]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _ORACLE_DIR], stdout=subprocess.DEVNULL)
    return so


ORC_ORIENT_HIST = 64


class OrcOrient(ctypes.Structure):
    """orc_orient (oracle/bfe_oracle.h)."""
    _fields_ = [("keep", ctypes.c_int), ("many", ctypes.c_int), ("oflags", ctypes.c_uint),
                ("cflags", ctypes.c_uint), ("deltaT", ctypes.c_double), ("damp", ctypes.c_double),
                ("linear", ctypes.c_int),
                ("center", ctypes.c_double * 3), ("center0", ctypes.c_double * 3),
                ("cenvel0", ctypes.c_double * 3), ("axis", ctypes.c_double * 3),
                ("axis1", ctypes.c_double * 3), ("center1", ctypes.c_double * 3),
                ("body", ctypes.c_double * 9), ("orig", ctypes.c_double * 9),
                ("lasttime", ctypes.c_double), ("Ecurr", ctypes.c_double), ("sigA", ctypes.c_double),
                ("sigC", ctypes.c_double), ("sigCz", ctypes.c_double), ("mtot", ctypes.c_double),
                ("used", ctypes.c_long), ("nA", ctypes.c_int), ("nC", ctypes.c_int),
                ("tA", ctypes.c_double * ORC_ORIENT_HIST),
                ("vA", (ctypes.c_double * 3) * ORC_ORIENT_HIST),
                ("tC", ctypes.c_double * ORC_ORIENT_HIST),
                ("vC", (ctypes.c_double * 3) * ORC_ORIENT_HIST)]


class _NBodyComp(ctypes.Structure):
    """orc_nbody_comp (oracle/nbody_oracle.h)."""
    _fields_ = [("kind", ctypes.c_int), ("sg", ctypes.POINTER(_SLGrid)), ("sp", ctypes.POINTER(_SphParams)),
                ("cg", ctypes.POINTER(_CylGrid)), ("n", ctypes.c_long)] + \
               [(k, c_double_p) for k in ("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "pot", "mass")] + \
               [("level", c_int_p), ("center", ctypes.c_double * 3), ("ncoef", ctypes.c_long),
                ("coefN", c_double_p), ("coefL", c_double_p), ("coef", c_double_p),
                ("cylmass", ctypes.c_double), ("used", ctypes.c_long), ("resetT", ctypes.c_double),
                # the keys that default to "off" (oracle/nbody_oracle.h)
                ("has_rtrunc", ctypes.c_int), ("rtrunc", ctypes.c_double), ("com0", ctypes.c_double * 3),
                ("adiabatic", ctypes.c_int), ("ton", ctypes.c_double), ("toff", ctypes.c_double),
                ("twid", ctypes.c_double), ("not_self_consistent", ctypes.c_int), ("coef_calls", ctypes.c_int),
                ("fix_l0", ctypes.c_int), ("have_c0", ctypes.c_int), ("C0", c_double_p),
                ("mlim", ctypes.c_int), ("has_mlim", ctypes.c_int), ("freeze_lev", ctypes.c_int),
                ("noswitch", ctypes.c_int), ("no_dtreset", ctypes.c_int), ("dtreq", ctypes.POINTER(ctypes.c_float)),
                ("noise", ctypes.c_void_p), ("noise_buf", c_double_p),
                ("ssfrac", ctypes.c_double), ("ss_nthrds", ctypes.c_int)]


class _NBody(ctypes.Structure):
    """orc_nbody (oracle/nbody_oracle.h)."""
    _fields_ = [("ncomp", ctypes.c_int), ("comp", ctypes.POINTER(_NBodyComp)), ("ninter", ctypes.c_int),
                ("inter", c_int_p), ("multistep", ctypes.c_int), ("dtime", ctypes.c_double),
                ("dynfrac", ctypes.c_double * 5), ("shiftlevl", ctypes.c_int), ("this_step", ctypes.c_long),
                ("tnow", ctypes.c_double), ("initializing", ctypes.c_int), ("no_eqmotion", ctypes.c_int)]


class NBodyOracle:
    """The multi-component block-multistep step loop of oracle/nbody_oracle.c (do_step, begin_run,
    ComponentContainer, adjust_multistep_level).  Components are added with ``add_sphere`` /
    ``add_cylinder``; the state arrays live in ``self.state[k]`` (numpy, updated in place)."""

    def __init__(self, oracle, multistep, dtime, dynfrac, shiftlevl=0):
        self.orc, self.lib = oracle, oracle.lib
        self.multistep, self.dtime, self.shiftlevl = int(multistep), float(dtime), int(shiftlevl)
        self.dyn = [float(v) for v in dynfrac]
        self.state, self._grids, self.inter = [], [], []
        self.S = None

    def _add(self, kind, grid_struct, prm, ncoef, mass, pos, vel, center):
        n = len(mass)
        st = {k: np.ascontiguousarray(pos[:, j], dtype=np.float64).copy() for j, k in enumerate("xyz")}
        st.update({"v" + k: np.ascontiguousarray(vel[:, j], dtype=np.float64).copy() for j, k in enumerate("xyz")})
        for k in ("ax", "ay", "az", "pot"):
            st[k] = np.zeros(n)
        st["mass"] = np.ascontiguousarray(mass, dtype=np.float64).copy()
        st["level"] = np.zeros(n, dtype=np.int32)
        st["coefN"] = np.zeros((self.multistep + 1, ncoef))
        st["coefL"] = np.zeros((self.multistep + 1, ncoef))
        st["coef"] = np.zeros(ncoef)
        st.update(kind=kind, ncoef=ncoef, center=np.asarray(center, dtype=np.float64), n=n)
        self.state.append(st)
        self._grids.append((grid_struct, prm))
        return len(self.state) - 1

    def add_sphere(self, g, prm, mass, pos, vel, center=(0.0, 0.0, 0.0)):
        return self._add(0, self.orc.grid(g), prm, (g.lmax + 1) ** 2 * g.nmax, mass, pos, vel, center)

    def add_cylinder(self, g, mass, pos, vel, center=(0.0, 0.0, 0.0), **kw):
        return self._add(1, self.orc.cylgrid(g, **kw), None, 2 * (g.mmax + 1) * g.norder, mass, pos, vel, center)

    def add_interaction(self, source, target):
        self.inter.append((int(source), int(target)))

    def set_options(self, k, rtrunc=None, com0=(0.0, 0.0, 0.0), adiabatic=None, self_consistent=True, fix_l0=False,
                    mlim=None, freeze_levels=False, noswitch=False, dtreset=True, ssfrac=None, nthrds=1, noise=None):
        """The keys of component ``k`` that default to off (oracle/nbody_oracle.h): ``rtrunc`` (+ ``com0``),
        ``adiabatic = (ton, toff, twid)``, ``self_consistent``, ``FIX_L0`` (sphere), ``mlim`` (cylinder), ``freezeL``, ``noswitch`` /
        ``dtreset``, the sphere's ``ssfrac`` (with the thread count ``nthrds`` its partition of the level list depends on) and
        ``noise = (meanC, rmsC, noiseN, seedN)`` (the NOISE mode: ``Oracle.noise_create`` is called with it)."""
        self.state[k]["options"] = dict(rtrunc=rtrunc, com0=tuple(com0), adiabatic=adiabatic,
                                        self_consistent=self_consistent, fix_l0=fix_l0, mlim=mlim,
                                        freeze_levels=bool(freeze_levels), noswitch=bool(noswitch), dtreset=bool(dtreset),
                                        ssfrac=ssfrac, nthrds=int(nthrds), noise=noise)

    def _build(self):
        nc = len(self.state)
        self._comps = (_NBodyComp * nc)()
        for k, st in enumerate(self.state):
            c = self._comps[k]
            G, prm = self._grids[k]
            c.kind = st["kind"]
            if st["kind"] == 0:
                c.sg, c.sp = ctypes.pointer(G), ctypes.pointer(prm)
            else:
                c.cg = ctypes.pointer(G)
            c.n = st["n"]
            for key in ("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "pot", "mass"):
                setattr(c, key, _dp(st[key]))
            c.level = st["level"].ctypes.data_as(c_int_p)
            for j in range(3):
                c.center[j] = st["center"][j]
            c.ncoef = st["ncoef"]
            c.coefN, c.coefL, c.coef = _dp(st["coefN"]), _dp(st["coefL"]), _dp(st["coef"])
            o = st.get("options")
            if o:
                if o["rtrunc"] is not None:
                    c.has_rtrunc, c.rtrunc = 1, float(o["rtrunc"])
                    for j in range(3):
                        c.com0[j] = o["com0"][j]
                if o["adiabatic"] is not None:
                    c.adiabatic = 1
                    c.ton, c.toff, c.twid = (float(v) for v in o["adiabatic"])
                c.not_self_consistent = 0 if o["self_consistent"] else 1
                if o["fix_l0"]:
                    st["C0"] = np.zeros(G.nmax)
                    c.fix_l0, c.C0 = 1, _dp(st["C0"])
                if o["mlim"] is not None:
                    c.has_mlim, c.mlim = 1, int(o["mlim"])
                c.freeze_lev = 1 if o.get("freeze_levels") else 0
                if o.get("ssfrac") is not None:
                    c.ssfrac, c.ss_nthrds = float(o["ssfrac"]), int(o.get("nthrds", 1))
                if o.get("noise") is not None:
                    st["noise_handle"] = self.orc.noise_create(G.lmax, G.nmax, *o["noise"])
                    st["noise_buf"] = np.zeros(st["ncoef"])
                    c.noise, c.noise_buf = st["noise_handle"], _dp(st["noise_buf"])
                if o.get("noswitch"):
                    st["dtreq"] = np.zeros(st["n"], np.float32)
                    c.noswitch, c.no_dtreset = 1, 0 if o.get("dtreset", True) else 1
                    c.dtreq = st["dtreq"].ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        self._inter = np.ascontiguousarray(np.array(self.inter, dtype=np.int32).reshape(-1))
        S = _NBody()
        S.ncomp, S.comp = nc, self._comps
        S.ninter = len(self.inter)
        S.inter = self._inter.ctypes.data_as(c_int_p) if len(self.inter) else None
        S.multistep, S.dtime, S.shiftlevl = self.multistep, self.dtime, self.shiftlevl
        S.no_eqmotion = 0 if getattr(self, "eqmotion", True) else 1        # (the global "eqmotion", src/global.cc:54)
        for j in range(5):
            S.dynfrac[j] = self.dyn[j]
        self.S = S

    def init(self, pass0_only=False):
        self._build()
        (self.lib.orc_nbody_init_pass0 if pass0_only else self.lib.orc_nbody_init)(ctypes.byref(self.S))

    def step(self):
        """One master step; returns the level changes per component."""
        nsw = (ctypes.c_long * len(self.state))()
        self.lib.orc_nbody_step(ctypes.byref(self.S), nsw)
        return [int(v) for v in nsw]

    def cylmass(self, k):
        return float(self._comps[k].cylmass)

    def used(self, k):
        return int(self._comps[k].used)

    @property
    def time(self):
        return float(self.S.tnow)


class _CallOpts(ctypes.Structure):
    """orc_call_opts (oracle/bfe_oracle.h)"""
    _fields_ = [("adb", ctypes.c_double), ("frz", ctypes.c_int), ("rtrunc", ctypes.c_double),
                ("com0", ctypes.c_double * 3), ("fcenter", ctypes.c_double * 3), ("mlim", ctypes.c_int),
                ("ssfrac", ctypes.c_double), ("nthrds", ctypes.c_int)]


class Oracle:
    def call_opts(self, adb=1.0, rtrunc=None, com0=(0.0, 0.0, 0.0), fcenter=(0.0, 0.0, 0.0), mlim=None, ssfrac=None, nthrds=1):
        """Context manager: the per-call options of the thread bodies (Adiabatic factor, Component::freeze of the
        component walked, the cylinder's mlim, the sphere's ssfrac with its thread count) for the oracle calls made inside
        the ``with`` block."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            o = _CallOpts(float(adb), 0 if rtrunc is None else 1, 1e20 if rtrunc is None else float(rtrunc),
                          (ctypes.c_double * 3)(*com0), (ctypes.c_double * 3)(*fcenter), -1 if mlim is None else int(mlim),
                          1.0 if ssfrac is None else float(ssfrac), int(nthrds))
            self.lib.orc_set_call_opts(ctypes.byref(o))
            try:
                yield
            finally:
                self.lib.orc_set_call_opts(None)
        return cm()

    def __init__(self):
        self.lib = ctypes.CDLL(build_oracle())
        L = self.lib
        L.orc_adiabatic.restype = ctypes.c_double
        L.orc_adiabatic.argtypes = [ctypes.c_double] * 4
        L.orc_sph_accumulate.restype = ctypes.c_long
        L.orc_level_select.restype = ctypes.c_int
        L.orc_mstep_create.restype = ctypes.c_void_p
        L.orc_sl_r_to_xi.restype = ctypes.c_double
        self._keep = []

    # -- helpers ---------------------------------------------------------------------
    def grid(self, g) -> _SLGrid:
        arrs = [np.ascontiguousarray(getattr(g, k), dtype=np.float64)
                for k in ("xi", "p0", "d0", "ev", "ef")]
        self._keep.append(arrs)
        return _SLGrid(g.lmax, g.nmax, g.numr, g.cmap, g.rmin, g.rmax, g.rmap, g.xmin, g.xmax,
                       g.dxi, *[_dp(a) for a in arrs])

    @staticmethod
    def params(scale=1.0, rmin=0.0, rmax=1e30, NO_L0=False, NO_L1=False, EVEN_L=False,
               EVEN_M=False, M0_only=False, N1=0, N2=-1) -> _SphParams:
        """N1, N2: pyEXP's radial window of the l >= 1 sums (N2 < 0: no upper limit)."""
        return _SphParams(scale, rmin, rmax, int(NO_L0), int(NO_L1), int(EVEN_L), int(EVEN_M),
                          int(M0_only), int(N1), int(N2))

    # -- Legendre / trig ---------------------------------------------------------------
    def legendre(self, lmax, x):
        p = np.zeros((lmax + 1, lmax + 1))
        self.lib.orc_legendre_R(ctypes.c_int(lmax), ctypes.c_double(x), _dp(p))
        return p

    def dlegendre(self, lmax, x):
        p = np.zeros((lmax + 1, lmax + 1))
        dp = np.zeros((lmax + 1, lmax + 1))
        self.lib.orc_dlegendre_R(ctypes.c_int(lmax), ctypes.c_double(x), _dp(p), _dp(dp))
        return p, dp

    def factorial(self, lmax):
        f = np.zeros((lmax + 1, lmax + 1))
        self.lib.orc_factorial_table(ctypes.c_int(lmax), _dp(f))
        return f

    # -- SL grid -----------------------------------------------------------------------
    def get_pot(self, g, r):
        G = self.grid(g)
        m = np.zeros((g.lmax + 1, g.nmax))
        self.lib.orc_sl_get_pot(ctypes.byref(G), ctypes.c_double(r), _dp(m))
        return m

    def get_force(self, g, r):
        G = self.grid(g)
        m = np.zeros((g.lmax + 1, g.nmax))
        self.lib.orc_sl_get_force(ctypes.byref(G), ctypes.c_double(r), _dp(m))
        return m

    def get_dens(self, g, r):
        G = self.grid(g)
        m = np.zeros((g.lmax + 1, g.nmax))
        self.lib.orc_sl_get_dens(ctypes.byref(G), ctypes.c_double(r), _dp(m))
        return m

    def orthocheck(self, g, num):
        G = self.grid(g)
        x, w = npleg.leggauss(num)
        knots = np.ascontiguousarray(0.5 * (x + 1.0))
        weights = np.ascontiguousarray(0.5 * w)
        ret = np.zeros((g.lmax + 1, g.nmax, g.nmax))
        self.lib.orc_sl_orthocheck(ctypes.byref(G), ctypes.c_int(num), _dp(knots), _dp(weights),
                                   _dp(ret))
        return ret

    # -- spherical hot path ------------------------------------------------------------
    def sph_accumulate(self, g, prm, pos, mass, center=(0.0, 0.0, 0.0), kahan=False):
        G = self.grid(g)
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        c = np.asarray(center, dtype=np.float64)
        coef = np.zeros(((g.lmax + 1) ** 2, g.nmax))
        used = self.lib.orc_sph_accumulate(ctypes.byref(G), ctypes.byref(prm),
                                           ctypes.c_long(len(m)), _dp(x), _dp(y), _dp(z), _dp(m),
                                           _dp(c), _dp(coef), ctypes.c_int(int(kahan)))
        return coef, int(used)

    def fix_positions(self, mass, pos, vel, acc, level, multistep, mlevel, lev_sums):
        """src/Component.cc:3280-3554; lev_sums [(multistep+1), 10] is updated in place."""
        cols = [np.ascontiguousarray(a[:, k], dtype=np.float64) for a in (pos, vel, acc) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        lv = np.ascontiguousarray(level, dtype=np.int32)
        out = np.zeros(10)
        self.lib.orc_fix_positions(ctypes.c_long(len(m)), _dp(m), *[_dp(c) for c in cols],
                                   lv.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(multistep),
                                   ctypes.c_int(mlevel), _dp(lev_sums), _dp(out))
        return out

    def fix_positions_opts(self, mass, pos, vel, acc, level, multistep, mlevel, lev_sums, com0, center, rcom=None,
                           iattr=None, rtrunc=1.0e20):
        """src/Component.cc:3280-3554 with the escape (consp / tidal / rcom: ``iattr`` int32 [n], updated in place) and
        freeze tests of the thread body (:3317-3336)."""
        cols = [np.ascontiguousarray(a[:, k], dtype=np.float64) for a in (pos, vel, acc) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        lv = np.ascontiguousarray(level, dtype=np.int32)
        c0 = np.ascontiguousarray(com0, dtype=np.float64)
        ce = np.ascontiguousarray(center, dtype=np.float64)
        out = np.zeros(10)
        if iattr is not None:
            assert iattr.dtype == np.int32 and iattr.flags.c_contiguous
        self.lib.orc_fix_positions_opts(ctypes.c_long(len(m)), _dp(m), *[_dp(c) for c in cols],
                                        lv.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(multistep), ctypes.c_int(mlevel),
                                        _dp(c0), _dp(ce), ctypes.c_double(rcom if rcom is not None else 1.0e20),
                                        iattr.ctypes.data_as(ctypes.c_void_p) if iattr is not None else None,
                                        ctypes.c_double(rtrunc), _dp(lev_sums), _dp(out))
        return out

    def orient(self, keep, many, oflags, cflags=0, deltaT=0.0, damp=1.0):
        """A fresh ``orc_orient`` (src/Orient.cc:38-80)."""
        o = OrcOrient()
        self.lib.orc_orient_init(ctypes.byref(o), int(keep), int(many), ctypes.c_uint(oflags),
                                 ctypes.c_uint(cflags), ctypes.c_double(deltaT), ctypes.c_double(damp))
        return o

    def orient_restart(self, o, logfile, restart, tnow, dtime, Mstep, naccel=0):
        """The restart block of the constructor (src/Orient.cc:84-335) -> (rows, queue [nq, 7])."""
        q = np.zeros((max(naccel, 1), 7))
        nq = ctypes.c_int(0)
        self.lib.orc_orient_restart.restype = ctypes.c_long
        rows = self.lib.orc_orient_restart(ctypes.byref(o), str(logfile).encode(), ctypes.c_int(int(restart)),
                                           ctypes.c_double(tnow), ctypes.c_double(dtime), ctypes.c_int(Mstep),
                                           ctypes.c_int(naccel), _dp(q), ctypes.byref(nq))
        return int(rows), q[:nq.value].copy()

    def orient_log_entry(self, o, logfile, time, com=(0, 0, 0), com0=(0, 0, 0), accel=(0, 0, 0),
                         omega=(0, 0, 0), domdt=(0, 0, 0)):
        """``Orient::logEntry`` (src/Orient.cc:742-785)."""
        v = [np.ascontiguousarray(a, dtype=np.float64) for a in (com, com0, accel, omega, domdt)]
        rc = self.lib.orc_orient_log_entry(ctypes.byref(o), str(logfile).encode(), ctypes.c_double(time),
                                           *[_dp(a) for a in v])
        assert rc == 0

    def orient_accumulate(self, o, time, dtime, mass, pos, vel, pot):
        """src/Orient.cc:325-747 (one process)."""
        cols = [np.ascontiguousarray(a[:, k], dtype=np.float64) for a in (pos, vel) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        p = np.ascontiguousarray(pot, dtype=np.float64)
        self.lib.orc_orient_accumulate(ctypes.byref(o), ctypes.c_double(time), ctypes.c_double(dtime),
                                       ctypes.c_long(len(m)), _dp(m), *[_dp(c) for c in cols], _dp(p))

    def pyexp_sph_covariance(self, g, prm, pos, mass, sampT, used0=0, acc=None):
        """Spherical::accumulate with pcavar (expui/BiorthBasis.cc:583-665) -> dict(counts, masses,
        mean [T, ltot, nmax] complex, covr [T, ltot, nmax, nmax] real, used); ``acc`` continues one."""
        G = self.grid(g)
        ltot = (g.lmax + 1) * (g.lmax + 2) // 2
        if acc is None:
            acc = {"counts": np.zeros(sampT, dtype=np.int64), "masses": np.zeros(sampT),
                   "mean2": np.zeros((sampT, ltot, g.nmax, 2)), "covr": np.zeros((sampT, ltot, g.nmax, g.nmax)),
                   "used": int(used0)}
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        self.lib.orc_pyexp_sph_covariance.restype = ctypes.c_long
        n = self.lib.orc_pyexp_sph_covariance(ctypes.byref(G), ctypes.byref(prm), ctypes.c_long(len(m)),
                                              _dp(x), _dp(y), _dp(z), _dp(m), ctypes.c_int(sampT),
                                              ctypes.c_long(acc["used"]),
                                              acc["counts"].ctypes.data_as(ctypes.c_void_p),
                                              _dp(acc["masses"]), _dp(acc["mean2"]), _dp(acc["covr"]))
        acc["used"] += int(n)
        acc["mean"] = acc["mean2"][..., 0] + 1j * acc["mean2"][..., 1]
        return acc

    # -- tuned CPU baseline (oracle/tuned_cpu.c) ------------------------------------------------------
    def tuned(self, g):
        self.lib.orc_tuned_create.restype = ctypes.c_void_p
        return ctypes.c_void_p(self.lib.orc_tuned_create(ctypes.byref(self.grid(g))))

    def tuned_moments(self, g, t, prm, pos, mass, W, center=(0.0, 0.0, 0.0)):
        """W [(numr-1), nrows, 2] += moments of these particles; returns the number used."""
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        c = np.asarray(center, dtype=np.float64)
        self.lib.orc_tuned_moments.restype = ctypes.c_long
        return int(self.lib.orc_tuned_moments(ctypes.byref(self.grid(g)), t, ctypes.byref(prm),
                                              ctypes.c_long(len(m)), _dp(x), _dp(y), _dp(z), _dp(m), _dp(c), _dp(W)))

    def tuned_contract(self, g, t, W):
        coef = np.zeros(((g.lmax + 1) ** 2, g.nmax))
        self.lib.orc_tuned_contract(ctypes.byref(self.grid(g)), t, _dp(W), _dp(coef))
        return coef

    def tuned_project(self, g, t, coef):
        G = np.zeros((g.numr, (g.lmax + 1) ** 2))
        cf = np.ascontiguousarray(coef, dtype=np.float64)
        self.lib.orc_tuned_project(ctypes.byref(self.grid(g)), t, _dp(cf), _dp(G))
        return G

    def tuned_accel(self, g, t, prm, pos, G, center=(0.0, 0.0, 0.0)):
        n = pos.shape[0]
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        c = np.asarray(center, dtype=np.float64)
        ax, ay, az, pot = [np.zeros(n) for _ in range(4)]
        self.lib.orc_tuned_accel(ctypes.byref(self.grid(g)), t, ctypes.byref(prm), ctypes.c_long(n), _dp(x),
                                 _dp(y), _dp(z), _dp(c), _dp(G), _dp(ax), _dp(ay), _dp(az), _dp(pot))
        return np.stack([ax, ay, az], axis=1), pot

    # -- phase-space files and particle histograms (oracle/psp_oracle.c) ------------------------------------
    def psp_write(self, path, time, comps, real4=False):
        """comps: dicts with info (str), indexing (bool), mass, pos, vel, pot and optionally potext, indx, iattrib,
        dattrib -- Component::write_binary per component (src/Component.cc:2385-2454)."""
        n = [len(c["mass"]) for c in comps]
        def two_d(c, key, k, dtype):
            a = np.asarray(c[key], dtype=dtype) if c.get(key) is not None else np.zeros((k, 0), dtype=dtype)
            return a.reshape(k, a.size // k if k else (a.shape[1] if a.ndim == 2 else 0))
        ia = [two_d(c, "iattrib", k, np.int32) for c, k in zip(comps, n)]
        da = [two_d(c, "dattrib", k, np.float64) for c, k in zip(comps, n)]
        cat = lambda key, w: np.ascontiguousarray(np.concatenate(
            [np.asarray(c.get(key, np.zeros((k,) + w)), dtype=np.float64).reshape((k,) + w) for c, k in zip(comps, n)]))
        mass, pos, vel, pot, potext = cat("mass", ()), cat("pos", (3,)), cat("vel", (3,)), cat("pot", ()), cat("potext", ())
        indx = np.ascontiguousarray(np.concatenate(
            [np.asarray(c["indx"], dtype=np.uint64) if c.get("indx") is not None else np.arange(1, k + 1, dtype=np.uint64)
             for c, k in zip(comps, n)]))
        iav = np.ascontiguousarray(np.concatenate([a.reshape(-1) for a in ia])) if ia else np.zeros(0, np.int32)
        dav = np.ascontiguousarray(np.concatenate([a.reshape(-1) for a in da])) if da else np.zeros(0)
        arr_i = lambda v: (ctypes.c_int * len(v))(*[int(x) for x in v])
        infos = (ctypes.c_char_p * len(comps))(*[(c["info"] if c["info"].endswith("\n") else c["info"] + "\n").encode() for c in comps])
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = self.lib.orc_psp_write(str(path).encode(), ctypes.c_double(time), len(comps), arr_i(n),
                                    arr_i([a.shape[1] for a in ia]), arr_i([a.shape[1] for a in da]), infos,
                                    arr_i([int(bool(c.get("indexing", False))) for c in comps]), int(bool(real4)),
                                    vp(indx), vp(mass), vp(pos), vp(vel), vp(pot), vp(potext), vp(iav), vp(dav))
        assert rc == 0

    def psp_read(self, path, indexing, numprocs=1, myid=0):
        """PSPout constructor + firstParticle / nextParticle of every stanza -> (time, ntot, [dict per stanza])."""
        class Stanza(ctypes.Structure):
            _fields_ = [("nbod", ctypes.c_int), ("niatr", ctypes.c_int), ("ndatr", ctypes.c_int), ("ninfochar", ctypes.c_int),
                        ("r_size", ctypes.c_ulong), ("index_size", ctypes.c_ulong), ("pspos", ctypes.c_long),
                        ("info", ctypes.c_char * 8192)]
        st = (Stanza * 16)()
        time, ntot = ctypes.c_double(), ctypes.c_int()
        idx = (ctypes.c_int * 16)(*([int(bool(v)) for v in indexing] + [0] * (16 - len(indexing))))
        found = self.lib.orc_psp_scan(str(path).encode(), ctypes.byref(time), ctypes.byref(ntot), 16, st, idx)
        assert found >= 0, found
        out = []
        self.lib.orc_psp_read.restype = ctypes.c_long
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        for k in range(found):
            s = st[k]
            n = s.nbod
            indx, mass, pos, vel, pot = np.zeros(n, np.uint64), np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
            ia, da = np.zeros((n, s.niatr), np.int32), np.zeros((n, s.ndatr))
            got = self.lib.orc_psp_read(str(path).encode(), ctypes.byref(s), int(numprocs), int(myid), vp(indx), vp(mass),
                                        vp(pos), vp(vel), vp(pot), vp(ia), vp(da))
            out.append({"nbod": n, "niatr": s.niatr, "ndatr": s.ndatr, "r_size": int(s.r_size), "info": s.info.split(b"\0")[0].decode(),
                        "indx": indx[:got], "mass": mass[:got], "pos": pos[:got], "vel": vel[:got], "pot": pot[:got],
                        "iattrib": ia[:got], "dattrib": da[:got]})
        return time.value, ntot.value, out

    def histo2d(self, mass, pos, ctr, pmin, pmax, grid):
        m, p = np.ascontiguousarray(mass, dtype=np.float64), np.ascontiguousarray(pos, dtype=np.float64)
        g = [int(v) for v in grid]
        out = [np.zeros((max(g[a], 0), max(g[b], 0)), dtype=np.float32) for a, b in ((0, 1), (0, 2), (1, 2))]
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        self.lib.orc_histo2d(ctypes.c_long(len(m)), _dp(m), _dp(p), _dp(np.asarray(ctr, dtype=np.float64)),
                             _dp(np.asarray(pmin, dtype=np.float64)), _dp(np.asarray(pmax, dtype=np.float64)),
                             (ctypes.c_int * 3)(*g), vp(out[0]), vp(out[1]), vp(out[2]))
        return {k: o for k, o, (a, b) in zip(("xy", "xz", "yz"), out, ((0, 1), (0, 2), (1, 2))) if g[a] > 0 and g[b] > 0}

    def histo1d(self, mass, pos, ctr, rmax, nbins, proj):
        m, p = np.ascontiguousarray(mass, dtype=np.float64), np.ascontiguousarray(pos, dtype=np.float64)
        out = np.zeros(nbins, dtype=np.float32)
        self.lib.orc_histo1d(ctypes.c_long(len(m)), _dp(m), _dp(p), _dp(np.asarray(ctr, dtype=np.float64)),
                             ctypes.c_double(rmax), int(nbins), {"xy": 0, "xz": 1, "yz": 2, "r": 3}[proj],
                             out.ctypes.data_as(ctypes.c_void_p))
        return out

    def histo1dlog(self, mass, pos, vel, ctr, rmin, rmax, nbins):
        m, p, v = [np.ascontiguousarray(a, dtype=np.float64) for a in (mass, pos, vel)]
        out = [np.zeros(nbins, dtype=np.float32) for _ in range(3)]
        self.lib.orc_histo1dlog(ctypes.c_long(len(m)), _dp(m), _dp(p), _dp(v), _dp(np.asarray(ctr, dtype=np.float64)),
                                ctypes.c_double(rmin), ctypes.c_double(rmax), int(nbins),
                                *[o.ctypes.data_as(ctypes.c_void_p) for o in out])
        return tuple(out)

    def outlog_sums(self, mass, pos, vel, acc, pot):
        """The particle loop of OutLog::Run for one component (src/OutLog.cc:392-446) -> the dict Component.log_sums gives."""
        m, p, v, a, ph = [np.ascontiguousarray(x, dtype=np.float64) for x in (mass, pos, vel, acc, pot)]
        out = np.zeros(13)
        self.lib.orc_outlog_sums(ctypes.c_long(len(m)), _dp(m), _dp(p), _dp(v), _dp(a), _dp(ph), _dp(out))
        return {"mtot": out[0], "com": out[1:4].copy(), "cov": out[4:7].copy(), "angm": out[7:10].copy(),
                "ektot": out[10], "eptot": out[11], "clausius": out[12], "nbodies": len(m)}

    def outlog_row(self, tnow, sums, centers, used, wtime, precision=10):
        flat = np.ascontiguousarray(np.array([[s["mtot"], *s["com"], *s["cov"], *s["angm"], s["ektot"], s["eptot"], s["clausius"]]
                                              for s in sums], dtype=np.float64))
        n = len(sums)
        buf = ctypes.create_string_buffer(1 << 16)
        self.lib.orc_outlog_row(buf, len(buf), ctypes.c_double(tnow), n, _dp(flat),
                                (ctypes.c_int * n)(*[int(s["nbodies"]) for s in sums]), (ctypes.c_int * n)(*[int(u) for u in used]),
                                _dp(np.ascontiguousarray(centers, dtype=np.float64)), ctypes.c_double(wtime), int(precision))
        return buf.value.decode()

    def quadls(self, x, y):
        x, y = [np.ascontiguousarray(v, dtype=np.float64) for v in (x, y)]
        out = np.zeros(3)
        self.lib.orc_quadls(ctypes.c_int(len(x)), _dp(x), _dp(y), _dp(out))
        return out

    def pseudo_accel_fit(self, rows):
        """rows [n, 7] = {t, centre, axis} -> (accel, omega, domdt) (include/PseudoAccel.H:45-91)."""
        r = np.ascontiguousarray(rows, dtype=np.float64)
        a, o, d = np.zeros(3), np.zeros(3), np.zeros(3)
        self.lib.orc_pseudo_accel_fit(ctypes.c_int(len(r)), _dp(r), _dp(a), _dp(o), _dp(d))
        return a, o, d

    def get_pseudo_accel(self, center, axis, accel, omega, domdt, pos, vel):
        """src/Component.cc:4407-4427 for every row of pos / vel."""
        out = np.zeros_like(pos)
        a, o, d = [np.ascontiguousarray(v, dtype=np.float64) for v in (accel, omega, domdt)]
        for i in range(len(pos)):
            p, v, q = (np.ascontiguousarray(pos[i], dtype=np.float64),
                       np.ascontiguousarray(vel[i], dtype=np.float64), np.zeros(3))
            self.lib.orc_get_pseudo_accel(ctypes.c_int(center), ctypes.c_int(axis), _dp(a), _dp(o), _dp(d),
                                          _dp(p), _dp(v), _dp(q))
            out[i] = q
        return out

    def euler_slater(self, phi, theta, psi, body):
        out = np.zeros(9)
        self.lib.orc_euler_slater(ctypes.c_double(phi), ctypes.c_double(theta), ctypes.c_double(psi),
                                  ctypes.c_int(body), _dp(out))
        return out.reshape(3, 3)

    # -- pyEXP-literal twins (expui/BiorthBasis.cc:583-665, :818-926, :1804-1857) ---------------------
    def pyexp_sph_accumulate(self, g, prm, pos, mass):
        G = self.grid(g)
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        coef = np.zeros(((g.lmax + 1) ** 2, g.nmax))
        self.lib.orc_pyexp_sph_accumulate.restype = ctypes.c_long
        used = self.lib.orc_pyexp_sph_accumulate(ctypes.byref(G), ctypes.byref(prm), ctypes.c_long(len(m)),
                                                 _dp(x), _dp(y), _dp(z), _dp(m), _dp(coef))
        return coef, int(used)

    def pyexp_sph_accel(self, g, prm, coef, pos):
        G = self.grid(g)
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        cf = np.ascontiguousarray(coef, dtype=np.float64)
        acc = np.zeros((len(x), 3))
        with np.errstate(all="ignore"):
            self.lib.orc_pyexp_sph_accel(ctypes.byref(G), ctypes.byref(prm), _dp(cf), ctypes.c_long(len(x)),
                                         _dp(x), _dp(y), _dp(z), _dp(acc))
        return acc

    def pyexp_cyl_accumulate(self, g, pos, mass, **kw):
        G = self.cylgrid(g, **kw)
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        cosN, sinN = np.zeros((g.mmax + 1, g.norder)), np.zeros((g.mmax + 1, g.norder))
        self.lib.orc_pyexp_cyl_accumulate.restype = ctypes.c_long
        n = self.lib.orc_pyexp_cyl_accumulate(ctypes.byref(G), ctypes.c_long(len(m)), _dp(x), _dp(y), _dp(z),
                                              _dp(m), _dp(cosN), _dp(sinN))
        return cosN, sinN, int(n)

    def pyexp_cyl_accel(self, g, cosN, sinN, pos, **kw):
        G = self.cylgrid(g, **kw)
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        cc, ss = [np.ascontiguousarray(a, dtype=np.float64) for a in (cosN, sinN)]
        acc = np.zeros((len(x), 3))
        self.lib.orc_pyexp_cyl_accel(ctypes.byref(G), _dp(cc), _dp(ss), ctypes.c_long(len(x)), _dp(x), _dp(y),
                                     _dp(z), _dp(acc))
        return acc

    def sph_fields(self, g, prm, coef, c1, c2, c3, coord="cartesian"):
        """pyEXP field evaluation (expui/BiorthBasis.cc:711-816, :930-958) -> [n, 9]."""
        G = self.grid(g)
        a, b, c = [np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64) for v in (c1, c2, c3)]
        cf = np.ascontiguousarray(coef, dtype=np.float64)
        out = np.zeros((len(a), 9))
        code = {"spherical": 0, "cylindrical": 1, "cartesian": 2}[coord]
        self.lib.orc_pyexp_sph_fields(ctypes.byref(G), ctypes.byref(prm), _dp(cf),
                                      ctypes.c_long(len(a)), _dp(a), _dp(b), _dp(c),
                                      ctypes.c_int(code), _dp(out))
        return out

    def sph_accel(self, g, prm, pos, coef, center=(0.0, 0.0, 0.0)):
        G = self.grid(g)
        n = pos.shape[0]
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        c = np.asarray(center, dtype=np.float64)
        cf = np.ascontiguousarray(coef, dtype=np.float64)
        ax, ay, az, pot = [np.zeros(n) for _ in range(4)]
        self.lib.orc_sph_accel(ctypes.byref(G), ctypes.byref(prm), ctypes.c_long(n), _dp(x),
                               _dp(y), _dp(z), _dp(c), _dp(cf), _dp(ax), _dp(ay), _dp(az),
                               _dp(pot))
        return np.stack([ax, ay, az], axis=1), pot

    def sph_accel_pseudo(self, g, prm, pos, coef, pseudo, center=(0.0, 0.0, 0.0)):
        """orc_sph_accel with Component::AddAcc's pseudo-acceleration (one row per particle) subtracted
        on each of the reference's five AddAcc calls (src/SphericalBasis.cc:1645-1651)."""
        G = self.grid(g)
        n = pos.shape[0]
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        c = np.asarray(center, dtype=np.float64)
        cf = np.ascontiguousarray(coef, dtype=np.float64)
        ps = np.ascontiguousarray(pseudo, dtype=np.float64)
        ax, ay, az, pot = [np.zeros(n) for _ in range(4)]
        self.lib.orc_sph_accel_pseudo(ctypes.byref(G), ctypes.byref(prm), ctypes.c_long(n), _dp(x), _dp(y),
                                      _dp(z), _dp(c), _dp(cf), _dp(ps), _dp(ax), _dp(ay), _dp(az), _dp(pot))
        return np.stack([ax, ay, az], axis=1), pot

    def sph_step(self, g, prm, dt, pos, vel, acc, mass, center=(0.0, 0.0, 0.0)):
        """In-place KDK step on copies; returns (pos, vel, acc, pot, coef)."""
        G = self.grid(g)
        n = pos.shape[0]
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64).copy() for k in range(3)]
        vx, vy, vz = [np.ascontiguousarray(vel[:, k], dtype=np.float64).copy() for k in range(3)]
        ax, ay, az = [np.ascontiguousarray(acc[:, k], dtype=np.float64).copy() for k in range(3)]
        pot = np.zeros(n)
        m = np.ascontiguousarray(mass, dtype=np.float64)
        c = np.asarray(center, dtype=np.float64)
        coef = np.zeros(((g.lmax + 1) ** 2, g.nmax))
        self.lib.orc_sph_step(ctypes.byref(G), ctypes.byref(prm), ctypes.c_long(n),
                              ctypes.c_double(dt), _dp(x), _dp(y), _dp(z), _dp(vx), _dp(vy),
                              _dp(vz), _dp(ax), _dp(ay), _dp(az), _dp(pot), _dp(m), _dp(c),
                              _dp(coef))
        return (np.stack([x, y, z], 1), np.stack([vx, vy, vz], 1), np.stack([ax, ay, az], 1),
                pot, coef)

    # -- cylindrical hot path ----------------------------------------------------------------
    def cylgrid(self, g, rcylmax=None, EVEN_M=False) -> _CylGrid:
        tab = np.ascontiguousarray(g.tab, dtype=np.float64)
        self._keep.append(tab)
        return _CylGrid(g.mmax, g.norder, g.numx, g.numy, g.cmapr, g.cmapz, int(EVEN_M), g.ascale,
                        g.hscale, g.rtable, g.xmin, g.dx, g.ymin, g.dy,
                        g.rmax if rcylmax is None else rcylmax, g.ascale, _dp(tab))

    def cyl_accumulate(self, g, pos, mass, center=(0.0, 0.0, 0.0), **kw):
        G = self.cylgrid(g, **kw)
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        c = np.asarray(center, dtype=np.float64)
        cosN = np.zeros((g.mmax + 1, g.norder))
        sinN = np.zeros((g.mmax + 1, g.norder))
        cylmass = ctypes.c_double(0.0)
        self.lib.orc_cyl_accumulate.restype = ctypes.c_long
        used = self.lib.orc_cyl_accumulate(ctypes.byref(G), ctypes.c_long(len(m)), _dp(x), _dp(y),
                                           _dp(z), _dp(m), _dp(c), _dp(cosN), _dp(sinN),
                                           ctypes.byref(cylmass))
        return cosN, sinN, int(used), cylmass.value

    def cyl_covariance(self, g, pos, mass, sampT, seq=None, acc=None, **kw):
        """The `covar` branch of EmpCylSL::accumulate (exputil/EmpCylSL.cc:4049-4146) -> dict(counts,
        masses, mean [T, mmax+1, norder] complex, covr [T, mmax+1, norder, norder] complex, used)."""
        G = self.cylgrid(g, **kw)
        if acc is None:
            acc = {"counts": np.zeros(sampT, dtype=np.int64), "masses": np.zeros(sampT),
                   "vc2": np.zeros((sampT, g.mmax + 1, g.norder, 2)),
                   "mv2": np.zeros((sampT, g.mmax + 1, g.norder, g.norder, 2)), "used": 0}
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        m = np.ascontiguousarray(mass, dtype=np.float64)
        sq = None if seq is None else np.ascontiguousarray(seq, dtype=np.int64)
        self.lib.orc_cyl_covariance.restype = ctypes.c_long
        n = self.lib.orc_cyl_covariance(ctypes.byref(G), ctypes.c_long(len(m)), _dp(x), _dp(y), _dp(z), _dp(m),
                                        None if sq is None else sq.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.c_int(sampT), acc["counts"].ctypes.data_as(ctypes.c_void_p),
                                        _dp(acc["masses"]), _dp(acc["vc2"]), _dp(acc["mv2"]))
        acc["used"] += int(n)
        acc["mean"] = acc["vc2"][..., 0] + 1j * acc["vc2"][..., 1]
        acc["covr"] = acc["mv2"][..., 0] + 1j * acc["mv2"][..., 1]
        return acc

    def emp_legendre(self, lmax, x):
        """EmpCylSL::legendre_R (exputil/EmpCylSL.cc:6493-6569) -> p[l, m]."""
        p = np.zeros((lmax + 1, lmax + 1))
        self.lib.orc_emp_legendre_R(ctypes.c_int(lmax), ctypes.c_double(x), _dp(p))
        return p

    def cyl_accumulate_eof(self, sl, m, ascale, rtable, rmax2, pos, mass):
        """EmpCylSL::accumulate_eof (exputil/EmpCylSL.cc:2686-2862) of every particle under Cylinder's cut, harmonic m ->
        (SC, SS [rank, rank], used, cylmass); sl: the helper SLGridSph."""
        G = self.grid(sl)
        rank = sl.nmax * (sl.lmax - m + 1)
        SC, SS = np.zeros((rank, rank)), np.zeros((rank, rank))
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        mm = np.ascontiguousarray(mass, dtype=np.float64)
        cm = ctypes.c_double(0.0)
        self.lib.orc_cyl_accumulate_eof.restype = ctypes.c_long
        used = self.lib.orc_cyl_accumulate_eof(ctypes.byref(G), ctypes.c_int(m), ctypes.c_double(ascale), ctypes.c_double(rtable),
                                               ctypes.c_double(rmax2), ctypes.c_long(len(mm)), _dp(x), _dp(y), _dp(z), _dp(mm),
                                               _dp(SC), _dp(SS), ctypes.byref(cm))
        return SC, SS, int(used), cm.value

    def cyl_accel(self, g, pos, cosN, sinN, cylmass, center=(0.0, 0.0, 0.0), **kw):
        G = self.cylgrid(g, **kw)
        n = pos.shape[0]
        x, y, z = [np.ascontiguousarray(pos[:, k], dtype=np.float64) for k in range(3)]
        c = np.asarray(center, dtype=np.float64)
        cc = np.ascontiguousarray(cosN, dtype=np.float64)
        ss = np.ascontiguousarray(sinN, dtype=np.float64)
        ax, ay, az, pot = [np.zeros(n) for _ in range(4)]
        self.lib.orc_cyl_accel(ctypes.byref(G), ctypes.c_long(n), _dp(x), _dp(y), _dp(z), _dp(c),
                               _dp(cc), _dp(ss), ctypes.c_double(cylmass), _dp(ax), _dp(ay),
                               _dp(az), _dp(pot))
        return np.stack([ax, ay, az], axis=1), pot

    # -- the NOISE mode (oracle/noise_oracle.cc) --------------------------------------------------
    def sph_compute_rms_coefs(self, g, rtab, dtab, scale=1.0, numg=100):
        """SphericalBasis::compute_rms_coefs (src/SphericalBasis.cc:2108-2147) -> (meanC[nmax], rmsC[lmax+1, nmax])"""
        from numpy.polynomial import legendre as npleg
        x, w = npleg.leggauss(int(numg))
        kn, wt = np.ascontiguousarray(0.5 * (x + 1.0)), np.ascontiguousarray(0.5 * w)
        G = self.grid(g)
        rt, dt = np.ascontiguousarray(rtab, dtype=np.float64), np.ascontiguousarray(dtab, dtype=np.float64)
        meanC, rmsC = np.zeros(g.nmax), np.zeros((g.lmax + 1, g.nmax))
        self.lib.orc_sph_compute_rms_coefs(ctypes.byref(G), ctypes.c_double(scale), ctypes.c_int(len(rt)), _dp(rt), _dp(dt),
                                           ctypes.c_int(int(numg)), _dp(kn), _dp(wt), _dp(meanC), _dp(rmsC))
        return meanC, rmsC

    def noise_create(self, lmax, nmax, meanC, rmsC, noiseN, seedN):
        self.lib.orc_noise_create.restype = ctypes.c_void_p
        m, r = np.ascontiguousarray(meanC, dtype=np.float64), np.ascontiguousarray(rmsC, dtype=np.float64)
        return ctypes.c_void_p(self.lib.orc_noise_create(ctypes.c_int(lmax), ctypes.c_int(nmax), _dp(m), _dp(r),
                                                         ctypes.c_double(noiseN), ctypes.c_uint(int(seedN) & 0xffffffff)))

    def noise_update(self, handle, lmax, nmax):
        """one SphericalBasis::update_noise (src/SphericalBasis.cc:2150-2210) -> [(lmax+1)^2, nmax]"""
        out = np.zeros(((lmax + 1) ** 2, nmax))
        self.lib.orc_noise_update(handle, _dp(out))
        return out

    def cyl_fields(self, g, cosN, sinN, c1, c2, c3, coord="cartesian", **kw):
        """pyEXP Cylindrical::sph_eval / cyl_eval / crt_eval (expui/BiorthBasis.cc:1749-1849)."""
        G = self.cylgrid(g, **kw)
        a, b, c = [np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64) for v in (c1, c2, c3)]
        cc = np.ascontiguousarray(cosN, dtype=np.float64)
        ss = np.ascontiguousarray(sinN, dtype=np.float64)
        dens = np.ascontiguousarray(g.dens, dtype=np.float64)
        out = np.zeros((len(a), 9))
        code = {"spherical": 0, "cylindrical": 1, "cartesian": 2}[coord]
        self.lib.orc_pyexp_cyl_fields(ctypes.byref(G), _dp(dens), _dp(cc), _dp(ss),
                                      ctypes.c_long(len(a)), _dp(a), _dp(b), _dp(c),
                                      ctypes.c_int(code), _dp(out))
        return out

    def sph_multistep_init(self, g, prm, multistep, dtime, dynfrac, shiftlevl, pos, vel, mass,
                           center=(0.0, 0.0, 0.0)):
        """begin_run for one spherical component; returns a state dict."""
        G = self.grid(g)
        n = pos.shape[0]
        st = {k: np.ascontiguousarray(pos[:, j], dtype=np.float64).copy()
              for j, k in enumerate("xyz")}
        st.update({"v" + k: np.ascontiguousarray(vel[:, j], dtype=np.float64).copy()
                   for j, k in enumerate("xyz")})
        for k in ("ax", "ay", "az", "pot"):
            st[k] = np.zeros(n)
        st["mass"] = np.ascontiguousarray(mass, dtype=np.float64)
        st["level"] = np.zeros(n, dtype=np.int32)
        ncoef = (g.lmax + 1) ** 2 * g.nmax
        st["coefN"] = np.zeros((multistep + 1, ncoef))
        st["coefL"] = np.zeros((multistep + 1, ncoef))
        st["coef"] = np.zeros(ncoef)
        st["center"] = np.asarray(center, dtype=np.float64)
        st["dyn"] = np.asarray(dynfrac, dtype=np.float64)
        st.update(multistep=multistep, dtime=dtime, shiftlevl=shiftlevl, this_step=0)
        self.lib.orc_sph_multistep_init(
            ctypes.byref(G), ctypes.byref(prm), ctypes.c_int(multistep), ctypes.c_double(dtime),
            _dp(st["dyn"]), ctypes.c_int(shiftlevl), ctypes.c_long(n), _dp(st["x"]), _dp(st["y"]),
            _dp(st["z"]), _dp(st["vx"]), _dp(st["vy"]), _dp(st["vz"]), _dp(st["ax"]), _dp(st["ay"]),
            _dp(st["az"]), _dp(st["pot"]), _dp(st["mass"]), st["level"].ctypes.data_as(c_int_p),
            _dp(st["center"]), _dp(st["coefN"]), _dp(st["coefL"]), _dp(st["coef"]))
        return st

    def sph_multistep_step(self, g, prm, st):
        G = self.grid(g)
        n = st["x"].size
        nsw = ctypes.c_long(0)
        self.lib.orc_sph_multistep_step(
            ctypes.byref(G), ctypes.byref(prm), ctypes.c_int(st["multistep"]),
            ctypes.c_double(st["dtime"]), _dp(st["dyn"]), ctypes.c_int(st["shiftlevl"]),
            ctypes.c_long(n), _dp(st["x"]), _dp(st["y"]), _dp(st["z"]), _dp(st["vx"]), _dp(st["vy"]),
            _dp(st["vz"]), _dp(st["ax"]), _dp(st["ay"]), _dp(st["az"]), _dp(st["pot"]),
            _dp(st["mass"]), st["level"].ctypes.data_as(c_int_p), _dp(st["center"]),
            _dp(st["coefN"]), _dp(st["coefL"]), ctypes.c_int(st["this_step"]), _dp(st["coef"]),
            ctypes.byref(nsw))
        st["this_step"] += 1
        return int(nsw.value)

    # -- multistep ---------------------------------------------------------------------
    def mstep_tables(self, multistep):
        t = ctypes.c_void_p(self.lib.orc_mstep_create(ctypes.c_int(multistep)))
        Mstep = 1 << multistep
        mintvl = np.zeros(multistep + 1, dtype=np.int32)
        mfirst = np.zeros(Mstep + 1, dtype=np.int32)
        mactive = np.zeros((Mstep + 1, multistep + 1), dtype=np.int32)
        dstepL = np.zeros((multistep + 1, Mstep), dtype=np.int32)
        dstepN = np.zeros((multistep + 1, Mstep), dtype=np.int32)
        ip = lambda a: a.ctypes.data_as(c_int_p)
        self.lib.orc_mstep_export(t, ip(mintvl), ip(mfirst), ip(mactive), ip(dstepL), ip(dstepN))
        self.lib.orc_mstep_free(t)
        return dict(multistep=multistep, Mstep=Mstep, mintvl=mintvl, mfirst=mfirst,
                    mactive=mactive, dstepL=dstepL, dstepN=dstepN)

    def mstep_combine(self, multistep, mdrft, coefL, coefN):
        t = ctypes.c_void_p(self.lib.orc_mstep_create(ctypes.c_int(multistep)))
        coefL = np.ascontiguousarray(coefL, dtype=np.float64)
        coefN = np.ascontiguousarray(coefN, dtype=np.float64)
        ncoef = coefL[0].size
        out = np.zeros(ncoef)
        self.lib.orc_mstep_combine(t, ctypes.c_int(mdrft), ctypes.c_long(ncoef), _dp(coefL),
                                   _dp(coefN), _dp(out))
        self.lib.orc_mstep_free(t)
        return out.reshape(coefL[0].shape)

    def level_select(self, dtime, multistep, mfirst_mdrft, cur, shiftlevl, dynfrac, scale, v, a,
                     pot):
        dyn = np.asarray(dynfrac, dtype=np.float64)
        v = np.asarray(v, dtype=np.float64)
        a = np.asarray(a, dtype=np.float64)
        dtreq = ctypes.c_double(0.0)
        lev = self.lib.orc_level_select(ctypes.c_double(dtime), ctypes.c_int(multistep),
                                        ctypes.c_int(mfirst_mdrft), ctypes.c_int(cur),
                                        ctypes.c_int(shiftlevl), _dp(dyn),
                                        ctypes.c_double(scale), _dp(v), _dp(a),
                                        ctypes.c_double(pot), ctypes.byref(dtreq))
        return int(lev), dtreq.value

    # -- pyEXP.basis beyond accumulate / getAccel / getFields (oracle/pyexp_oracle.c) ----------------------------
    @staticmethod
    def legequad(knots):
        """LegeQuad(knots): Gauss-Legendre knots and weights on [0, 1]."""
        x, w = npleg.leggauss(knots)
        return np.ascontiguousarray(0.5 * (x + 1.0)), np.ascontiguousarray(0.5 * w)

    def pyexp_sph_get_basis(self, g, logxmin, logxmax, numgrid):
        out = np.zeros((3, g.lmax + 1, g.nmax, numgrid))
        self.lib.orc_pyexp_sph_get_basis(ctypes.byref(self.grid(g)), ctypes.c_double(logxmin), ctypes.c_double(logxmax),
                                         ctypes.c_int(numgrid), _dp(out))
        return out

    def pyexp_sph_quad_points(self, rmin, rmax, rmapping, knots):
        k, w = self.legequad(knots)
        xyz = np.zeros((knots ** 3, 3))
        self.lib.orc_pyexp_sph_quad_points(ctypes.c_double(rmin), ctypes.c_double(rmax), ctypes.c_double(rmapping),
                                           ctypes.c_int(knots), _dp(k), _dp(w), _dp(xyz))
        return xyz

    def pyexp_sph_make_from_function(self, g, rmin, rmax, rmapping, knots, fv, potential=False):
        k, w = self.legequad(knots)
        fv = np.ascontiguousarray(fv, dtype=np.float64)
        assert fv.size == knots ** 3
        mat = np.zeros(((g.lmax + 1) * (g.lmax + 2) // 2, g.nmax, 2))
        self.lib.orc_pyexp_sph_make_from_function(ctypes.byref(self.grid(g)), ctypes.c_double(rmin), ctypes.c_double(rmax),
                                                  ctypes.c_double(rmapping), ctypes.c_int(knots), _dp(k), _dp(w), _dp(fv),
                                                  ctypes.c_int(int(potential)), _dp(mat))
        return mat[..., 0] + 1j * mat[..., 1]

    def pyexp_sph_compute_quadrature(self, rmin, rmax, rmapping, knots, fv):
        k, w = self.legequad(knots)
        fv = np.ascontiguousarray(fv, dtype=np.float64)
        self.lib.orc_pyexp_sph_compute_quadrature.restype = ctypes.c_double
        return float(self.lib.orc_pyexp_sph_compute_quadrature(ctypes.c_double(rmin), ctypes.c_double(rmax),
                                                               ctypes.c_double(rmapping), ctypes.c_int(knots), _dp(k),
                                                               _dp(w), _dp(fv)))

    def _cyl_dens(self, g):
        d = np.ascontiguousarray(g.dens, dtype=np.float64)
        self._keep.append(d)
        return d

    def cyl_get_all(self, g, cylmass, mm, nn, R, z, phi, **kw):
        out = np.zeros(5)
        self.lib.orc_cyl_get_all(ctypes.byref(self.cylgrid(g, **kw)), _dp(self._cyl_dens(g)), ctypes.c_double(cylmass),
                                 ctypes.c_int(mm), ctypes.c_int(nn), ctypes.c_double(R), ctypes.c_double(z),
                                 ctypes.c_double(phi), _dp(out))
        return out

    def pyexp_cyl_get_basis(self, g, cylmass, xmin, xmax, numR, zmin, zmax, numZ, linear=True, **kw):
        out = np.zeros((4, g.mmax + 1, g.norder, numR, numZ))
        self.lib.orc_pyexp_cyl_get_basis(ctypes.byref(self.cylgrid(g, **kw)), _dp(self._cyl_dens(g)),
                                         ctypes.c_double(cylmass), ctypes.c_double(xmin), ctypes.c_double(xmax),
                                         ctypes.c_int(numR), ctypes.c_double(zmin), ctypes.c_double(zmax),
                                         ctypes.c_int(numZ), ctypes.c_int(int(linear)), _dp(out))
        return out

    def cyl_orthocheck(self, g, **kw):
        out = np.zeros((g.mmax + 1, g.norder, g.norder))
        self.lib.orc_cyl_orthocheck(ctypes.byref(self.cylgrid(g, **kw)), _dp(self._cyl_dens(g)), _dp(out))
        return out

    def pyexp_cyl_quad_points(self, g, rmin, knots, **kw):
        k, _ = self.legequad(knots)
        xyz = np.zeros((knots ** 3, 3))
        self.lib.orc_pyexp_cyl_quad_points(ctypes.byref(self.cylgrid(g, **kw)), ctypes.c_double(rmin), ctypes.c_int(knots),
                                           _dp(k), _dp(xyz))
        return xyz

    def pyexp_cyl_make_from_function(self, g, rmin, knots, fv, potential=False, **kw):
        k, w = self.legequad(knots)
        fv = np.ascontiguousarray(fv, dtype=np.float64)
        assert fv.size == knots ** 3
        mat = np.zeros((g.mmax + 1, g.norder, 2))
        self.lib.orc_pyexp_cyl_make_from_function(ctypes.byref(self.cylgrid(g, **kw)), _dp(self._cyl_dens(g)),
                                                  ctypes.c_double(rmin), ctypes.c_int(knots), _dp(k), _dp(w), _dp(fv),
                                                  ctypes.c_int(int(potential)), _dp(mat))
        return mat[..., 0] + 1j * mat[..., 1]

    def pyexp_cyl_compute_quadrature(self, g, rmin, knots, fv, **kw):
        k, w = self.legequad(knots)
        fv = np.ascontiguousarray(fv, dtype=np.float64)
        self.lib.orc_pyexp_cyl_compute_quadrature.restype = ctypes.c_double
        return float(self.lib.orc_pyexp_cyl_compute_quadrature(ctypes.byref(self.cylgrid(g, **kw)), ctypes.c_double(rmin),
                                                               ctypes.c_int(knots), _dp(k), _dp(w), _dp(fv)))

    # -- CPU baseline in the reference's data structure (oracle/refstruct_cpu.c) --------------------------------
    def refstruct(self, mass, pos, vel):
        m = np.ascontiguousarray(mass, dtype=np.float64)
        p = np.ascontiguousarray(pos, dtype=np.float64)
        v = np.ascontiguousarray(vel, dtype=np.float64)
        self.lib.orc_refstruct_create.restype = ctypes.c_void_p
        return ctypes.c_void_p(self.lib.orc_refstruct_create(ctypes.c_long(len(m)), _dp(m), _dp(p), _dp(v)))

    def refstruct_field(self, rs, g, prm, nthreads):
        self.lib.orc_refstruct_field.restype = ctypes.c_long
        return int(self.lib.orc_refstruct_field(rs, ctypes.byref(self.grid(g)), ctypes.byref(prm), ctypes.c_int(nthreads)))

    def refstruct_step(self, rs, g, prm, dt, nthreads, G=None):
        self.lib.orc_refstruct_step.restype = ctypes.c_long
        return int(self.lib.orc_refstruct_step(rs, ctypes.byref(G if G is not None else self.grid(g)), ctypes.byref(prm),
                                               ctypes.c_double(dt), ctypes.c_int(nthreads)))

    def refstruct_get(self, rs, n, ncoef=None):
        pos, vel, acc, pot = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
        self.lib.orc_refstruct_get(rs, _dp(pos), _dp(vel), _dp(acc), _dp(pot))
        out = {"pos": pos, "vel": vel, "acc": acc, "pot": pot}
        if ncoef:
            self.lib.orc_refstruct_coef.restype = c_double_p
            out["coef"] = np.ctypeslib.as_array(self.lib.orc_refstruct_coef(rs), shape=(ncoef,)).copy()
        return out

    def refstruct_free(self, rs):
        self.lib.orc_refstruct_free(rs)
