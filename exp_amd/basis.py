"""``pyEXP.basis``-shaped front end over the C ABI (SURVEY.md section 8 row a21 / boundary B2).

Mirrors the calls of ``BasisClasses::BiorthBasis`` that ``pyEXP/BasisWrappers.cc:984-2260``
exposes and that the reference's tests use (``tests/Halo/createCoefs.py:72-96``,
``tests/Halo/sph_basis.py``, ``tests/Disk/cyl_basis.py``):

    basis = Basis.factory(yaml_text)              # id: sphereSL | cylinder
    coefs = basis.createFromArray(mass, pos, time=0.0, center=[0,0,0])
    basis.initFromArray(); basis.addFromArray(m, p); coefs = basis.makeFromArray(time)
    basis.set_coefs(coefs); acc = basis.getAccel(pos)            # [N,3]
    basis.orthoCheck(); basis.cacheInfo(cachename)

Semantics follow the pyEXP twins, not the n-body classes, where they differ
(``expui/BiorthBasis.cc:583-665``, ``:818-926``, ``:4585-4757``): array layout rules of
``addFromArray`` (rows vs columns, ``posvelrows``), rotation/centre applied as
``rot @ (pos - ctr)``, NO exterior multipole continuation in ``getAccel``, complex
(l, m>=0) x n coefficient packing of ``load_coefs`` / ``set_coefs`` (``:482-581``).
All particle work runs on the GPU through libexp_amd.so; this file only parses the YAML keys
(same names as the reference), builds / caches the tables and marshals arrays.
"""
from __future__ import annotations

import dataclasses
import math
import os
from typing import Optional

import numpy as np
import yaml

from .empcyl import EmpCylGrid, build_empcyl
from .models import TableModel
from .runtime import Component, Context, Cylinder, SphereSL
from .slgrid import SLGridSph, build_slgrid

# expui/BiorthBasis.cc valid_keys of Spherical / SphericalSL
SPH_KEYS = {"rmapping", "cmap", "Lmax", "dof", "npca", "npca0", "pcavar", "pcadiag", "pcavtk",
            "subsamp", "hexp", "snr", "samplesz", "vtkfreq", "tksmooth", "tkcum", "tk_type", "nmax",
            "modelname", "cachename", "scale", "rmin", "rmax", "numr", "nums", "N1", "N2", "NO_L0",
            "NO_L1", "EVEN_L", "EVEN_M", "M0_ONLY", "diverge", "dfac", "dtime", "noff", "mtype"}
CYL_KEYS = {"tk_type", "rcylmin", "rcylmax", "acyl", "hcyl", "sech2", "snr", "evcut", "nmaxfid",
            "lmaxfid", "mmax", "mlim", "nmax", "ncylodd", "ncylnx", "ncylny", "ncylr", "ncylorder",
            "ncylrecomp", "npca", "npca0", "nvtk", "cachename", "oldcache", "eof_file", "override",
            "samplesz", "rnum", "pnum", "tnum", "ashift", "expcond", "ignore", "deproject", "logr",
            "pcavar", "pcaeof", "pcavtk", "pcadiag", "subsamp", "try_cache", "density", "EVEN_M",
            "cmap", "cmapr", "cmapz", "aratio", "hratio", "dweight", "Mfac", "HERNA", "rwidth",
            "rfactor", "rtrunc", "rpow", "mtype", "dtype", "vflag", "self_consistent", "playback",
            "coefCompute", "coefMaster", "pyname", "dumpbasis"}

_default_ctx: Optional[Context] = None


def _ctx() -> Context:
    global _default_ctx
    if _default_ctx is None or _default_ctx.h is None:
        _default_ctx = Context(0)
    return _default_ctx



def _legequad(n: int):
    """``LegeQuad(n)`` (include/gaussQ.H:104-108, exputil/gaussQ.cc -> Jacobi.c): the Gauss-Legendre knots and weights on
    [0, 1] of makeFromFunction / computeQuadrature.  The reference finds them by Newton iteration on the Jacobi
    polynomials and lists them in descending order; these are the same numbers to 4e-15 (tests/test_ref_util.py against
    the reference's own gaussQ.cc), ascending -- the quadrature sums do not depend on the order."""
    x, w = np.polynomial.legendre.leggauss(int(n))
    return 0.5 * (x + 1.0), 0.5 * w


class _CoefStructAPI:
    """What pyEXP exposes of ``CoefClasses::CoefStruct`` (expui/CoefStruct.H:18-113; pyEXP/CoefWrappers.cc:720-960):
    the time / centre / orientation accessors, the flat data store (the matrix in column-major order, as Eigen maps
    it), ``deepcopy``, and the gravitational constant a set carries when no container owns it."""

    G = 1.0
    _owner = None                     # the Coefs container that holds the set (CoefStruct::C), or None

    @property
    def center(self):
        return self.ctr

    @property
    def orient(self):
        return self.rot

    def deepcopy(self):
        """new arrays, same owner (CoefStruct::copyfields copies the pointer to the container, expui/CoefStruct.cc:24-34)"""
        import copy
        owner, self._owner = self._owner, None
        try:
            ret = copy.deepcopy(self)
        finally:
            self._owner = owner
        ret._owner = owner
        return ret

    def zerodata(self) -> None:
        self.coefs = np.zeros_like(np.asarray(self.coefs))

    def getCoefTime(self) -> float:
        return self.time

    def setCoefTime(self, tval: float) -> None:
        self.time = float(tval)

    def getCoefCenter(self) -> np.ndarray:
        return self.ctr

    def setCoefCenter(self, mat) -> None:
        self.ctr = np.asarray(mat, dtype=np.float64).reshape(3).copy()

    def getCoefRotation(self) -> np.ndarray:
        return self.rot

    def setCoefRotation(self, mat) -> None:
        self.rot = np.asarray(mat, dtype=np.float64).reshape(3, 3).copy()

    def getCoefs(self) -> np.ndarray:
        return np.asarray(self.coefs, dtype=np.complex128).reshape(-1, order="F")

    def setCoefs(self, mat=None):
        """``setCoefs(vector)`` replaces the store (sizes must agree); ``setCoefs()`` hands out the coefficient array
        itself for in-place changes (the reference returns an Eigen::Ref to the store)."""
        if mat is None:
            self.coefs = np.ascontiguousarray(self.coefs, dtype=np.complex128)
            return self.coefs
        v = np.asarray(mat, dtype=np.complex128).reshape(-1)
        cur = np.asarray(self.coefs)
        if v.size != cur.size:
            raise ValueError("CoefStruct::setCoefs: coefficient vector size does not match")
        self.coefs = v.reshape(cur.shape, order="F").copy()
        return None

    def setGravConstant(self, G: float) -> None:
        self.G = float(G)

    def getGravConstant(self) -> float:
        return self._owner.getGravConstant() if self._owner is not None else self.G


@dataclasses.dataclass
class SphStruct(_CoefStructAPI):
    """``CoefClasses::SphStruct``: complex coefficients [(L+1)(L+2)/2, nmax] (m >= 0 only)."""
    lmax: int = 0
    nmax: int = 0
    scale: float = 1.0
    time: float = 0.0
    coefs: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros((0, 0), dtype=np.complex128))
    ctr: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(3))
    rot: np.ndarray = dataclasses.field(default_factory=lambda: np.eye(3))
    normed: bool = True
    geometry: str = "sphere"

    def create(self) -> None:
        """``SphStruct::create`` (expui/CoefStruct.cc:43-49): zeroed storage for the orders that are set."""
        if self.nmax <= 0:
            raise RuntimeError("SphStruct::create: nmax must be >0")
        self.coefs = np.zeros(((self.lmax + 1) * (self.lmax + 2) // 2, self.nmax), dtype=np.complex128)

    def assign(self, mat, lmax: int, nmax: int) -> None:
        """``SphStruct::assign`` (expui/CoefStruct.H:158-164)."""
        mat = np.asarray(mat, dtype=np.complex128)
        self.lmax, self.nmax = int(lmax), int(nmax)
        rows = (self.lmax + 1) * (self.lmax + 2) // 2
        if mat.shape != (rows, self.nmax):
            raise ValueError(f"SphStruct::assign: matrix of shape {mat.shape}, expected {(rows, self.nmax)}")
        self.coefs = mat.copy()


@dataclasses.dataclass
class CylStruct(_CoefStructAPI):
    """``CoefClasses::CylStruct``: complex coefficients [mmax+1, nmax] = cos + i sin."""
    mmax: int = 0
    nmax: int = 0
    time: float = 0.0
    coefs: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros((0, 0), dtype=np.complex128))
    ctr: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(3))
    rot: np.ndarray = dataclasses.field(default_factory=lambda: np.eye(3))
    geometry: str = "cylinder"

    def create(self) -> None:
        """``CylStruct::create`` (expui/CoefStruct.cc:35-41)."""
        if self.nmax <= 0:
            raise RuntimeError("CylStruct::create: nmax must be >0")
        self.coefs = np.zeros((self.mmax + 1, self.nmax), dtype=np.complex128)

    def assign(self, mat, mmax: int, nmax: int) -> None:
        """``CylStruct::assign`` (expui/CoefStruct.H:204-210)."""
        mat = np.asarray(mat, dtype=np.complex128)
        self.mmax, self.nmax = int(mmax), int(nmax)
        if mat.shape != (self.mmax + 1, self.nmax):
            raise ValueError(f"CylStruct::assign: matrix of shape {mat.shape}, expected {(self.mmax + 1, self.nmax)}")
        self.coefs = mat.copy()


class BiorthBasis:
    name = ""

    def __init__(self, conf: dict, ctx: Optional[Context] = None):
        self.conf = conf
        self.ctx = ctx or _ctx()
        self.coefctr = np.zeros(3)
        self.coefrot = np.eye(3)
        self.coefret = None
        self.coefindx = 0
        self._ftor = None
        self.pseudo = np.zeros(3)
        self.totalMass = 0.0
        self.t_accel, self.p_accel = np.zeros(0), np.zeros((0, 3))

    # -- array front end (expui/BiorthBasis.cc:4585-4757) -------------------------------------------
    def initFromArray(self, center=(0.0, 0.0, 0.0), rot=None) -> None:
        self.coefctr = np.asarray(center, dtype=np.float64).reshape(3)
        self.coefrot = np.eye(3) if rot is None else np.asarray(rot, dtype=np.float64).reshape(3, 3)
        self.reset_coefs()
        self.coefindx = 0
        self.coefret = True

    def setSelector(self, ftor) -> None:
        self._ftor = ftor

    def clrSelector(self) -> None:
        """``Basis::clrSelector`` (expui/BasisFactory.H:281; pyEXP/BasisWrappers.cc:1144)."""
        self._ftor = None

    # -- non-inertial frame (expui/BasisFactory.H:160-178, :286-310; expui/BasisFactory.cc:286-404) ---------
    Naccel = 0

    def setInertial(self) -> None:
        """``Basis::setInertial``: back to an inertial frame, the pseudo-acceleration is zero."""
        self.Naccel = 0
        self.pseudo = np.zeros(3)

    def usingNonInertial(self) -> bool:
        return self.Naccel > 0

    def setNonInertial(self, N: int, times_or_orient, pos=None) -> None:
        """``Basis::setNonInertial(N, times, pos)`` -- the expansion centre's track, one row per time --
        or ``setNonInertial(N, orient_file)`` -- the track read from an Orient log (src/Orient.cc:742-785):
        rows that do not start with ``#``, time in the first column, and, after two more columns, eight
        triples of which the LAST one read is kept as the position (the reference's loop reads all eight
        into the same vector, expui/BasisFactory.cc:329-337: with the current 33-column log that is the
        pseudo-acceleration triple; restated, not repaired)."""
        if isinstance(times_or_orient, (str, bytes, os.PathLike)) and pos is None:
            path = os.fspath(times_or_orient)
            if not os.path.exists(path):
                raise RuntimeError("Cannot open Orient file with centering data: " + str(path))
            times, centers = [], []
            with open(path) as f:
                for line in f:
                    if line.startswith("#"):
                        continue
                    tok = line.split()
                    if len(tok) < 3 + 24:                # the row must hold all eight triples
                        continue
                    times.append(float(tok[0]))
                    centers.append([float(v) for v in tok[3 + 21:3 + 24]])
            t, p = np.array(times), np.array(centers).reshape(-1, 3)
        else:
            t = np.asarray(times_or_orient, dtype=np.float64).reshape(-1)
            p = np.asarray(pos, dtype=np.float64)
            if t.size < 1:
                raise RuntimeError("Basis: setNonInertial: no times in time array")
            if p.ndim != 2 or t.size != p.shape[0]:
                raise RuntimeError("Basis::setNonInertial: size mismatch in time and position arrays")
        self.Naccel, self.t_accel, self.p_accel = int(N), t, p

    @staticmethod
    def _quadls_a(x, y) -> float:
        """leading coefficient of ``QuadLS`` (include/QuadLS.H:17-53), sums taken in the reference's order"""
        n = len(x)
        if n == 0:
            return 0.0
        sumx = sumy = sumxy = sumx2y = sumx2 = sumx3 = sumx4 = 0.0
        for xi, yi in zip(x, y):
            sumx += xi; sumy += yi; sumx2 += xi * xi; sumxy += xi * yi
            sumx2y += xi * xi * yi; sumx3 += xi * xi * xi; sumx4 += xi * xi * xi * xi
        Sxx, Sxy = sumx2 - sumx * sumx / n, sumxy - sumx * sumy / n
        Sxx2, Sx2y = sumx3 - sumx * sumx2 / n, sumx2y - sumx2 * sumy / n
        Sx2x2 = sumx4 - sumx2 * sumx2 / n
        denom = Sxx * Sx2x2 - Sxx2 * Sxx2
        return (Sx2y * Sxx - Sxy * Sxx2) / denom if abs(denom) > 0.0 else 0.0

    def currentAccel(self, time: float) -> np.ndarray:
        """``Basis::currentAccel`` (expui/BasisFactory.cc:358-401): twice the leading coefficient of a
        quadratic least-squares fit to ~Naccel track points around ``time``."""
        t, n = self.t_accel, len(self.t_accel)
        if n < 2 or time < t[0] - 0.5 * (t[1] - t[0]) or time > t[n - 1] + 0.5 * (t[n - 1] - t[n - 2]):
            raise RuntimeError(f"Basis::currentAccel: {time} is outside the range of the non-inertial DB "
                               f"[{t[0]}, {t[n - 1]}]")
        imax = int(np.searchsorted(t, time, side="left"))
        imax = min(n - 1, imax + self.Naccel // 2)
        imin = max(imax - self.Naccel, 0)
        tt = [float(v) for v in t[imin:imax + 1]]
        return np.array([2.0 * self._quadls_a(tt, [float(v) for v in self.p_accel[imin:imax + 1, k]])
                         for k in range(3)])

    def setNonInertialAccel(self, time: float) -> None:
        """``Basis::setNonInertialAccel`` (expui/BasisFactory.H:297-300)."""
        if self.Naccel > 0:
            self.pseudo = self.currentAccel(time)

    def getCenter(self) -> np.ndarray:
        return self.coefctr

    def getRotation(self) -> np.ndarray:
        return self.coefrot

    def getCoefficients(self):
        return self.coefret

    def getMass(self) -> float:
        """``BiorthBasis::getMass`` (expui/BiorthBasis.H:189): the mass accumulated on the grid."""
        return float(self.totalMass)

    def setCovarH5Compress(self, level: int, chunksize: int, shuffle: bool, szip: bool = False) -> None:
        """``Basis::setCovarH5Compress`` (expui/BasisFactory.H:351-358)."""
        if not getattr(self, "pcavar", False):
            raise RuntimeError("Basis::setCovarH5Compress: covariance storage not initialized")
        from . import h5cache
        h5cache.covar_set_compress(level, chunksize, shuffle, szip)

    def _layout(self, p: np.ndarray, posvelrows: bool):
        """the dimension-deduction rule of addFromArray (:4638-4655)"""
        rows, cols = p.shape
        if cols in (3, 6):
            if rows not in (3, 6):
                posvelrows = False
        if rows in (3, 6):
            if cols not in (3, 6):
                posvelrows = True
        if posvelrows:
            if rows < 3:
                raise RuntimeError("Basis::addFromArray: you must pass a position array with at "
                                   f"least three rows for x, y, z.  Yours has {rows}.")
            return p[:3].T, (p[3:6].T if rows == 6 else None)
        if cols < 3:
            raise RuntimeError("Basis::addFromArray: you must pass a position array with at "
                               f"least three columns for x, y, z.  Yours has {cols}.")
        return p[:, :3], (p[:, 3:6] if cols == 6 else None)

    def addFromArray(self, m, p, roundrobin: bool = True, posvelrows: bool = False) -> None:
        if not self.coefret:
            raise RuntimeError("Basis::addFromArray: you must initialize coefficient accumulation "
                               "with a call to Basis::initFromArray()")
        m = np.asarray(m, dtype=np.float64).reshape(-1)
        p = np.asarray(p, dtype=np.float64)
        if p.ndim != 2:
            p = np.stack([np.asarray(a, dtype=np.float64) for a in p])
        pos, vel = self._layout(p, posvelrows)
        seq = np.arange(len(m), dtype=np.uint32)      # the index accumulate() is handed (:4616-4738)
        if self._ftor is None:
            # the expansion frame rot (x - ctr) is applied on the device as the particles are uploaded
            self.coefindx += len(m)
            self._accumulate_batch(m, pos, seq, frame=(self.coefctr, self.coefrot))
            return
        pos = (pos - self.coefctr) @ self.coefrot.T
        v = np.zeros_like(pos) if vel is None else vel @ self.coefrot.T
        keep = np.array([bool(self._ftor(m[i], pos[i], v[i], self.coefindx + i))
                         for i in range(len(m))])
        pos, m, seq = pos[keep], m[keep], seq[keep]
        self.coefindx += len(m)
        self._accumulate_batch(m, pos, seq)

    def makeFromArray(self, time: float = 0.0):
        self.make_coefs()
        return self.load_coefs(time)

    def createFromArray(self, m, p, time: float = 0.0, center=(0.0, 0.0, 0.0), rot=None,
                        roundrobin: bool = True, posvelrows: bool = False):
        self.initFromArray(center, rot)
        self.addFromArray(m, p, roundrobin, posvelrows)
        return self.makeFromArray(time)

    def createFromReader(self, reader, center=(0.0, 0.0, 0.0), rot=None):
        """``BiorthBasis::createFromReader(reader, ctr, rot)`` (expui/BiorthBasis.cc:4517-4581; pyEXP/BasisWrappers.cc:
        1287): the coefficients of the particles a ``ParticleReader`` (``exp_amd.reader``) delivers -- its selected type,
        this rank's share -- at ``rot (x - ctr)``, stamped with the reader's time.  The selector functor sees the
        transformed position, the rotated velocity and the particle's own index; ``accumulate`` is handed that index too
        (the sub-sample of the covariance is chosen from it).  The reference walks the particles one at a time; here
        they go to the device in batches."""
        ctr = np.asarray(center, dtype=np.float64).reshape(3)
        R = np.eye(3) if rot is None else np.asarray(rot, dtype=np.float64).reshape(3, 3)
        self.coefctr, self.coefrot = ctr, R
        self.reset_coefs()
        a = reader.arrays()
        n, step = len(a["mass"]), 1 << 24
        # every batch ends in one all-reduce of the coefficient buffer: all ranks must issue the same number of them
        # (shares that straddle a multiple of 2^24 differently would hang or mis-sum) -- empty batches take part
        nbatch = max(1, -(-n // step))
        # ... agreed on the transport the coefficient all-reduce itself uses (the library's RCCL communicator or the
        # host's callback), so that hosts without a torch process group are covered too
        info = self.ctx.comm_info()
        if info["kind"] != "none":
            if info["kind"] == "callback" and info["nranks"] <= 1:
                import sys as _sys
                _dist = _sys.modules.get("torch.distributed")
                if _dist is not None and _dist.is_available() and _dist.is_initialized():
                    if _dist.get_world_size() > 1:
                        box = [None] * _dist.get_world_size()
                        _dist.all_gather_object(box, nbatch)
                        nbatch = max(box)
                else:
                    # a callback installed without its world and no process group to ask: a single-process host (e.g. a
                    # pass-through callback) -- this rank's batch count stands.  A multi-rank host must declare the world,
                    # or ranks whose shares straddle a multiple of 2^24 differently would issue unequal numbers of reductions
                    import warnings
                    warnings.warn("createFromReader: the context reduces through a callback but was not told its world "
                                  "(Context.set_allreduce(fn, nranks, rank)); assuming a single rank", RuntimeWarning,
                                  stacklevel=2)
            else:
                nbatch = int(self.ctx.allreduce_max(nbatch))
        sampT = int(getattr(self, "sampT", 0) or 0)
        for lo in range(0, nbatch * step, step):
            sl = slice(min(lo, n), min(lo + step, n))
            m = np.ascontiguousarray(a["mass"][sl], dtype=np.float64)
            # the particle's own index picks the cylinder's sub-sample (seq % sampT with an unsigned long index in the
            # reference, exputil/EmpCylSL.cc:4062-4075): reduced BEFORE the 32-bit cast so that indices >= 2^32 agree
            idx64 = np.asarray(a["indx"][sl]).astype(np.uint64)
            seq = (idx64 % np.uint64(sampT) if sampT > 1 else idx64 & np.uint64(0xffffffff)).astype(np.uint32)
            if self._ftor is None:
                self._accumulate_batch(m, np.asarray(a["pos"][sl], dtype=np.float64), seq, frame=(ctr, R))
                continue
            pos = (np.asarray(a["pos"][sl], dtype=np.float64) - ctr) @ R.T
            if self._ftor is not None:
                v = np.asarray(a["vel"][sl], dtype=np.float64) @ R.T
                idx = a["indx"][sl]
                keep = np.array([bool(self._ftor(m[i], pos[i], v[i], int(idx[i]))) for i in range(len(m))], dtype=bool)
                m, pos, seq = m[keep], pos[keep], seq[keep]
            self._accumulate_batch(m, pos, seq)
        self.make_coefs()
        cs = self.load_coefs(reader.CurrentTime())
        cs.ctr, cs.rot = ctr.copy(), R.copy()
        return cs

    def accumulate(self, x, y, z, mass, indx: int = 0) -> None:
        self._accumulate_batch(np.atleast_1d(np.float64(mass)),
                               np.array([[x, y, z]], dtype=np.float64), np.array([indx], dtype=np.uint32))

    def make_coefs(self) -> None:
        """single process: the MPI reduction of expui/BiorthBasis.cc:667-709 is the device
        all-reduce already applied per batch"""

    def getAccel(self, *args) -> np.ndarray:
        """getAccel(x, y, z) -> [3]; getAccel(xv, yv, zv) or getAccel(pos[N,3]) -> [N,3]
        (expui/BiorthBasis.H:251-266)"""
        if len(args) == 3:
            x, y, z = [np.atleast_1d(np.asarray(a, dtype=np.float64)) for a in args]
            pos = np.stack([x, y, z], axis=1)
            single = np.ndim(args[0]) == 0
            if not (x.shape == y.shape == z.shape):
                raise RuntimeError("BiorthBasis::getAccel: x, y, z vectors must be of the same size")
        else:
            pos = np.asarray(args[0], dtype=np.float64)
            single = False
            if pos.ndim != 2 or pos.shape[1] != 3:
                raise RuntimeError("BiorthBasis::getAccel: input array must have 3 columns")
        # "in centered coordinate system" (expui/BiorthBasis.H:219-266): computeAccel takes the point as it
        # is; centre and rotation are the caller's business (AccelFunc::evalaccel applies them, :4760-4790)
        acc = self._accel(np.ascontiguousarray(pos))
        return acc[0] if single else acc

    def getAccelArray(self, x, y, z) -> np.ndarray:
        """``getAccelArray(x, y, z)`` (pyEXP/BasisWrappers.cc:1616): the three-vector overload of getAccel
        under a name of its own -> [N, 3]."""
        return self.getAccel(np.atleast_1d(x), np.atleast_1d(y), np.atleast_1d(z))

    # ---- field evaluation (expui/BasisFactory.cc:218-234, expui/BiorthBasis.cc:71-97, :711-958) ----
    FIELD_LABELS = ["dens m=0", "dens m>0", "dens", "potl m=0", "potl m>0", "potl"]
    FORCE_LABELS = {"spherical": ["rad force", "mer force", "azi force"],
                    "cylindrical": ["rad force", "ver force", "azi force"],
                    "cartesian": ["x force", "y force", "z force"]}
    coordinates = "spherical"            # BasisFactory.H: default field type

    def setFieldType(self, coord_type: str) -> None:
        key = coord_type.strip().lower()
        if key not in self.FORCE_LABELS:
            raise RuntimeError(f"Basis: unknown coordinate type <{coord_type}>")
        self.coordinates = key

    def getFieldType(self) -> str:
        """``Basis::getFieldType`` (expui/BasisFactory.H:242; coordLabels, expui/BasisFactory.cc:16-20)."""
        return self.coordinates.capitalize()

    def getFieldLabels(self, ctype: Optional[str] = None):
        return self.FIELD_LABELS + self.FORCE_LABELS[(ctype or self.coordinates).lower()]

    # midplane evaluation (expui/BasisFactory.H:127-131, :283-288): only a disk basis acts on it
    midplane = False
    colh = 4.0

    def setMidplane(self, value: bool) -> None:
        self.midplane = bool(value)

    def setColumnHeight(self, value: float) -> None:
        self.colh = float(value)

    def __call__(self, x1, x2, x3, ctype: str = "spherical"):
        """Fields in the requested coordinates: (r, cos theta, phi) | (R, z, phi) | (x, y, z)
        -> the 9 values of getFieldLabels(ctype); arrays give [N, 9]."""
        single = np.ndim(x1) == 0
        out = self.force.fields(x1, x2, x3, ctype.lower())
        return out[0] if single else out

    def getFields(self, x, y, z):
        """Basis::getFields -> crt_eval(x, y, z) (expui/BasisFactory.cc:231-234)."""
        return self(x, y, z, "cartesian")

    def evaluate(self, x, y, z):
        return self.getFields(x, y, z), self.getFieldLabels("cartesian")

    def getFieldsCoefs(self, x, y, z, coefs):
        """``Basis::getFieldsCoefs`` (expui/BasisFactory.cc:236-265): the fields at one point for
        every coefficient set of a ``Coefs`` container -> ({label: array over time}, times)."""
        times = coefs.Times()
        labels = self.getFieldLabels(self.coordinates)
        ret = {s: np.zeros(len(times)) for s in labels}
        for i, t in enumerate(times):
            self.set_coefs(coefs.getCoefStruct(t))
            v = self.getFields(x, y, z)          # crt_eval, as the reference does
            for j, s in enumerate(labels):
                ret[s][i] = v[j]
        return ret, np.asarray(times)


class SphericalSL(BiorthBasis):
    """``sphereSL`` (expui/BiorthBasis.H:503): YAML keys Lmax, nmax, numr, rmin, rmax, scale,
    rmapping, cmap, modelname, cachename, NO_L0, NO_L1, EVEN_L, EVEN_M, M0_ONLY."""

    name = "sphereSL"

    def __init__(self, conf: dict, ctx: Optional[Context] = None):
        super().__init__(conf, ctx)
        bad = set(conf) - SPH_KEYS
        if bad:
            raise RuntimeError(f"Basis::Basis::Spherical: unmatched parameter(s) {sorted(bad)}")
        self.cmap = int(conf.get("cmap", 1))
        self.lmax = int(conf.get("Lmax", 6))
        self.nmax = int(conf.get("nmax", 18))
        self.rmap = float(conf.get("rmapping", 1.0))
        self.scale = float(conf.get("scale", 1.0))
        self.numr = int(conf.get("numr", 800))
        model_file = conf.get("modelname", "SLGridSph.model")
        self.cachename = conf.get("cachename", "")
        if not self.cachename:
            raise RuntimeError("SphericalSL requires a specified cachename in your YAML config")
        self.model_file = model_file
        self.model = TableModel(model_file)
        rmin = float(conf.get("rmin", 0.0))
        rmax = float(conf.get("rmax", np.finfo(np.float64).max))
        if "rmin" not in conf or rmin < self.model.rmin:
            rmin = self.model.rmin
        if "rmax" not in conf or rmax > self.model.rmax:
            rmax = self.model.rmax * 0.99
        self.rmin, self.rmax = rmin, rmax
        self.grid = self._load_or_build()
        flags = {k: bool(conf.get(k, False)) for k in ("NO_L0", "NO_L1", "EVEN_L", "EVEN_M")}
        flags["M0_only"] = bool(conf.get("M0_ONLY", False))
        self.force = SphereSL(self.ctx, self.grid, scale=self.scale, rmin=rmin, rmax=rmax, **flags)
        _lib_check = self.force.lib.exp_amd_sph_set_exterior(self.force.h, 0)   # pyEXP semantics
        assert _lib_check == 0
        # Spherical::accumulate applies none of the flags (expui/BiorthBasis.cc:583-665): with M0_ONLY the coefficients
        # it returns still hold every m, only the evaluation drops them
        _lib_check = self.force.lib.exp_amd_sph_set_accumulate_all_m(self.force.h, 1)
        assert _lib_check == 0
        # N1, N2: the radial window of the l >= 1 sums in computeAccel / sph_eval
        # (expui/BiorthBasis.cc:761, :780, :876, :894; the l = 0 term takes every n).  The reference
        # reads both keys with `.as<bool>()` (:264-265), so the only values a YAML file can give them
        # are booleans, stored as 0 / 1; anything else makes yaml-cpp throw, and is refused here too.
        self.N1, self.N2 = 0, np.iinfo(np.int32).max
        for key in ("N1", "N2"):
            if key in conf:
                v = conf[key]
                if isinstance(v, str) and v.strip().lower() in ("true", "yes", "on", "y", "false", "no", "off", "n"):
                    v = v.strip().lower() in ("true", "yes", "on", "y")
                if not isinstance(v, bool) and v not in (0, 1):
                    raise RuntimeError(f"Basis::Basis::Spherical: {key} is read as a boolean by the reference "
                                       f"(expui/BiorthBasis.cc:264-265); <{conf[key]}> does not convert")
                setattr(self, key, int(bool(v)))
        self.nrows = (self.lmax + 1) ** 2
        self.expcoef = np.zeros((self.nrows, self.nmax))
        self.used = 0
        self.orthoTest(200)
        # coefficient covariance by sub-sampling (expui/BiorthBasis.cc:271-274, :335)
        self.pcavar = bool(conf.get("pcavar", False))
        self.sampT = max(1, int(conf.get("subsamp", 100)))
        if self.pcavar:
            self.enableCoefCovariance(True, self.sampT)

    # -- tables / cache -----------------------------------------------------------------------------
    def _cache_path(self) -> str:
        return self.cachename if self.cachename.endswith(".npz") else self.cachename + ".npz"

    def _params(self) -> dict:
        return dict(lmax=self.lmax, nmax=self.nmax, numr=self.numr, cmap=self.cmap, rmin=self.rmin,
                    rmax=self.rmax, rmapping=self.rmap)

    def _load_or_build(self) -> SLGridSph:
        """EXP's own HDF5 cache ``cachename`` (exputil/SLGridMP2.cc:490-696) is read when it exists
        and matches -- a basis built by EXP drives the kernels as it is -- and written after a
        build; where the HDF5 shim is unavailable an ``.npz`` twin is used instead."""
        from . import h5cache
        if h5cache.available():
            if os.path.exists(self.cachename):
                try:
                    return h5cache.read_slgrid_cache(
                        self.cachename, self.model,
                        check=dict(lmax=self.lmax, nmax=self.nmax, numr=self.numr, cmap=self.cmap,
                                   rmin=float(self.rmin), rmax=float(self.rmax),
                                   rmapping=float(self.rmap)))
                except RuntimeError:
                    pass                                  # the reference rebuilds on any mismatch
            g = build_slgrid(self.model, self.lmax, self.nmax, numr=self.numr, rmin=self.rmin,
                             rmax=self.rmax, cmap=self.cmap, rmap=self.rmap)
            if os.path.exists(self.cachename):            # :631-648 back up, then write afresh
                os.replace(self.cachename, self.cachename + ".bak")
            h5cache.write_slgrid_cache(self.cachename, g, self.model_file)
            return g
        path = self._cache_path()
        if os.path.exists(path):
            g = SLGridSph.load(path)
            same = (g.lmax == self.lmax and g.nmax == self.nmax and g.numr == self.numr and
                    g.cmap == self.cmap and abs(g.rmin - self.rmin) < 1e-12 * max(1.0, self.rmin) and
                    abs(g.rmax - self.rmax) < 1e-12 * self.rmax and abs(g.rmap - self.rmap) < 1e-14)
            if same:
                return g
        g = build_slgrid(self.model, self.lmax, self.nmax, numr=self.numr, rmin=self.rmin,
                         rmax=self.rmax, cmap=self.cmap, rmap=self.rmap)
        g.save(path)
        return g

    # -- index helpers of the pybind layer (pyEXP/BasisWrappers.cc:2065-2110) ---------------------------
    def getLmax(self) -> int:
        return self.lmax

    def getNmax(self) -> int:
        return self.nmax

    def I(self, l: int, m: int, n: int = 0) -> int:
        """Packing index of the coefficient (l, m, n): (lmax+1)(lmax+2)/2 * n + l(l+1)/2 + m."""
        if l < 0:
            raise RuntimeError("l must be greater than 0")
        if m < 0:
            raise RuntimeError("m must be greater than 0")
        if n < 0:
            raise RuntimeError("n must be greater than 0")
        if abs(m) > l:
            raise RuntimeError("m must be less than or equal to l")
        return (self.lmax + 1) * (self.lmax + 2) // 2 * n + l * (l + 1) // 2 + m

    def invI(self, I: int):
        """(l, m, n) of a packing index (the inverse of ``I``)."""
        ltot = (self.lmax + 1) * (self.lmax + 2) // 2
        n = I // ltot
        L = I - n * ltot
        l = int(math.floor(0.5 * (-1.0 + math.sqrt(1.0 + 8.0 * L))))
        return l, L - l * (l + 1) // 2, n

    def cacheInfo(self, cachefile: Optional[str] = None) -> dict:
        """``SLGridSph::cacheInfo`` (the header of the cache file)."""
        from . import h5cache
        path = cachefile or self.cachename
        if h5cache.available() and os.path.exists(path):
            h = h5cache.read_slgrid_header(path)
            return dict(geometry=h["geometry"], forceID=h["forceID"], model=h["model"],
                        lmax=h["lmax"], nmax=h["nmax"], numr=h["numr"], cmap=h["cmap"],
                        rmin=h["rmin"], rmax=h["rmax"], rmapping=h["rmapping"],
                        diverge=h["diverge"], dfac=h["dfac"])
        if not path.endswith(".npz"):
            path += ".npz"
        g = SLGridSph.load(path)
        return dict(geometry="sphere", lmax=g.lmax, nmax=g.nmax, numr=g.numr, cmap=g.cmap,
                    rmin=g.rmin, rmax=g.rmax, rmapping=g.rmap)

    def orthoCheck(self, num: int = 200):
        """exputil/SLGridMP2.cc:1775-1824 on the host tables"""
        from .slgrid import orthocheck
        return orthocheck(self.grid, num)

    def orthoTest(self, num: int = 200) -> None:
        """exputil/orthoTest.cc:19-87, tolerance orthoTol = 1e-2 (exputil/libvars.cc:40)"""
        worst = max(np.abs(m - np.eye(self.nmax)).max() for m in self.orthoCheck(num))
        if worst > 1e-2:
            raise RuntimeError(f"SphericalSL: orthogonality check failed, worst={worst:.3e}")

    # -- coefficients ---------------------------------------------------------------------------------
    def reset_coefs(self) -> None:
        self.expcoef[:] = 0.0
        self.used = 0
        self.totalMass = 0.0                                 # (:475)
        if getattr(self, "pcavar", False):                   # zero_covariance (:478)
            self.force.cov_reset()

    def _dsmall(self, v: float) -> None:
        from ._lib import check
        check(self.force.lib.exp_amd_sph_set_dsmall(self.force.h, float(v)), self.ctx.h)

    def _accumulate_batch(self, m, pos, seq=None, frame=None) -> None:
        if len(m) == 0:
            return
        c = Component.from_arrays(self.ctx, m, pos) if frame is None else Component.from_frame(self.ctx, m, pos, None, *frame)
        self._dsmall(1.0e-20)                                # Spherical::accumulate (:588)
        self.force.determine_coefficients(c)
        self.expcoef += self.force.get_coefs()
        if getattr(self, "pcavar", False):                   # the pcavar block of accumulate (:613-660)
            self.force.cov_accumulate(c, self.used)
        self.used += self.force.Used()
        self.totalMass += self.force.window_mass(c)          # totalMass += mass (:607)
        c.close()

    # -- the basis functions on a grid (expui/BiorthBasis.cc:960-993; pyEXP/BasisWrappers.cc:2142) -------
    def getBasis(self, logxmin: float = -3.0, logxmax: float = 0.5, numgrid: int = 2000):
        """``SphericalSL::getBasis(logxmin, logxmax, numgrid)`` -> ``ret[l][n]`` = dict of ``potential``,
        ``density``, ``rforce`` arrays on the grid r_i = 10^(logxmin + i (logxmax - logxmin)/(numgrid-1))."""
        dx = (logxmax - logxmin) / (numgrid - 1)
        r = np.array([math.pow(10.0, logxmin + dx * i) for i in range(numgrid)])
        t = self.force.basis(r)                              # [3, L+1, nmax, numgrid]
        return [[{"potential": t[0, l, n].copy(), "density": t[1, l, n].copy(), "rforce": t[2, l, n].copy()}
                 for n in range(self.nmax)] for l in range(self.lmax + 1)]

    # -- coefficients of a function by quadrature (expui/BiorthBasis.cc:5230-5458) -------------------------
    @staticmethod
    def _call_on_points(func, x, y, z, *extra):
        """func(x, y, z[, time]) at arrays of points: one vectorised call when the callable takes arrays
        (and returns one value per point), else point by point as pybind11 would call it."""
        try:
            v = np.asarray(func(x, y, z, *extra), dtype=np.float64)
            if v.shape == x.shape:
                return v
        except Exception:
            pass
        return np.array([float(func(float(a), float(b), float(c), *extra)) for a, b, c in zip(x, y, z)])

    def _quadrature_points(self, params, who):
        """The knots^3 product rule of makeFromFunction / computeQuadrature: Gauss-Legendre in the mapped
        radius xi in [xi(rmin), xi(rmax)] and in cos(theta), uniform in phi -> (x, y, z, weight)."""
        rmapping = float(params.get("rmapping", self.rmap))
        knots = int(params.get("knots", 200))
        r_to_x = lambda r: (r / rmapping - 1.0) / (r / rmapping + 1.0)
        ximin, ximax = r_to_x(self.rmin), r_to_x(self.rmax)
        for v, msg in ((ximin, "x<=-1"), (ximax, "x<=-1")):
            if v <= -1.0:
                raise RuntimeError(f"BiorthBasis::{who}: {msg}")
        for v in (ximin, ximax):
            if v >= 1.0:
                raise RuntimeError(f"BiorthBasis::{who}: x>=+1")
        xk, wk = _legequad(knots)
        xx = ximin + (ximax - ximin) * xk
        rr = (1.0 + xx) / (1.0 - xx) * rmapping
        dxr = 0.5 * (1.0 - xx) * (1.0 - xx) / rmapping
        costh = -1.0 + 2.0 * xk
        sinth = np.sqrt(np.abs(1.0 - costh * costh))
        phi = 2.0 * math.pi / knots * np.arange(knots)
        R, T, P = np.meshgrid(np.arange(knots), np.arange(knots), np.arange(knots), indexing="ij")
        R, T, P = R.ravel(), T.ravel(), P.ravel()
        x = rr[R] * sinth[T] * np.cos(phi[P])
        y = rr[R] * sinth[T] * np.sin(phi[P])
        z = rr[R] * costh[T]
        w = (ximax - ximin) * rr[R] * rr[R] / dxr[R] * 2.0 * wk[R] * wk[T] * 2.0 * math.pi / knots
        return x, y, z, w

    def makeFromFunction(self, func, params=None, time: float = 0.0, potential: bool = False) -> SphStruct:
        """``Spherical::makeFromFunction(func, params, time, potential)`` (expui/BiorthBasis.cc:5230-5362;
        pyEXP/BasisWrappers.cc:1464): the expansion of a density (``potential`` False: against the potential
        functions) or of a potential (True: against the density functions) given as a callable
        ``func(x, y, z, time)``, by the knots^3 product rule (``params``: knots, default 200; rmapping).
        The quadrature points go through the SAME accumulation kernels as particles do, each carrying its
        weight as a mass: mat(l, n) = sum_p w_p f(p) factorial(l,m) P_lm e^{i m phi} phi_ln(r_p) -- no -4 pi,
        tables at r itself.  Nothing of the basis' own coefficient state is touched."""
        params = dict(params or {})
        x, y, z, w = self._quadrature_points(params, "makeFromFunction")
        fval = self._call_on_points(func, x, y, z, time) * w
        force = self._quadrature_force(potential)
        c = Component.from_arrays(self.ctx, fval / (-4.0 * math.pi), np.stack([x, y, z], 1))
        force.determine_coefficients(c)
        rows = force.get_coefs()
        c.close()
        from .coefs import real_rows_to_complex
        return SphStruct(self.lmax, self.nmax, self.scale, time, real_rows_to_complex(rows, self.lmax),
                         np.zeros(3), np.eye(3))

    def _quadrature_force(self, potential: bool) -> SphereSL:
        """A force object over the same tables with scale 1, no flags and no r offset (get_pot(r) as it is);
        for ``potential`` the density functions ef sqrt(ev) d0 in the place of ef / sqrt(ev) p0
        (SLGridSph::get_dens, exputil/SLGridMP2.cc:913-950)."""
        key = "_qforce_dens" if potential else "_qforce_pot"
        f = getattr(self, key, None)
        if f is None:
            g = self.grid
            if potential:
                import copy
                g = copy.copy(self.grid)
                g.ef = self.grid.ef * self.grid.ev[:, :, None]       # (ef ev) / sqrt(ev) = ef sqrt(ev)
                g.p0 = np.asarray(self.grid.d0, dtype=np.float64).copy()
            f = SphereSL(self.ctx, g, scale=1.0, rmin=self.rmin, rmax=self.rmax)
            from ._lib import check
            check(f.lib.exp_amd_sph_set_dsmall(f.h, 0.0), self.ctx.h)
            setattr(self, key, f)
        return f

    def computeQuadrature(self, func, params=None) -> float:
        """``Spherical::computeQuadrature(func, params)`` (expui/BiorthBasis.cc:5364-5457;
        pyEXP/BasisWrappers.cc:1496): the integral of ``func(x, y, z)`` over rmin <= r <= rmax by the
        same product rule -- the check of a makeFromFunction input (no basis function is involved)."""
        x, y, z, w = self._quadrature_points(dict(params or {}), "computeQuadrature")
        return float(np.sum(self._call_on_points(func, x, y, z) * w))

    # -- covariance by sub-sampling (expui/BiorthBasis.H:425-470) -------------------------------------
    def enableCoefCovariance(self, pcavar: bool, sampT: int = 100, ftype: bool = False,
                             covr_tot: bool = True, covar: bool = True) -> None:
        """``enableCoefCovariance(pcavar, sampT, ftype, covr_tot, covar)`` (expui/BiorthBasis.H:467-482):
        (re)initialise the sub-sample means and covariances (init_covariance, expui/BiorthBasis.cc:342-365)
        and the HDF5 store's flags: summed covariance only (covr_tot), full matrices or diagonals (covar);
        32-bit storage (ftype) is not written by this build."""
        self.pcavar, self.sampT = bool(pcavar), max(1, int(sampT))
        self._covar_flags = (bool(covr_tot), bool(covar))
        if ftype:
            raise RuntimeError("enableCoefCovariance: 32-bit covariance storage (ftype) is not supported")
        self.force.cov_enable(self.sampT if self.pcavar else 0)

    def writeCoefCovariance(self, compname: str, runtag: str, time: float = 0.0) -> str:
        """``Spherical::writeCoefCovariance`` (expui/BiorthBasis.H:433-463) -> ``SubsampleCovariance::
        writeCoefCovariance`` (expui/Covariance.cc:283-417): create or extend ``coefcovar.<compname>.
        <runtag>.h5`` with the current sub-sample counts, masses, means and covariances."""
        if not getattr(self, "pcavar", False):
            raise RuntimeError("Spherical::writeCoefCovariance: covariance storage not initialized")
        from . import h5cache
        d = self.force.cov_get()
        fname = f"coefcovar.{compname}.{runtag}.h5"
        summed, covar = getattr(self, "_covar_flags", (True, True))
        h5cache.covar_append(fname, "SphereSL", 0, (self.lmax, self.nmax), (self.scale, self.rmin, self.rmax),
                             time, d["counts"], d["masses"], d["mean"], d["covr"].astype(np.complex128),
                             summed=summed, covar=covar)
        return fname

    def getCovarSamples(self):
        """(sampleCounts, sampleMasses), one entry per sub-sample."""
        d = self.force.cov_get()
        return d["counts"], d["masses"]

    def getCoefCovariance(self):
        """[T][lm] -> (mean vector [nmax] complex, covariance matrix [nmax, nmax] complex), lm the
        (l, m >= 0) index of the coefficient packing (``Spherical::getCoefCovariance``)."""
        d = self.force.cov_get()
        return [[(d["mean"][t, lm].copy(), d["covr"][t, lm].astype(np.complex128))
                 for lm in range(d["mean"].shape[1])] for t in range(self.sampT)]

    def load_coefs(self, time: float = 0.0) -> SphStruct:
        """real rows -> complex (l, m>=0) packing (expui/BiorthBasis.cc:482-517)"""
        ldim = (self.lmax + 1) * (self.lmax + 2) // 2
        cf = np.zeros((ldim, self.nmax), dtype=np.complex128)
        L0 = L1 = 0
        for l in range(self.lmax + 1):
            for mm in range(l + 1):
                if mm == 0:
                    cf[L0] = self.expcoef[L1]
                    L1 += 1
                else:
                    cf[L0] = self.expcoef[L1] + 1j * self.expcoef[L1 + 1]
                    L1 += 2
                L0 += 1
        self.coefret = SphStruct(self.lmax, self.nmax, self.scale, time, cf, self.coefctr.copy(),
                                 self.coefrot.copy())
        self.force.set_coefs(self._windowed(self.expcoef))
        return self.coefret

    def _windowed(self, expcoef: np.ndarray) -> np.ndarray:
        """The coefficient set the EVALUATIONS see: rows of l >= 1 restricted to n in
        [max(0, N1), min(nmax-1, N2)]; the monopole row keeps every n, as in the reference."""
        lo, hi = max(0, self.N1), min(self.nmax - 1, self.N2)
        if lo == 0 and hi == self.nmax - 1:
            return expcoef
        w = expcoef.copy()
        w[1:, :lo] = 0.0
        w[1:, hi + 1:] = 0.0
        return w

    def set_coefs(self, coef: SphStruct) -> None:
        """expui/BiorthBasis.cc:519-581"""
        ldim = (self.lmax + 1) * (self.lmax + 2) // 2
        if coef.coefs.shape != (ldim, self.nmax):
            raise RuntimeError(f"Spherical::set_coefs: the basis has (lmax, nmax)=({self.lmax}, "
                               f"{self.nmax}) and the dimensions must be (rows, cols)=({ldim}, "
                               f"{self.nmax}). The coef structure has (rows, cols)={coef.coefs.shape}")
        L0 = L1 = 0
        for l in range(self.lmax + 1):
            for mm in range(l + 1):
                self.expcoef[L1] = coef.coefs[L0].real
                if mm == 0:
                    L1 += 1
                else:
                    self.expcoef[L1 + 1] = coef.coefs[L0].imag
                    L1 += 2
                L0 += 1
        self.coefret = coef
        self.coefctr = np.asarray(coef.ctr, dtype=np.float64) if np.size(coef.ctr) else np.zeros(3)
        self.force.set_coefs(self._windowed(self.expcoef))

    def _accel(self, pos: np.ndarray) -> np.ndarray:
        """``Spherical::computeAccel`` (expui/BiorthBasis.cc:818-926): the tables at r/scale whatever r
        is (no exterior continuation: exp_amd_sph_set_exterior(0)).  One deliberate difference: ON the
        polar axis the reference divides the azimuthal term by x^2 + y^2 = 0 and returns NaN in x and
        y; the device guards that term (x^2 + y^2 > 1e-16, as the n-body force does) and returns the
        finite limit."""
        c = Component.from_frame(self.ctx, np.ones(len(pos)), pos)   # (no frame: the [n, 3] array goes up as it lies)
        self._dsmall(1.0e-18)                                # Spherical::computeAccel (:824-825)
        self.force.get_acceleration_and_potential(c, external=True)
        acc = c.download(("acc",))["acc"]
        c.close()
        return acc


class Cylindrical(BiorthBasis):
    """``cylinder`` (expui/BiorthBasis.cc Cylindrical): YAML keys acyl, hcyl, mmax, nmax, ncylnx,
    ncylny, rcylmin, rcylmax, lmaxfid, nmaxfid, ncylr, rnum, tnum, cmapr, cmapz, cachename."""

    coordinates = "cylindrical"          # expui/BiorthBasis.cc:1744-1746

    name = "cylinder"

    def __init__(self, conf: dict, ctx: Optional[Context] = None):
        super().__init__(conf, ctx)
        bad = set(conf) - CYL_KEYS
        if bad:
            raise RuntimeError(f"Basis::Basis::Cylindrical: unmatched parameter(s) {sorted(bad)}")
        g = conf.get
        self.acyl, self.hcyl = float(g("acyl", 0.01)), float(g("hcyl", 0.002))
        self.mmax, self.nmax = int(g("mmax", 6)), int(g("nmax", 18))
        self.ncylnx, self.ncylny = int(g("ncylnx", 256)), int(g("ncylny", 128))
        self.rcylmin, self.rcylmax = float(g("rcylmin", 0.001)), float(g("rcylmax", 20.0))
        self.lmaxfid, self.nmaxfid = int(g("lmaxfid", 72)), int(g("nmaxfid", 64))
        self.ncylr = int(g("ncylr", 2000))
        self.rnum, self.tnum = int(g("rnum", 200)), int(g("tnum", 80))
        self.cmapr, self.cmapz = int(g("cmapr", g("cmap", 1))), int(g("cmapz", 1))
        # the number of vertically antisymmetric functions per m (expui/BiorthBasis.cc:1389, :1470; EmpCylSL's constructor
        # keeps the even / odd split only for 0 <= ncylodd <= nmax, exputil/EmpCylSL.cc:178-185)
        self.ncylodd = int(g("ncylodd", 9))
        # the conditioning of the basis (expui/BiorthBasis.cc:1397-1440, :1475-1495, :1543): azimuthal knots, the shift of the
        # m >= 1 targets along x, the functional form of the target disk and its parameters
        self.pnum = max(1, int(g("pnum", 1)))
        self.ashift = float(g("ashift", 0.0))
        self.sech2 = bool(g("sech2", False))
        self.dtype = str(g("dtype", "exponential")).lower()
        if self.dtype not in ("constant", "gaussian", "mn", "exponential", "doubleexpon", "diskbulge", "python"):
            raise RuntimeError("Cylindrical::initialize: invalid DiskType")
        if self.dtype == "python":
            raise RuntimeError("Cylindrical: dtype 'python' (a user density through pyname) is not supported: pass the "
                               "callable to exp_amd.empcyl.build_empcyl(dens=...) and hand the cache to the basis")
        for key, why in (("deproject", "the deprojected helper basis (EmpCylSL::create_deprojection)"),
                         ("logr", "the logarithmic radial grid of the helper model (EmpCylSL::logarithmic)")):
            if bool(g(key, False)):
                raise RuntimeError(f"Cylindrical: '{key}: true' is not supported here -- {why} would give another basis than "
                                   "the one this build makes; drop the key or supply the reference's cache file")
        self.aratio, self.hratio = float(g("aratio", 1.0)), float(g("hratio", 1.0))
        self.dweight, self.Mfac, self.HERNA = float(g("dweight", 1.0)), float(g("Mfac", 1.0)), float(g("HERNA", 0.10))
        self.rwidth, self.rtrunc = float(g("rwidth", 0.0)), float(g("rtrunc", 0.1))
        self.cachename = g("cachename", "")
        if "eof_file" in conf:                   # the deprecated spelling wins when both are given (:1471-1472, :1517-1526)
            print("Cylinder: parameter 'eof_file' is deprecated. and will be removed in a future release. Please use "
                  "'cachename' instead.")
            self.cachename = str(conf["eof_file"])
        if "density" in conf:
            print("Cylindrical: parameter 'density' is deprecated. The density field will be computed regardless.")
        if not self.cachename:
            raise RuntimeError("Cylindrical requires a specified cachename in your YAML config")
        self.grid = self._load_or_build()
        # mlim (expui/BiorthBasis.cc:1384, :1466, :1620: `if (mlim>=0) sl->set_mlim(mlim)`): harmonics above it take no part
        self.mlim = int(g("mlim", np.iinfo(np.int32).max))
        self.force = Cylinder(self.ctx, self.grid, rcylmax=self.rcylmax,
                              EVEN_M=bool(g("EVEN_M", False)), mlim=self.mlim)
        self.cos = np.zeros((self.mmax + 1, self.nmax))
        self.sin = np.zeros((self.mmax + 1, self.nmax))
        self.cylmass = 0.0
        self.used = 0

    def getFieldLabels(self, ctype: Optional[str] = None):
        """... with "midplane" behind the cylindrical forces while midplane evaluation is on
        (expui/BiorthBasis.cc:79-84; the reference adds the label for every basis but only ``Cylindrical::cyl_eval``
        returns the value)."""
        labels = super().getFieldLabels(ctype)
        if self.midplane and (ctype or self.coordinates).lower() == "cylindrical":
            labels = labels + ["midplane"]
        return labels

    def __call__(self, x1, x2, x3, ctype: str = "spherical"):
        """``Cylindrical::cyl_eval`` (expui/BiorthBasis.cc:1823-1849): with midplane evaluation on, a tenth value --
        the height of the density peak in the column |z| <= colh * hcyl above (R, phi):
        ``EmpCylSL::accumulated_midplane_eval`` (exputil/EmpCylSL.cc:5506-5554; 40 samples, the peak sample refined by
        the parabola through its neighbours, the peak DENSITY when it sits at an end of the column).  The 40 columns
        of all points are one launch of the fields kernel."""
        out = super().__call__(x1, x2, x3, ctype)
        if not (self.midplane and ctype.lower() == "cylindrical"):
            return out
        single = np.ndim(x1) == 0
        R, phi = np.atleast_1d(np.asarray(x1, np.float64)), np.atleast_1d(np.asarray(x3, np.float64))
        num = 40
        zmin, zmax = -self.colh * self.hcyl, self.colh * self.hcyl
        dz = (zmax - zmin) / (num - 1)
        zk = zmin + dz * np.arange(num)
        dens = self.force.fields(np.repeat(R, num), np.tile(zk, R.size), np.repeat(phi, num), "cylindrical")[:, 2]
        dens = dens.reshape(R.size, num)
        kp = np.argmax(dens, axis=1)                      # (first maximum: the reference's scan keeps it on ties)
        rows = np.arange(R.size)
        pval = dens[rows, kp]
        inner = (kp > 0) & (kp < num - 1)
        km, kq = np.clip(kp - 1, 0, num - 1), np.clip(kp + 1, 0, num - 1)
        f0, f1, f2 = dens[rows, km], dens[rows, kp], dens[rows, kq]
        z0 = zmin + dz * kp
        denom = f0 - 2.0 * f1 + f2
        with np.errstate(divide="ignore", invalid="ignore"):
            quad = ((2 * z0 + dz) * f0 * 0.5 - 2 * z0 * f1 + (2 * z0 - dz) * f2 * 0.5) / denom
        height = np.where(inner, np.where(np.abs(denom) < 1.0e-16, z0, quad), pval)
        full = np.concatenate([np.atleast_2d(out), height[:, None]], axis=1)
        return full[0] if single else full

    def _cache_path(self) -> str:
        return self.cachename if self.cachename.endswith(".npz") else self.cachename + ".npz"

    def _build(self) -> EmpCylGrid:
        return build_empcyl(mmax=self.mmax, norder=self.nmax, numx=self.ncylnx, numy=self.ncylny,
                            acyl=self.acyl, hcyl=self.hcyl, rcylmin=self.rcylmin,
                            rcylmax=self.rcylmax, lmaxfid=self.lmaxfid, nmaxfid=self.nmaxfid,
                            numr=self.ncylr, cmapr=self.cmapr, cmapz=self.cmapz, rnum=self.rnum,
                            tnum=self.tnum, nodd=self.ncylodd, pnum=self.pnum, ashift=self.ashift,
                            dens=self.DiskDens)

    def DiskDens(self, R, z, phi=0.0):
        """``Cylindrical::DiskDens`` (expui/BiorthBasis.cc:1265-1341): the target density the basis is conditioned on, by
        ``dtype``; numpy arrays in, arrays out."""
        R, z = np.asarray(R, dtype=np.float64), np.asarray(z, dtype=np.float64)
        a, h0 = self.acyl, self.hcyl
        pi = math.pi

        def sech2_of(zz, h):                       # 1 / cosh^2 without overflow
            f = np.exp(-np.abs(zz) / h)
            sq = 2.0 * f / (1.0 + f * f)
            return sq * sq

        if self.dtype == "constant":
            ans = np.where((R < a) & (np.abs(z) < h0), 1.0 / (2.0 * h0 * pi * a * a), 0.0)
        elif self.dtype == "gaussian":
            ans = np.where(np.abs(z) < h0, 1.0 / (2.0 * h0 * 2.0 * pi * a * a) * np.exp(-R * R / (2.0 * a * a)), 0.0)
        elif self.dtype == "mn":
            Z2 = z * z + h0 * h0
            Z = np.sqrt(Z2)
            Q2 = (a + Z) * (a + Z)
            ans = 0.25 * h0 * h0 / pi * (a * R * R + (a + 3.0 * Z) * Q2) / ((R * R + Q2) ** 2.5 * Z * Z2)
        elif self.dtype == "doubleexpon":
            a1, a2, h1, h2 = a, a * self.aratio, h0, h0 * self.hratio
            w1, w2 = 1.0 / (1.0 + self.dweight), self.dweight / (1.0 + self.dweight)
            if self.sech2:
                h1, h2 = 0.5 * h1, 0.5 * h2
            ans = (w1 * np.exp(-R / a1) * sech2_of(z, h1) / (4.0 * pi * a1 * a1 * h1) +
                   w2 * np.exp(-R / a2) * sech2_of(z, h2) / (4.0 * pi * a2 * a2 * h2))
        elif self.dtype == "diskbulge":
            h = 0.5 * h0 if self.sech2 else h0
            rr = np.sqrt(R * R + z * z)
            w1, w2, as_ = self.Mfac, 1.0 - self.Mfac, self.HERNA
            with np.errstate(divide="ignore", invalid="ignore"):
                bulge = w2 * as_ ** 4 / (2.0 * pi * rr) * (rr + as_) ** -3.0 if w2 != 0.0 else 0.0
            ans = w1 * np.exp(-R / a) * sech2_of(z, h) / (4.0 * pi * a * a * h) + bulge
        else:                                      # exponential
            h = 0.5 * h0 if self.sech2 else h0
            ans = np.exp(-R / a) * sech2_of(z, h) / (4.0 * pi * a * a * h)
        if self.rwidth > 0.0:
            from scipy.special import erf
            ans = ans * erf((self.rtrunc - R) / self.rwidth)
        return ans

    def _even_odd(self):
        """(neven, nodd) as the cache records them (exputil/EmpCylSL.cc:98-99, :182-184): zeros without the split"""
        return (self.nmax - self.ncylodd, self.ncylodd) if 0 <= self.ncylodd <= self.nmax else (0, 0)

    def _load_or_build(self) -> EmpCylGrid:
        """EXP's own HDF5 cache ``cachename`` (exputil/EmpCylSL.cc:7378-7640) when the HDF5 shim is
        available -- read if it matches, else rebuilt and written --, an ``.npz`` twin otherwise."""
        from . import h5cache
        if h5cache.available():
            if os.path.exists(self.cachename):
                try:
                    return h5cache.read_empcyl_cache(
                        self.cachename,
                        check=dict(mmax=self.mmax, nmax=self.nmax, numx=self.ncylnx, numy=self.ncylny,
                                   lmaxfid=self.lmaxfid, nmaxfid=self.nmaxfid, cmapr=self.cmapr,
                                   cmapz=self.cmapz, rmin=float(self.rcylmin),
                                   rmax=float(self.rcylmax), ascl=float(self.acyl),
                                   hscl=float(self.hcyl), neven=self._even_odd()[0], nodd=self._even_odd()[1]))
                except RuntimeError:
                    pass                                  # the reference rebuilds on any mismatch
            gr = self._build()
            h5cache.write_empcyl_cache(self.cachename, gr, self.lmaxfid, self.nmaxfid, neven=self._even_odd()[0],
                                       nodd=self._even_odd()[1])
            return gr
        path = self._cache_path()
        if os.path.exists(path):
            gr = EmpCylGrid.load(path)
            if (gr.mmax == self.mmax and gr.norder == self.nmax and gr.numx == self.ncylnx and
                    gr.numy == self.ncylny and abs(gr.ascale - self.acyl) < 1e-15 and
                    abs(gr.hscale - self.hcyl) < 1e-15 and gr.cmapr == self.cmapr and
                    gr.cmapz == self.cmapz):
                return gr
        gr = self._build()
        gr.save(path)
        return gr

    def cacheInfo(self, cachefile: Optional[str] = None) -> dict:
        from . import h5cache
        path = cachefile or self.cachename
        if h5cache.available() and os.path.exists(path):
            h = h5cache.read_empcyl_header(path)
            return {k: h[k] for k in ("geometry", "forceID", "model", "mmax", "nmax", "numx", "numy",
                                      "lmaxfid", "nmaxfid", "neven", "nodd", "cmapr", "cmapz",
                                      "rmin", "rmax", "ascl", "hscl", "cmass")}
        if not path.endswith(".npz"):
            path += ".npz"
        gr = EmpCylGrid.load(path)
        return dict(geometry="cylinder", mmax=gr.mmax, nmax=gr.norder, numx=gr.numx, numy=gr.numy,
                    cmapr=gr.cmapr, cmapz=gr.cmapz, rmin=gr.rmin, rmax=gr.rmax, ascl=gr.ascale,
                    hscl=gr.hscale)

    def reset_coefs(self) -> None:
        self.cos[:] = 0.0
        self.sin[:] = 0.0
        self.cylmass = 0.0
        self.used = 0
        if getattr(self, "pcavar", False):                   # setup_accumulation zeroes VC / MV too
            self.force.cov_reset()

    def _accumulate_batch(self, m, pos, seq=None, frame=None) -> None:
        if len(m) == 0:
            return
        c = Component.from_arrays(self.ctx, m, pos) if frame is None else Component.from_frame(self.ctx, m, pos, None, *frame)
        self.force.determine_coefficients(c)
        cc, ss = self.force.get_coefs()
        self.cos += cc
        self.sin += ss
        self.cylmass += self.force.cylmass
        self.used += self.force.Used()
        if getattr(self, "pcavar", False):                   # sl->accumulate(..., indx, 0, 0, pcavar)
            self.force.cov_accumulate(c, seq)
        c.close()

    def getMass(self) -> float:
        """``getMass``: the reference's Cylindrical never updates ``totalMass`` (its accumulate hands the
        particle straight to EmpCylSL, expui/BiorthBasis.cc:1851-1857; the member is not even initialised);
        what a caller wants from "the mass on the grid" is EmpCylSL's own count, cylmass -- returned here."""
        return float(self.cylmass)

    def _fields_ready(self):
        self.force.fields(np.zeros(1), np.zeros(1), np.zeros(1), "cylindrical")     # uploads the density tables

    # -- the basis functions on a grid (expui/BiorthBasis.cc:1930-1974; pyEXP/BasisWrappers.cc:1811) --------
    def getBasis(self, xmin: float = 0.0, xmax: float = 1.0, numR: int = 40, zmin: float = -0.1,
                 zmax: float = 0.1, numZ: int = 40, linear: bool = True):
        """``Cylindrical::getBasis(xmin, xmax, numR, zmin, zmax, numZ, linear)`` -> ``ret[m][n]`` = dict of
        ``potential``, ``density``, ``rforce``, ``zforce`` [numR, numZ] arrays (R = 10^x when not linear):
        ``EmpCylSL::get_all`` at phi = 0 with the force's current cylmass beyond the table."""
        delR = (xmax - xmin) / max(numR - 1, 1)
        delZ = (zmax - zmin) / max(numZ - 1, 1)
        R = np.array([xmin + delR * i for i in range(numR)])
        if not linear:
            R = np.array([math.pow(10.0, v) for v in R])
        Z = np.array([zmin + delZ * j for j in range(numZ)])
        RR, ZZ = np.meshgrid(R, Z, indexing="ij")
        self._fields_ready()
        t = self.force.basis(RR.ravel(), ZZ.ravel()).reshape(4, self.mmax + 1, self.nmax, numR, numZ)
        keys = ("potential", "density", "rforce", "zforce")
        return [[{k: t[j, m, n].copy() for j, k in enumerate(keys)} for n in range(self.nmax)]
                for m in range(self.mmax + 1)]

    def orthoCheck(self, knots: int = 40):
        """``Cylindrical::orthoCheck`` (expui/BiorthBasis.H:1105-1109 -> EmpCylSL::orthoCheck,
        exputil/EmpCylSL.cc:7199-7260; pyEXP/BasisWrappers.cc:1854) -> one [nmax, nmax] matrix per m (the
        ``knots`` argument is ignored by the reference too: the integral runs over the table grid)."""
        self._fields_ready()
        return [m for m in self.force.orthocheck()]

    # -- coefficients of a function by quadrature (expui/BiorthBasis.cc:5459-5630) -------------------------
    def _quadrature_points(self, params):
        """knots^3 product rule over the table's own box: Gauss-Legendre in X = xi(R) and Y = y(z), uniform
        in phi (expui/BiorthBasis.cc:5474-5530) -> (x, y, z, weight)."""
        g = self.grid
        knots = int(params.get("knots", 200))
        A, H, Rtab = g.ascale, g.hscale, g.rtable
        r_to_xi = (lambda r: (r / A - 1.0) / (r / A + 1.0)) if g.cmapr > 0 else (lambda r: r)
        if g.cmapz == 1:
            z_to_y = lambda z: math.copysign(math.asinh(abs(z / H)), z)
            y_to_z, d_y_to_z = (lambda y: H * np.sinh(y)), (lambda y: H * np.cosh(y))
        elif g.cmapz == 2:
            z_to_y = lambda z: z / math.sqrt(z * z + H * H)
            y_to_z, d_y_to_z = (lambda y: y * H / np.sqrt(1.0 - y * y)), (lambda y: H * (1.0 - y * y) ** -1.5)
        else:
            z_to_y, y_to_z, d_y_to_z = (lambda z: z), (lambda y: y), (lambda y: np.ones_like(y))
        xmin, xmax = r_to_xi(g.rmin * A), r_to_xi(Rtab * A)
        ymin, ymax = z_to_y(-Rtab * A), z_to_y(Rtab * A)
        xk, wk = _legequad(knots)
        xx, yy = xmin + (xmax - xmin) * xk, ymin + (ymax - ymin) * xk
        if g.cmapr > 0:
            Rk, dxr = (1.0 + xx) / (1.0 - xx) * A, 0.5 * (1.0 - xx) * (1.0 - xx) / A
        else:
            Rk, dxr = xx, np.ones_like(xx)
        zk, dyz = y_to_z(yy), d_y_to_z(yy)
        phi = 2.0 * math.pi / knots * np.arange(knots)
        I, J, K = np.meshgrid(np.arange(knots), np.arange(knots), np.arange(knots), indexing="ij")
        I, J, K = I.ravel(), J.ravel(), K.ravel()
        x, y, z = Rk[I] * np.cos(phi[K]), Rk[I] * np.sin(phi[K]), zk[J]
        w = (xmax - xmin) * (ymax - ymin) * wk[I] * wk[J] * 2.0 * math.pi / knots * Rk[I] / dxr[I] * dyz[J]
        return x, y, z, w

    _call_on_points = staticmethod(SphericalSL._call_on_points)

    def makeFromFunction(self, func, params=None, time: float = 0.0, potential: bool = False) -> CylStruct:
        """``Cylindrical::makeFromFunction`` (expui/BiorthBasis.cc:5459-5556): mat(m, n) = sum_p w_p f(p)
        [potC cos(m phi) + i potS sin(m phi)] (densC / densS with ``potential``), the quadrature points
        taken through the accumulation kernels with their weights as masses.  The box corners lie outside
        the sphere EmpCylSL::accumulate cuts at but inside getPotSC's R <= Rtable: the transient force
        object's cut radius is widened to hold the whole box."""
        x, y, z, w = self._quadrature_points(dict(params or {}))
        fval = self._call_on_points(func, x, y, z, time) * w
        force = self._quadrature_force(potential)
        c = Component.from_arrays(self.ctx, fval / (-4.0 * math.pi), np.stack([x, y, z], 1))
        force.determine_coefficients(c)
        cc, ss = force.get_coefs()
        c.close()
        return CylStruct(self.mmax, self.nmax, time, cc + 1j * ss, np.zeros(3), np.eye(3))

    def _quadrature_force(self, potential: bool) -> Cylinder:
        key = "_qforce_dens" if potential else "_qforce_pot"
        f = getattr(self, key, None)
        if f is None:
            import copy
            g = copy.copy(self.grid)
            g.rtable = self.grid.rtable * 1.5                  # sphere around the (R, z) box of the tables
            if potential:
                if getattr(self.grid, "dens", None) is None:
                    raise RuntimeError("Cylindrical.makeFromFunction: the EmpCylSL grid has no density tables")
                tab = np.array(self.grid.tab, dtype=np.float64, copy=True)
                tab[0], tab[3] = self.grid.dens[0], self.grid.dens[1]       # potC <- densC, potS <- densS
                g.tab = tab
            f = Cylinder(self.ctx, g, rcylmax=1.0e30)
            setattr(self, key, f)
        return f

    def computeQuadrature(self, func, params=None) -> float:
        """``Cylindrical::computeQuadrature`` (expui/BiorthBasis.cc:5558-5630)."""
        x, y, z, w = self._quadrature_points(dict(params or {}))
        return float(np.sum(self._call_on_points(func, x, y, z) * w))

    # -- covariance by sub-sampling (expui/BiorthBasis.H:1120-1145; exputil/EmpCylSL.cc:4974-5015) ---
    def enableCoefCovariance(self, pcavar: bool, sampT: int = 100, ftype: bool = False,
                             covr_tot: bool = True, covar: bool = True) -> None:
        """expui/BiorthBasis.H:1133-1148"""
        self.pcavar, self.sampT = bool(pcavar), max(1, int(sampT))
        self._covar_flags = (bool(covr_tot), bool(covar))
        if ftype:
            raise RuntimeError("enableCoefCovariance: 32-bit covariance storage (ftype) is not supported")
        self.force.cov_enable(self.sampT if self.pcavar else 0)

    def writeCoefCovariance(self, compname: str, runtag: str, time: float = 0.0) -> str:
        """``Cylindrical::writeCoefCovariance`` (expui/BiorthBasis.H:1120-1131); parameters of
        ``Cylindrical::writeCovarH5Params`` (expui/BiorthBasis.cc:5201-5210)."""
        if not getattr(self, "pcavar", False):
            raise RuntimeError("Cylindrical::writeCoefCovariance: covariance storage not initialized")
        from . import h5cache
        d = self.force.cov_get()
        fname = f"coefcovar.{compname}.{runtag}.h5"
        summed, covar = getattr(self, "_covar_flags", (True, True))
        h5cache.covar_append(fname, "Cylindrical", 1, (self.mmax, self.nmax),
                             (self.rcylmin, self.rcylmax, self.acyl, float(self.conf.get("bias", 1.0)), self.hcyl),
                             time, d["counts"], d["masses"], d["mean"], d["covr"], summed=summed, covar=covar)
        return fname

    def getCovarSamples(self):
        d = self.force.cov_get()
        return d["counts"], d["masses"]

    def getCoefCovariance(self):
        """(VC [sampT, mmax+1, nmax], MV [sampT, mmax+1, nmax, nmax]), complex, as
        ``EmpCylSL::getCoefCovariance`` returns them."""
        d = self.force.cov_get()
        return d["mean"], d["covr"]

    def load_coefs(self, time: float = 0.0) -> CylStruct:
        self.coefret = CylStruct(self.mmax, self.nmax, time, self.cos + 1j * self.sin,
                                 self.coefctr.copy(), self.coefrot.copy())
        self.force.set_coefs(self.cos, self.sin)
        self.force.cylmass = self.cylmass
        return self.coefret

    def set_coefs(self, coef: CylStruct) -> None:
        if coef.coefs.shape != (self.mmax + 1, self.nmax):
            raise RuntimeError("Cylindrical::set_coefs: dimension mismatch")
        self.cos, self.sin = coef.coefs.real.copy(), coef.coefs.imag.copy()
        self.coefret = coef
        self.coefctr = np.asarray(coef.ctr, dtype=np.float64) if np.size(coef.ctr) else np.zeros(3)
        self.force.set_coefs(self.cos, self.sin)
        self.force.cylmass = self.cylmass

    def _accel(self, pos: np.ndarray) -> np.ndarray:
        """``Cylindrical::computeAccel`` (expui/BiorthBasis.cc:1804-1821): ``accumulated_eval``
        projected on x, y, z -- no taper, no monopole continuation beyond the table (those belong to
        the n-body ``Cylinder``, src/Cylinder.cc:1364-1414): the force columns of crt_eval."""
        if len(pos) == 0:
            return np.zeros((0, 3))
        return self.force.fields(pos[:, 0], pos[:, 1], pos[:, 2], "cartesian")[:, 6:9].copy()


class Basis:
    """``pyEXP.basis.Basis``: factory by the YAML ``id`` (expui/BasisFactory.cc:156-205)."""

    @staticmethod
    def factory(config, ctx: Optional[Context] = None) -> BiorthBasis:
        node = yaml.safe_load(config) if isinstance(config, str) else config
        if not isinstance(node, dict) or "id" not in node:
            raise RuntimeError("Basis::factory: the configuration needs an 'id'")
        params = node.get("parameters") or {}
        name = node["id"]
        if name == "sphereSL":
            return SphericalSL(params, ctx)
        if name == "cylinder":
            return Cylindrical(params, ctx)
        raise RuntimeError(f"Basis::factory: basis <{name}> is outside the scope of exp_amd "
                           "(sphereSL and cylinder are built)")


def CovarianceReader(filename: str, stride: int = 1):
    """``pyEXP.basis.CovarianceReader(filename, stride=1)`` (pyEXP/BasisWrappers.cc:3173-3215): the reader of
    the ``coefcovar.<name>.<runtag>.h5`` files ``writeCoefCovariance`` produces; ``Times()``,
    ``getCoefCovariance(time)`` -> (counts, masses, means, covariances)."""
    from .h5cache import SubsampleCovariance
    return SubsampleCovariance(filename, stride)


# ---- orbit integration in a time-dependent expansion (expui/BiorthBasis.H:1511-1599, expui/BiorthBasis.cc:
#      4759-5195; pyEXP/BasisWrappers.cc:3050-3170) -------------------------------------------------------------

class AccelFunc:
    """``BasisClasses::AccelFunc``: the acceleration of a (basis, coefficient container) pair at a time.
    A derived class provides ``evalcoefs(t, mod)``, which installs the coefficients for time ``t`` in the
    basis; ``F(t, ps, accel, mod)`` -- the callable IntegrateOrbits is handed -- then adds the field of that
    model at the phase-space points ``ps`` [n, 6] to ``accel`` [n, 3]."""

    def evalcoefs(self, t: float, mod) -> None:          # pragma: no cover - interface
        raise NotImplementedError("AccelFunc::evalcoefs is pure virtual")

    def evalaccel(self, ps: np.ndarray, accel: np.ndarray, mod) -> np.ndarray:
        """``AccelFunc::evalaccel`` (expui/BiorthBasis.cc:4759-4816): points into the expansion frame
        (minus the centre -- zero in a non-inertial frame --, then the rotation), ``getFields`` there, the
        three force columns added to ``accel`` minus the frame's pseudo-acceleration."""
        basis = mod[0]
        ctr = np.zeros(3) if basis.usingNonInertial() else np.asarray(basis.getCenter(), dtype=np.float64)
        rot = np.asarray(basis.getRotation(), dtype=np.float64)
        pp = (np.asarray(ps, dtype=np.float64)[:, :3] - ctr) @ rot.T
        v = np.atleast_2d(basis.getFields(pp[:, 0].copy(), pp[:, 1].copy(), pp[:, 2].copy()))
        accel += v[:, 6:9] - basis.pseudo
        return accel

    def F(self, t: float, ps: np.ndarray, accel: np.ndarray, mod) -> np.ndarray:
        self.evalcoefs(t, mod)
        return self.evalaccel(ps, accel, mod)

    __call__ = F


def _bracket(times, t, who):
    """the (it1, it2, a, b) of AllTimeAccel::evalcoefs / SingleTimeAccel (expui/BiorthBasis.cc:4833-4851)"""
    if t < times[0] or t > times[-1]:
        raise RuntimeError(f"Basis::OneAccel: time t={t} is out of bounds: [{times[0]}, {times[-1]}]")
    i2 = int(np.searchsorted(times, t, side="left"))
    i1 = i2
    if i2 == len(times):
        raise RuntimeError(f"Basis::{who}::evalcoefs: time t={t} out of bounds")
    if i2 == 0:
        i2 += 1
    else:
        i1 -= 1
    a = (times[i2] - t) / (times[i2] - times[i1])
    b = (t - times[i1]) / (times[i2] - times[i1])
    return times[i1], times[i2], a, b


class AllTimeAccel(AccelFunc):
    """``AllTimeAccel``: coefficients linearly interpolated in time between the two stored sets around ``t``;
    centre interpolated likewise, rotation interpolated and projected back onto the rotations by its
    polar decomposition (U V^T of the SVD); the pseudo-acceleration of a non-inertial frame updated."""

    def evalcoefs(self, t: float, mod) -> None:
        import copy
        basis, coefs = mod[0], mod[1]
        t1, t2, a, b = _bracket(coefs.Times(), t, "AllTimeAccel")
        A, B = coefs.getCoefStruct(t1), coefs.getCoefStruct(t2)
        new = A.deepcopy()
        new.time = t
        new.coefs = a * np.asarray(A.coefs) + b * np.asarray(B.coefs)
        new.ctr = a * np.asarray(A.ctr, dtype=np.float64) + b * np.asarray(B.ctr, dtype=np.float64)
        U, _, Vt = np.linalg.svd(a * np.asarray(A.rot, dtype=np.float64) + b * np.asarray(B.rot, dtype=np.float64))
        new.rot = U @ Vt
        basis.set_coefs(new)
        basis.coefrot = new.rot
        basis.setNonInertialAccel(t)


class SingleTimeAccel(AccelFunc):
    """``SingleTimeAccel(t, mod)``: every model's coefficients interpolated to the ONE time ``t`` at
    construction; ``evalcoefs`` then leaves them alone (a frozen potential)."""

    def __init__(self, t: float, mod):
        import copy
        for basis, coefs in [(m[0], m[1]) for m in mod]:
            t1, t2, a, b = _bracket(coefs.Times(), t, "SingleTimeAccel")
            A, B = coefs.getCoefStruct(t1), coefs.getCoefStruct(t2)
            new = A.deepcopy()
            new.time = t
            new.coefs = a * np.asarray(A.coefs) + b * np.asarray(B.coefs)
            if np.size(A.ctr) and np.size(B.ctr):
                new.ctr = a * np.asarray(A.ctr, dtype=np.float64) + b * np.asarray(B.ctr, dtype=np.float64)
            basis.set_coefs(new)

    def evalcoefs(self, t: float, mod) -> None:
        pass


def _one_step(t, h, ps, accel, bfe, F):
    """``OneStep`` (expui/BiorthBasis.cc:4936-5055), its leap-frog branch: drift h/2, kick h with the
    field at time t, drift h/2."""
    ps[:, :3] += ps[:, 3:6] * (0.5 * h)
    accel[:] = 0.0
    for mod in bfe:
        F(t, ps, accel, mod)
    ps[:, 3:6] += accel * h
    ps[:, :3] += ps[:, 3:6] * (0.5 * h)
    return t + h, ps


def IntegrateOrbits(tinit: float, tfinal: float, h: float, ps, bfe, F, nout: int = 0):
    """``pyEXP.basis.IntegrateOrbits(tinit, tfinal, h, ps, bfe, F, nout)`` (expui/BiorthBasis.cc:5056-5195):
    leap-frog orbits of the phase-space points ``ps`` [n, 6] in the fields of the models ``bfe`` = list of
    (basis, coefs) -- evaluated on the GPU through ``getFields`` -- from ``tinit`` to ``tfinal`` in steps of
    about ``h`` (re-fitted to land on ``tfinal``), every ``stride``-th state kept so that ``nout`` states
    come back.  Returns (times [nout], states [n, 6, nout] float32)."""
    ps = np.array(ps, dtype=np.float64)
    if ps.ndim != 2 or ps.shape[1] != 6:
        cols = ps.shape[1] if ps.ndim == 2 else ps.shape[-1]
        raise RuntimeError("IntegrateOrbits: phase space array should be n x 6 where n is the number of "
                           f"particles.  You specified {cols} columns")
    rows = ps.shape[0]
    accel = np.zeros((rows, 3))
    if tfinal == tinit:
        raise RuntimeError("BasisClasses::IntegrateOrbits: tinit cannot be equal to tfinal")
    if h < 0.0 and tfinal > tinit:
        raise RuntimeError("BasisClasses::IntegrateOrbits: tfinal must be smaller than tinit when step size "
                           "is negative")
    if h > 0.0 and tfinal < tinit:
        raise RuntimeError("BasisClasses::IntegrateOrbits: tfinal must be larger than tinit when step size "
                           "is positive")
    if (tfinal - tinit) / h > float(np.iinfo(np.int32).max):
        print("BasisClasses::IntegrateOrbits: step size is too small or time interval is too large.")
        return np.zeros(0), np.zeros((0, 0, 0), dtype=np.float32)
    numT = max(2, int(math.ceil((tfinal - tinit) / h + 0.5)))
    stride = 1
    if nout > 0:
        nout = max(2, int(nout))
        stride = int(math.ceil(numT / nout))
        numT = (nout - 1) * stride + 1
    else:
        nout = numT
    h = (tfinal - tinit) / (numT - 1)
    ret = np.zeros((rows, 6, nout), dtype=np.float32)
    times = np.zeros(nout)
    times[0] = tinit
    ret[:, :, 0] = ps
    sgn = (0 < h) - (h < 0)
    tnow, s, cnt = tinit, 0, 1
    while s < numT:                         # `while (s++ < numT)`: numT passes, s = 1 .. numT inside
        s += 1
        if (tfinal - tnow) * sgn < h * sgn:
            h = tfinal - tnow
        tnow, ps = _one_step(tnow, h, ps, accel, bfe, F)
        if cnt < nout and s % stride == 0:
            times[cnt] = tnow
            ret[:, :, cnt] = ps
            cnt += 1
    times[nout - 1] = tnow
    ret[:, :, nout - 1] = ps
    return times, ret
