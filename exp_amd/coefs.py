"""Coefficient sets in time: EXP's native coefficient stream format and the ``Coefs`` container.

What the reference does either side of the hot path with the coefficients it produces:

* ``SphericalBasis::dump_coefs(ostream&)`` (src/SphericalBasis.cc:1829-1904) appends one record per
  dump to ``outcoef.<name>.<runtag>``: magic ``0xc0a57a2``, a YAML header (id, time, scale, nmax,
  lmax, normed) and the doubles ordered ``[n][l][m: cos, (sin)]``; the legacy layout is the 88-byte
  ``SphCoefHeader`` (include/coef.H:18-25) followed by the same doubles, un-normalised.
* ``CoefClasses::SphStruct::read`` (expui/CoefStruct.cc:372-506) parses both into the complex
  ``[(L+1)(L+2)/2, nmax]`` array (m >= 0), ``SphCoefs::readNativeCoefs`` (expui/Coefficients.cc:
  796-838) collects them by (rounded) time, ``Coefs::interpolate`` (:183-226) blends the two
  bracketing sets linearly -- which is what ``determine_coefficients_playback``
  (src/SphericalBasis.cc:612-680) feeds to the force evaluation.

HDF5 coefficient files (``WriteH5Coefs`` / the reading constructor) go through the libhdf5 C shim
(``exp_amd/csrc_host/h5cache.c``); this image has no h5py/HighFive.  Host-side only: nothing here
touches the device.
"""
from __future__ import annotations

import math
import struct
from typing import BinaryIO, Dict, List, Optional, Tuple

import numpy as np

from .basis import CylStruct, SphStruct

builtins_max, builtins_min = max, min      # (Power's arguments are called min / max, as in the reference)

CMAGIC = 0x0C0A57A2                  # src/SphericalBasis.H:368
_LEGACY = struct.Struct("<64sddii")  # SphCoefHeader: id[64], tnow, scale, nmax, Lmax


def round_time(t: float) -> float:
    """expui/BasisFactory.H roundTime / Coefficients.H: 8 decimal places."""
    return math.floor(t * 1.0e8 + 0.5) / 1.0e8


def real_rows_to_complex(expcoef: np.ndarray, lmax: int) -> np.ndarray:
    """(L+1)^2 x nmax real rows (src/SphericalBasis.cc:513-590 order) -> complex (l, m>=0) rows."""
    nmax = expcoef.shape[1]
    out = np.zeros(((lmax + 1) * (lmax + 2) // 2, nmax), dtype=np.complex128)
    L = off = 0
    for l in range(lmax + 1):
        for m in range(l + 1):
            if m == 0:
                out[L] = expcoef[off]
                off += 1
            else:
                out[L] = expcoef[off] + 1j * expcoef[off + 1]
                off += 2
            L += 1
    return out


def complex_to_real_rows(coefs: np.ndarray, lmax: int) -> np.ndarray:
    nmax = coefs.shape[1]
    out = np.zeros(((lmax + 1) ** 2, nmax))
    L = off = 0
    for l in range(lmax + 1):
        for m in range(l + 1):
            out[off] = coefs[L].real
            off += 1
            if m:
                out[off] = coefs[L].imag
                off += 1
            L += 1
    return out


def write_native(out: BinaryIO, c: SphStruct, force_id: str = "sphereSL") -> None:
    """One record in the new-style native format (src/SphericalBasis.cc:1831-1879)."""
    import yaml
    hdr = yaml.safe_dump({"id": force_id, "time": float(c.time), "scale": float(c.scale),
                          "nmax": int(c.nmax), "lmax": int(c.lmax), "normed": True},
                         default_flow_style=False, sort_keys=False).encode()
    out.write(struct.pack("<II", CMAGIC, len(hdr)))
    out.write(hdr)
    # [n][l][m]: cos, then sin for m > 0  == the real-row order, n-major
    rows = complex_to_real_rows(np.asarray(c.coefs), c.lmax)
    out.write(np.ascontiguousarray(rows.T, dtype="<f8").tobytes())


def read_native_record(inp: BinaryIO, exp_type: bool = True) -> Optional[SphStruct]:
    """``SphStruct::read``: returns None at end of stream."""
    import yaml
    head = inp.read(4)
    if len(head) < 4:
        return None
    normed = False
    if struct.unpack("<I", head)[0] == CMAGIC:
        raw = inp.read(4)
        if len(raw) < 4:
            return None
        (hsize,) = struct.unpack("<I", raw)
        node = yaml.safe_load(inp.read(hsize).decode())
        lmax, nmax = int(node["lmax"]), int(node["nmax"])
        time, scale = float(node["time"]), float(node["scale"])
        normed = bool(node.get("normed", False))
    else:
        rest = inp.read(_LEGACY.size - 4)
        if len(rest) < _LEGACY.size - 4:
            return None
        _id, time, scale, nmax, lmax = _LEGACY.unpack(head + rest)
    nrow = (lmax + 1) ** 2
    buf = inp.read(8 * nrow * nmax)
    if len(buf) < 8 * nrow * nmax:
        return None
    rows = np.frombuffer(buf, dtype="<f8").reshape(nmax, nrow).T.copy()
    if exp_type and not normed:
        # True normed coefficients from a legacy dump.  This is a CORRECTED reading of the reference's
        # legacy branch (expui/CoefStruct.cc:481-503), not a statement-for-statement port: there the
        # running index k walks the rows of the COMPLEX (l, m >= 0) x nmax array but is advanced twice
        # for m > 0 (as if cosine and sine were separate rows), so the factors land on the wrong rows
        # for l >= 1 and run past the array; and the header's scale is ignored.  Here the factor of
        # each (l, m) multiplies its own cosine and sine REAL rows and scale is taken from the header.
        # Legacy files therefore do not round-trip identically to pyEXP; new-style ("normed") records,
        # which every current EXP writes, are unaffected.  Pinned by tests/test_coefs_cpu.py::
        # test_legacy_header_is_normalised.
        off = 0
        for l in range(lmax + 1):
            for m in range(l + 1):
                fac = math.sqrt((0.5 * l + 0.25) / math.pi
                                * math.exp(math.lgamma(1.0 + l - m) - math.lgamma(1.0 + l + m)))
                if m:
                    fac *= math.sqrt(2.0)
                rows[off] *= fac
                off += 1
                if m:
                    rows[off] *= fac
                    off += 1
    return SphStruct(lmax, nmax, scale, time, real_rows_to_complex(rows, lmax), np.zeros(3),
                     np.eye(3))


class Coefs:
    """``CoefClasses::Coefs`` (expui/Coefficients.H:31-290): what the containers of every geometry share
    -- the mnemonic name, the time-keyed map, ``factory`` / ``makecoefs`` / ``addcoef``, deep copies,
    comparison, per-harmonic power, and the HDF5 write / extend pair.  ``SphCoefs`` and ``CylCoefs`` add
    their geometry's shapes."""

    geometry = ""
    _what = "Coefs"

    def __init__(self, name="", verbose: bool = False):
        # pyEXP constructs ``SphCoefs(verbose: bool = False)`` (pyEXP/CoefWrappers.cc:1546); the mnemonic
        # name is set with setName.  A string here is the name, a bool the verbosity.
        if isinstance(name, bool):
            name, verbose = "", name
        self.name = name
        self.verbose = bool(verbose)
        self.coefs: Dict[float, object] = {}
        self.deltaT = 0.01                       # expui/Coefficients.H:133
        self.units: List[Tuple[str, str, float]] = [("G", "none", 1.0)]      # expui/Coefficients.H:121

    # -- unit metadata (expui/Coefficients.cc:74-149; the "Units" dataset of the HDF5 files) -------
    def setUnits(self, name, unit: Optional[str] = None, value: Optional[float] = None) -> None:
        """``setUnits(name, unit, value)`` or ``setUnits([(name, unit, value), ...])``: type and unit go through the
        validator and are stored under their canonical spellings; a type that is already there is updated."""
        from .units import UnitValidator
        if unit is None and value is None and not isinstance(name, str):
            for n, u, v in name:
                self.setUnits(n, u, v)
            return
        ok, cname, cunit = UnitValidator()(str(name), str(unit))
        if not ok:
            raise RuntimeError(f"Coefs::setUnits: Warning, type '{name}' with unit '{unit}' is incompatible or not recognized.")
        val = float(np.float32(value))                     # (Unit::value is a float)
        for k, (n, _, _) in enumerate(self.units):
            if n == cname:
                self.units[k] = (cname, cunit[:15], val)
                return
        self.units.append((cname[:15], cunit[:15], val))

    def removeUnits(self, name: str) -> None:
        self.units = [u for u in self.units if u[0] != name]

    def getUnits(self) -> List[Tuple[str, str, float]]:
        return list(self.units)

    def getGravConstant(self) -> float:
        for n, _, v in self.units:
            if n == "G":
                return v
        return 1.0

    def getAllowedUnitTypes(self) -> List[str]:
        from .units import UnitValidator
        return UnitValidator.getAllowedTypes()

    def getAllowedTypeAliases(self, type: str) -> List[str]:
        from .units import UnitValidator
        return UnitValidator.getAllowedTypeAliases(type)

    def getAllowedUnitNames(self, type: str) -> List[str]:
        from .units import UnitValidator
        return UnitValidator.getAllowedUnits(type)

    def _units_for_h5(self):
        """``Coefs::WriteH5Units`` (expui/Coefficients.cc:152-169): spherical and cylindrical sets must carry four
        units -- (length, mass, time, G) or (length, mass, velocity, G) -- or the write is refused; the records are
        {char name[16], char unit[16], float value}."""
        if len(self.units) != 4:
            raise RuntimeError("---- Coefs::WriteH5Units: Warning, expected 4 units: (length, mass, time, G) or (length, mass, "
                               f"velocity, G), etc. I found {len(self.units)} units instead.  Please  provide a consistent unit set.")
        rec = np.zeros(len(self.units), dtype=[("name", "S16"), ("unit", "S16"), ("value", "<f4")])
        for k, (n, u, v) in enumerate(self.units):
            rec[k] = (n.encode()[:15], u.encode()[:15], v)
        return rec

    def _read_units_h5(self, path: str) -> None:
        """``Coefs::ReadH5Units`` (expui/Coefficients.cc:171-182): taken over when the file has the dataset."""
        import ctypes
        from . import h5cache
        lib = h5cache._load()
        lib.exp_h5_coef_read_units.restype = ctypes.c_int
        rec = np.zeros(16, dtype=[("name", "S16"), ("unit", "S16"), ("value", "<f4")])
        n = ctypes.c_int(0)
        if lib.exp_h5_coef_read_units(path.encode(), 16, rec.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)) == 0 and n.value >= 0:
            self.units = [(r["name"].decode(), r["unit"].decode(), float(r["value"])) for r in rec[:n.value]]

    def setDeltaT(self, dT: float) -> None:
        self.deltaT = float(dT)

    # expui/Coefficients.H:222-228
    def getGeometry(self) -> str:
        return self.geometry

    def getName(self) -> str:
        return self.name

    def setName(self, newname: str) -> None:
        self.name = str(newname)

    def clear(self) -> None:
        self.coefs.clear()

    def zerodata(self) -> None:
        """``Coefs::zerodata``: every stored set keeps its shape and time, its values become zero."""
        for c in self.coefs.values():
            c.coefs = np.zeros_like(np.asarray(c.coefs))

    def deepcopy(self):
        """``SphCoefs::deepcopy`` / ``CylCoefs::deepcopy`` (expui/Coefficients.cc:337-377): new container,
        new structures, same fields."""
        import copy
        ret = type(self)(self.name)
        ret.deltaT = self.deltaT
        ret.units = list(self.units)
        for t, c in self.coefs.items():
            ret.coefs[t] = c.deepcopy() if hasattr(c, "deepcopy") else copy.deepcopy(c)
            ret.coefs[t]._owner = ret
        return ret

    # -- values at one time (``__call__`` is the pybind operator, pyEXP/CoefWrappers.cc:1559, :1634) --
    def getMatrix(self, time: float) -> np.ndarray:
        """``getMatrix(time)``: the complex [rows, nmax] array, EMPTY (0 x 0) when the time is not there
        (expui/Coefficients.cc:683-696)."""
        c = self.coefs.get(round_time(time))
        return np.zeros((0, 0), dtype=np.complex128) if c is None else np.asarray(c.coefs, dtype=np.complex128)

    __call__ = getMatrix

    def getData(self, time: float) -> np.ndarray:
        """``getData(time)``: the same values flattened in the reference's (column-major) storage order."""
        return self.getMatrix(time).reshape(-1, order="F")

    def setMatrix(self, time: float, mat) -> None:
        """``setMatrix(time, mat)``: replace the values of an EXISTING time (expui/Coefficients.cc:713-725)."""
        c = self.coefs.get(round_time(time))
        if c is None:
            raise RuntimeError(f"{self._what}::setMatrix: requested time={time} not found")
        mat = np.asarray(mat, dtype=np.complex128)
        if mat.shape != np.shape(c.coefs):
            raise RuntimeError(f"{self._what}::setMatrix: shape {mat.shape} does not match {np.shape(c.coefs)}")
        c.coefs = mat.copy()

    def setData(self, time: float, dat) -> None:
        c = self.coefs.get(round_time(time))
        if c is None:
            raise RuntimeError(f"{self._what}::setData: requested time={time} not found")
        c.coefs = np.asarray(dat, dtype=np.complex128).reshape(np.shape(c.coefs), order="F").copy()

    def CompareStanzas(self, check: "Coefs") -> bool:
        """``CompareStanzas`` (expui/Coefficients.cc:979-1031, :3056-3100): same times, same orders, same
        values -- exact comparison, as there."""
        if type(check) is not type(self):
            return False
        ret = True
        for t in self.coefs:
            if t not in check.coefs:
                print(f"Can't find Time={t}")
                ret = False
        if not ret:
            print("Times in other coeffcients are:", *check.Times())
            return False
        for t, c in self.coefs.items():
            o = check.coefs[t]
            if (c.nmax != o.nmax or c.time != o.time or getattr(c, "lmax", None) != getattr(o, "lmax", None)
                    or getattr(c, "mmax", None) != getattr(o, "mmax", None)):
                return False
        for t, c in self.coefs.items():
            if not np.array_equal(np.asarray(c.coefs), np.asarray(check.coefs[t].coefs)):
                return False
        return True

    # -- construction from files and single structures ------------------------------------------------
    @staticmethod
    def factory(file: str, stride: int = 1, tmin: float = -math.inf, tmax: float = math.inf) -> "Coefs":
        """``Coefs::factory`` (expui/Coefficients.cc:2911-3018): HDF5 first -- the ``geometry`` attribute
        picks the class --, then EXP's native streams by their magic number (0xc0a57a2 sphere, 0xc0a57a3
        cylinder; anything else is a legacy spherical stream or an ascii table, which this mirror does not
        carry)."""
        import os
        from . import h5cache
        geo = None
        try:
            geo = h5cache.coef_geometry(file)
        except Exception:
            geo = None
        if geo is not None:
            if geo == "sphere":
                return SphCoefs.readH5Coefs(file, stride, tmin, tmax)
            if geo == "cylinder":
                return CylCoefs.readH5Coefs(file, stride, tmin, tmax)
            raise RuntimeError("Coefs::factory: unknown H5 coefficient file geometry: " + geo)
        if not os.path.exists(file):
            raise RuntimeError("Coefs::factory: file <" + file + "> does not exist")
        with open(file, "rb") as f:
            head = f.read(4)
        magic = struct.unpack("<I", head)[0] if len(head) == 4 else 0
        if magic == CMAGIC_CYL:
            return CylCoefs.readNativeCoefs(file, stride, tmin, tmax)
        # (the reference sends every other magic to TableData; a legacy spherical stream starts with its
        # 64-character id instead and is read by SphCoefs there only when asked for explicitly)
        return SphCoefs.readNativeCoefs(file, stride, tmin, tmax)

    @staticmethod
    def makecoefs(coef, name: str = "") -> "Coefs":
        """``Coefs::makecoefs`` (expui/Coefficients.cc:3020-3043): an empty container of the structure's
        geometry."""
        if isinstance(coef, SphStruct):
            return SphCoefs(name)
        if isinstance(coef, CylStruct):
            return CylCoefs(name)
        raise RuntimeError("Coefs::makecoefs: cannot deduce coefficient file type")

    @staticmethod
    def addcoef(coefs: Optional["Coefs"], coef) -> "Coefs":
        """``Coefs::addcoef`` (:3046-3054)."""
        ret = coefs if coefs is not None else Coefs.makecoefs(coef)
        ret.add(coef)
        return ret

    # -- HDF5: extend an existing file (expui/Coefficients.cc:3165-3204) -------------------------------
    def _h5_arrays(self):
        times = self.Times()
        first = self.coefs[times[0]]
        rows = np.shape(first.coefs)[0]
        data = np.zeros((len(times), rows, first.nmax, 2))
        ctr = np.zeros((len(times), 3))
        rot = np.zeros((len(times), 3, 3))
        for k, t in enumerate(times):
            c = self.coefs[t]
            data[k, :, :, 0], data[k, :, :, 1] = np.real(c.coefs), np.imag(c.coefs)
            ctr[k] = np.asarray(c.ctr, dtype=np.float64).reshape(3) if np.size(c.ctr) == 3 else 0.0
            rot[k] = np.asarray(c.rot, dtype=np.float64).reshape(3, 3) if np.size(c.rot) == 9 else np.eye(3)
        tarr = np.array([self.coefs[t].time for t in times])
        return tarr, ctr, rot, data, rows, first.nmax

    def ExtendH5Coefs(self, path: str) -> None:
        """``Coefs::ExtendH5Coefs``: the stored sets are appended to an existing file after
        ``CheckH5Params`` (orders, scale to 1e-8, forceID: :855-905, :1334-1373); the snapshot numbering
        continues at the file's ``count``."""
        import ctypes
        from . import h5cache
        if not self.Times():
            return
        if not self._check_h5_params(path):
            raise RuntimeError("Coefs::ExtendH5Coefs: H5 parameter check failed, aborting extension")
        tarr, ctr, rot, data, rows, nmax = self._h5_arrays()
        lib = h5cache._load()
        lib.exp_h5_coef_extend.restype = ctypes.c_int
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        if lib.exp_h5_coef_extend(path.encode(), rows, nmax, len(tarr), vp(tarr), vp(ctr), vp(rot), vp(data)):
            raise RuntimeError(f"ExtendH5Coefs: HDF5 error extending <{path}>")


class SphCoefs(Coefs):
    """``CoefClasses::SphCoefs``: spherical coefficient sets keyed by (rounded) time."""

    geometry = "sphere"
    _what = "SphCoefs"

    def makeKeys(self, subkey=()) -> List[List[int]]:
        """``SphCoefs::makeKeys`` (expui/Coefficients.cc:750-792): every key [l, m, n] under a leading sub-key -- ``[]``: all,
        ``[l]``: all (m, n) of that l, ``[l, m]``: all n; l is clamped to Lmax, m to l; nothing without stored sets."""
        if not self.coefs:
            return []
        first = self.coefs[self.Times()[0]]
        Lmax, Nmax = first.lmax, first.nmax
        k = [int(v) for v in subkey]
        if len(k) > 2:
            raise RuntimeError("SphCoefs::makeKeys: the subkey must have rank 0, 1 or 2")
        if k:
            k[0] = min(max(k[0], 0), Lmax)
        if len(k) > 1:
            k[1] = min(max(k[1], 0), k[0])
        if not k:
            return [[l, m, n] for l in range(Lmax + 1) for m in range(l + 1) for n in range(Nmax)]
        if len(k) == 1:
            return [[k[0], m, n] for m in range(k[0] + 1) for n in range(Nmax)]
        return [[k[0], k[1], n] for n in range(Nmax)]

    def Power(self, min: int = 0, max: int = 2 ** 31 - 1) -> np.ndarray:
        """``SphCoefs::Power`` (expui/Coefficients.cc:1033-1060): [ntimes, lmax + 1], the sum over m and over
        the radial orders min <= n < max of |c|^2."""
        times = self.Times()
        if not times:
            return np.zeros((0, 0))
        first = self.coefs[times[0]]
        lo, hi = builtins_max(0, min), builtins_min(first.nmax, max)
        out = np.zeros((len(times), first.lmax + 1))
        for T, t in enumerate(times):
            a2 = np.abs(np.asarray(self.coefs[t].coefs)[:, lo:hi]) ** 2
            L = 0
            for l in range(first.lmax + 1):
                out[T, l] = a2[L:L + l + 1].sum()
                L += l + 1
        return out

    def _check_h5_params(self, path: str) -> bool:
        import ctypes
        from . import h5cache
        lib = h5cache._load()
        lib.exp_h5_sphcoef_info.restype = ctypes.c_int
        lmax, nmax, count, hasv = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        scale = ctypes.c_double()
        name, fid, geo = (ctypes.create_string_buffer(256) for _ in range(3))
        if lib.exp_h5_sphcoef_info(path.encode(), ctypes.byref(lmax), ctypes.byref(nmax), ctypes.byref(scale),
                                   ctypes.byref(count), name, 256, fid, 256, geo, 256, ctypes.byref(hasv)):
            return False
        first = self.coefs[self.Times()[0]]
        ok = lmax.value == first.lmax and nmax.value == first.nmax
        s0, s1 = float(first.scale), scale.value
        if abs(s0 - s1) > 1.0e-8 * builtins_max(abs(s0), abs(s1)):
            ok = False
        if fid.value.decode() != getattr(self, "_force_id", "sphereSL"):
            ok = False
        return ok

    # -- container (expui/Coefficients.H) -------------------------------------------------------
    def add(self, c: SphStruct) -> None:
        c._owner = self                      # CoefStruct::setOwner: the set reports the container's G from now on
        self.coefs[round_time(c.time)] = c

    def Times(self) -> List[float]:
        return sorted(self.coefs)

    def getCoefStruct(self, time: float) -> SphStruct:
        try:
            return self.coefs[round_time(time)]
        except KeyError:
            raise RuntimeError(f"SphCoefs: no coefficients at time {time}") from None

    def getAllCoefs(self) -> np.ndarray:
        """[(L+1)(L+2)/2, nmax, ntimes] complex (SphCoefs::getAllCoefs)."""
        return np.stack([np.asarray(self.coefs[t].coefs) for t in self.Times()], axis=2)

    # -- native stream files ---------------------------------------------------------------------
    @classmethod
    def readNativeCoefs(cls, path: str, stride: int = 1, tmin: float = -math.inf,
                        tmax: float = math.inf, name: str = "") -> "SphCoefs":
        """expui/Coefficients.cc:796-838"""
        self = cls(name)
        count = 0
        with open(path, "rb") as f:
            while True:
                c = read_native_record(f)
                if c is None:
                    break
                keep = count % stride == 0
                count += 1
                if keep and tmin <= c.time <= tmax:
                    self.add(c)
        return self

    def writeNativeCoefs(self, path: str, append: bool = False) -> None:
        with open(path, "ab" if append else "wb") as f:
            for t in self.Times():
                write_native(f, self.coefs[t])

    # -- playback --------------------------------------------------------------------------------
    def interpolate(self, time: float) -> Tuple[np.ndarray, bool]:
        """``Coefs::interpolate`` (expui/Coefficients.cc:183-226), statement for statement: the pair
        is (lower_bound, lower_bound + 1) -- NOT the bracketing pair when ``time`` lies strictly
        between two stored times, where the reference extrapolates linearly from the two sets at
        and after ``time`` -- or the last two sets at and beyond the end.  The flag is the
        reference's off-grid test (its count of 8 tolerated attempts is not kept here)."""
        times = self.Times()
        if len(times) < 2:
            raise RuntimeError("SphCoefs.interpolate: need at least two coefficient sets")
        on_grid = not (time < times[0] - self.deltaT or time > times[-1] + self.deltaT)
        it = int(np.searchsorted(times, time, side="left"))          # std::lower_bound
        if it >= len(times) - 1:
            hi, lo = len(times) - 1, len(times) - 2
        else:
            lo, hi = it, it + 1
        A = (times[hi] - time) / (times[hi] - times[lo])
        B = (time - times[lo]) / (times[hi] - times[lo])
        return A * np.asarray(self.coefs[times[lo]].coefs) + B * np.asarray(self.coefs[times[hi]].coefs), on_grid

    # -- HDF5 coefficient files (pyEXP's default format) -------------------------------------------
    def WriteH5Coefs(self, path: str, config: str = "", force_id: str = "sphereSL") -> None:
        """``Coefs::WriteH5Coefs`` + ``SphCoefs::WriteH5Params/WriteH5Times`` (expui/Coefficients.cc:
        3100-3163, :841-853, :907-944) through the HDF5 C shim (``exp_amd.h5cache``)."""
        import ctypes
        from . import h5cache
        times = self.Times()
        if not times:
            raise RuntimeError("Coefs::WriteH5Coefs: we have NO coefficient sets")
        first = self.coefs[times[0]]
        lmax, nmax = first.lmax, first.nmax
        ldim = (lmax + 1) * (lmax + 2) // 2
        data = np.zeros((len(times), ldim, nmax, 2))
        ctr = np.zeros((len(times), 3))
        rot = np.zeros((len(times), 3, 3))
        for k, t in enumerate(times):
            c = self.coefs[t]
            data[k, :, :, 0], data[k, :, :, 1] = np.real(c.coefs), np.imag(c.coefs)
            ctr[k] = np.asarray(c.ctr, dtype=np.float64).reshape(3) if np.size(c.ctr) == 3 else 0.0
            rot[k] = np.asarray(c.rot, dtype=np.float64).reshape(3, 3) if np.size(c.rot) == 9 else np.eye(3)
        tarr = np.array([self.coefs[t].time for t in times])
        self._force_id = force_id
        urec = self._units_for_h5()
        lib = h5cache._load()
        lib.exp_h5_coef_set_units(len(urec), urec.ctypes.data_as(ctypes.c_void_p))
        lib.exp_h5_sphcoef_write.restype = ctypes.c_int
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        rc = lib.exp_h5_sphcoef_write(path.encode(), self.name.encode(), config.encode(),
                                      force_id.encode(), lmax, nmax, ctypes.c_double(first.scale),
                                      len(times), vp(tarr), vp(ctr), vp(rot), vp(data))
        if rc:
            raise RuntimeError(f"WriteH5Coefs: HDF5 error writing <{path}>")

    @classmethod
    def readH5Coefs(cls, path: str, stride: int = 1, tmin: float = -math.inf,
                    tmax: float = math.inf) -> "SphCoefs":
        """The reading constructor ``SphCoefs(HighFive::File&, stride, Tmin, Tmax)``
        (expui/Coefficients.cc:228-330)."""
        import ctypes
        from . import h5cache
        lib = h5cache._load()
        lib.exp_h5_sphcoef_info.restype = ctypes.c_int
        lib.exp_h5_sphcoef_read.restype = ctypes.c_int
        lmax, nmax, count, hasv = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        scale = ctypes.c_double()
        name, fid, geo = (ctypes.create_string_buffer(256) for _ in range(3))
        if lib.exp_h5_sphcoef_info(path.encode(), ctypes.byref(lmax), ctypes.byref(nmax),
                                   ctypes.byref(scale), ctypes.byref(count), name, 256, fid, 256,
                                   geo, 256, ctypes.byref(hasv)):
            raise RuntimeError(f"readH5Coefs: <{path}> is not a readable coefficient file")
        if geo.value.decode() != "sphere":
            raise RuntimeError(f"readH5Coefs: geometry <{geo.value.decode()}> is not spherical")
        L, N, C = lmax.value, nmax.value, count.value
        ldim = (L + 1) * (L + 2) // 2
        times, ctr, rot = np.zeros(C), np.zeros((C, 3)), np.zeros((C, 3, 3))
        data = np.zeros((C, ldim, N, 2))
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        if lib.exp_h5_sphcoef_read(path.encode(), C, L, N, vp(times), vp(ctr), vp(rot), vp(data)):
            raise RuntimeError(f"readH5Coefs: <{path}>: snapshots missing or of the wrong shape")
        self = cls(name.value.decode())
        self._read_units_h5(path)
        self._force_id = fid.value.decode()
        for k in range(0, C, stride):
            if times[k] < tmin or times[k] > tmax:
                continue
            self.add(SphStruct(L, N, scale.value, float(times[k]), data[k, :, :, 0] + 1j * data[k, :, :, 1],
                               ctr[k].copy(), rot[k].copy()))
        return self


# ---- cylindrical basis -----------------------------------------------------------------------------
CMAGIC_CYL = 0x0C0A57A3              # expui/CoefStruct.cc:270 (EmpCylSL::dump_coefs_binary)
_LEGACY_CYL = struct.Struct("<dii")  # CylCoefHeader: time, mmax, nmax (include/coef.H)


def write_native_cyl(out: BinaryIO, c: CylStruct) -> None:
    """``EmpCylSL::dump_coefs_binary`` (exputil/EmpCylSL.cc:5868-5920), new-style header: magic, YAML
    {time, mmax, nmax}, then per m the cosine row and (m > 0) the sine row."""
    import yaml
    hdr = yaml.safe_dump({"time": float(c.time), "mmax": int(c.mmax), "nmax": int(c.nmax)},
                         default_flow_style=False, sort_keys=False).encode()
    out.write(struct.pack("<II", CMAGIC_CYL, len(hdr)))
    out.write(hdr)
    cf = np.asarray(c.coefs)
    for m in range(c.mmax + 1):
        out.write(np.ascontiguousarray(cf[m].real, dtype="<f8").tobytes())
        if m:
            out.write(np.ascontiguousarray(cf[m].imag, dtype="<f8").tobytes())


def read_native_cyl_record(inp: BinaryIO) -> Optional[CylStruct]:
    """``CylStruct::read`` (expui/CoefStruct.cc:258-370): new-style or 16-byte legacy header."""
    import yaml
    head = inp.read(4)
    if len(head) < 4:
        return None
    if struct.unpack("<I", head)[0] == CMAGIC_CYL:
        raw = inp.read(4)
        if len(raw) < 4:
            return None
        node = yaml.safe_load(inp.read(struct.unpack("<I", raw)[0]).decode())
        time, mmax, nmax = float(node["time"]), int(node["mmax"]), int(node["nmax"])
    else:
        rest = inp.read(_LEGACY_CYL.size - 4)
        if len(rest) < _LEGACY_CYL.size - 4:
            return None
        time, mmax, nmax = _LEGACY_CYL.unpack(head + rest)
    nrow = 2 * mmax + 1
    buf = inp.read(8 * nrow * nmax)
    if len(buf) < 8 * nrow * nmax:
        return None
    rows = np.frombuffer(buf, dtype="<f8").reshape(nrow, nmax)
    cf = np.zeros((mmax + 1, nmax), dtype=np.complex128)
    k = 0
    for m in range(mmax + 1):
        cf[m] = rows[k]
        k += 1
        if m:
            cf[m] = cf[m] + 1j * rows[k]
            k += 1
    return CylStruct(mmax, nmax, time, cf, np.zeros(3), np.eye(3))


class CylCoefs(Coefs):
    """``CoefClasses::CylCoefs``: cylindrical coefficient sets keyed by (rounded) time."""

    geometry = "cylinder"
    _what = "CylCoefs"

    def makeKeys(self, subkey=()) -> List[List[int]]:
        """``CylCoefs::makeKeys`` (expui/Coefficients.cc:1257-1287): every key [m, n], or those of one m (clamped to Mmax)."""
        if not self.coefs:
            return []
        first = self.coefs[self.Times()[0]]
        Mmax, Nmax = first.mmax, first.nmax
        k = [int(v) for v in subkey]
        if len(k) > 1:
            raise RuntimeError("CylCoefs::makeKeys: the subkey must have rank 1")
        if not k:
            return [[m, n] for m in range(Mmax + 1) for n in range(Nmax)]
        m = min(max(k[0], 0), Mmax)
        return [[m, n] for n in range(Nmax)]

    def Power(self, min: int = 0, max: int = 2 ** 31 - 1) -> np.ndarray:
        """``CylCoefs::Power`` (expui/Coefficients.cc:1442-1470): [ntimes, mmax + 1]."""
        times = self.Times()
        if not times:
            return np.zeros((0, 0))
        first = self.coefs[times[0]]
        lo, hi = builtins_max(0, min), builtins_min(first.nmax, max)
        return np.stack([(np.abs(np.asarray(self.coefs[t].coefs)[:, lo:hi]) ** 2).sum(axis=1) for t in times])

    def EvenOddPower(self, nodd: int = -1, min: int = 0, max: int = 2 ** 31 - 1):
        """``CylCoefs::EvenOddPower`` (:1472-1537): the radial orders split at nmax - nodd into the
        vertically even and odd families; ``nodd`` must be given (this mirror keeps no YAML config to take
        ``ncylodd`` from)."""
        times = self.Times()
        if not times:
            return np.zeros((0, 0)), np.zeros((0, 0))
        if nodd < 0:
            raise RuntimeError("CylCoefs::EvenOddPower: ncylodd is not in the YAML config stanza.  Please "
                               "specify this explicitly as the first argument to EvenOddPower()")
        first = self.coefs[times[0]]
        cut = first.nmax - nodd
        ev = np.stack([(np.abs(np.asarray(self.coefs[t].coefs)[:, builtins_max(0, min):builtins_min(cut, max)]) ** 2).sum(axis=1)
                       for t in times])
        od = np.stack([(np.abs(np.asarray(self.coefs[t].coefs)[:, builtins_max(cut, min):builtins_min(first.nmax, max)]) ** 2).sum(axis=1)
                       for t in times])
        return ev, od

    def _check_h5_params(self, path: str) -> bool:
        import ctypes
        from . import h5cache
        lib = h5cache._load()
        lib.exp_h5_cylcoef_info.restype = ctypes.c_int
        mmax, nmax, count, hasv = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        name, fid, geo = (ctypes.create_string_buffer(256) for _ in range(3))
        if lib.exp_h5_cylcoef_info(path.encode(), ctypes.byref(mmax), ctypes.byref(nmax), ctypes.byref(count),
                                   name, 256, fid, 256, geo, 256, ctypes.byref(hasv)):
            return False
        first = self.coefs[self.Times()[0]]
        return (mmax.value == first.mmax and nmax.value == first.nmax
                and fid.value.decode() == getattr(self, "_force_id", "cylinder"))

    def add(self, c: CylStruct) -> None:
        c._owner = self
        self.coefs[round_time(c.time)] = c

    def Times(self) -> List[float]:
        return sorted(self.coefs)

    def getCoefStruct(self, time: float) -> CylStruct:
        try:
            return self.coefs[round_time(time)]
        except KeyError:
            raise RuntimeError(f"CylCoefs: no coefficients at time {time}") from None

    def getAllCoefs(self) -> np.ndarray:
        return np.stack([np.asarray(self.coefs[t].coefs) for t in self.Times()], axis=2)

    @classmethod
    def readNativeCoefs(cls, path: str, stride: int = 1, tmin: float = -math.inf,
                        tmax: float = math.inf, name: str = "") -> "CylCoefs":
        self = cls(name)
        count = 0
        with open(path, "rb") as f:
            while True:
                c = read_native_cyl_record(f)
                if c is None:
                    break
                keep = count % stride == 0
                count += 1
                if keep and tmin <= c.time <= tmax:
                    self.add(c)
        return self

    def writeNativeCoefs(self, path: str, append: bool = False) -> None:
        with open(path, "ab" if append else "wb") as f:
            for t in self.Times():
                write_native_cyl(f, self.coefs[t])

    interpolate = SphCoefs.interpolate          # Coefs::interpolate is geometry-independent

    def setDeltaT(self, dT: float) -> None:
        self.deltaT = float(dT)

    # -- HDF5 coefficient files --------------------------------------------------------------------
    def WriteH5Coefs(self, path: str, config: str = "", force_id: str = "cylinder") -> None:
        """``Coefs::WriteH5Coefs`` + ``CylCoefs::WriteH5Params/WriteH5Times`` (expui/Coefficients.cc:
        3100-3163, :1323-1332, :1375-1405): attributes mmax / nmax / forceID, one (mmax+1) x nmax
        complex ``coefficients`` dataset with Time / Center / Rotation per snapshot."""
        import ctypes
        from . import h5cache
        times = self.Times()
        if not times:
            raise RuntimeError("Coefs::WriteH5Coefs: we have NO coefficient sets")
        first = self.coefs[times[0]]
        mmax, nmax = first.mmax, first.nmax
        data = np.zeros((len(times), mmax + 1, nmax, 2))
        ctr = np.zeros((len(times), 3))
        rot = np.zeros((len(times), 3, 3))
        for k, t in enumerate(times):
            c = self.coefs[t]
            data[k, :, :, 0], data[k, :, :, 1] = np.real(c.coefs), np.imag(c.coefs)
            ctr[k] = np.asarray(c.ctr, dtype=np.float64).reshape(3) if np.size(c.ctr) == 3 else 0.0
            rot[k] = np.asarray(c.rot, dtype=np.float64).reshape(3, 3) if np.size(c.rot) == 9 else np.eye(3)
        tarr = np.array([self.coefs[t].time for t in times])
        self._force_id = force_id
        urec = self._units_for_h5()
        lib = h5cache._load()
        lib.exp_h5_coef_set_units(len(urec), urec.ctypes.data_as(ctypes.c_void_p))
        lib.exp_h5_cylcoef_write.restype = ctypes.c_int
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        if lib.exp_h5_cylcoef_write(path.encode(), self.name.encode(), config.encode(),
                                    force_id.encode(), mmax, nmax, len(times), vp(tarr), vp(ctr),
                                    vp(rot), vp(data)):
            raise RuntimeError(f"WriteH5Coefs: HDF5 error writing <{path}>")

    @classmethod
    def readH5Coefs(cls, path: str, stride: int = 1, tmin: float = -math.inf,
                    tmax: float = math.inf) -> "CylCoefs":
        """The reading constructor ``CylCoefs(HighFive::File&, stride, Tmin, Tmax)``
        (expui/Coefficients.cc:1075-1176), files written with CoefficientOutputVersion (the legacy
        transposed storage, ``H5back``, is refused); the m = 0 row is forced real as there (:1157)."""
        import ctypes
        from . import h5cache
        lib = h5cache._load()
        lib.exp_h5_cylcoef_info.restype = ctypes.c_int
        lib.exp_h5_cylcoef_read.restype = ctypes.c_int
        mmax, nmax, count, hasv = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        name, fid, geo = (ctypes.create_string_buffer(256) for _ in range(3))
        if lib.exp_h5_cylcoef_info(path.encode(), ctypes.byref(mmax), ctypes.byref(nmax),
                                   ctypes.byref(count), name, 256, fid, 256, geo, 256,
                                   ctypes.byref(hasv)):
            raise RuntimeError(f"readH5Coefs: <{path}> is not a readable cylindrical coefficient file")
        if geo.value.decode() != "cylinder":
            raise RuntimeError(f"readH5Coefs: geometry <{geo.value.decode()}> is not cylindrical")
        if not hasv.value:
            raise RuntimeError("readH5Coefs: legacy (pre-CoefficientOutputVersion) storage order is not supported")
        M, N, C = mmax.value, nmax.value, count.value
        times, ctr, rot = np.zeros(C), np.zeros((C, 3)), np.zeros((C, 3, 3))
        data = np.zeros((C, M + 1, N, 2))
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        if lib.exp_h5_cylcoef_read(path.encode(), C, M, N, vp(times), vp(ctr), vp(rot), vp(data)):
            raise RuntimeError(f"readH5Coefs: <{path}>: snapshots missing or of the wrong shape")
        self = cls(name.value.decode())
        self._read_units_h5(path)
        self._force_id = fid.value.decode()
        for k in range(0, C, stride):
            if times[k] < tmin or times[k] > tmax:
                continue
            cf = data[k, :, :, 0] + 1j * data[k, :, :, 1]
            cf[0] = cf[0].real
            self.add(CylStruct(M, N, float(times[k]), cf, ctr[k].copy(), rot[k].copy()))
        return self
