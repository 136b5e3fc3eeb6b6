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


class SphCoefs:
    """``CoefClasses::SphCoefs``: spherical coefficient sets keyed by (rounded) time."""

    geometry = "sphere"

    def __init__(self, name: str = ""):
        self.name = name
        self.coefs: Dict[float, SphStruct] = {}
        self.deltaT = 0.01                       # expui/Coefficients.H:133

    def setDeltaT(self, dT: float) -> None:
        self.deltaT = float(dT)

    # -- container (expui/Coefficients.H) -------------------------------------------------------
    def add(self, c: SphStruct) -> None:
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
        lib = h5cache._load()
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


class CylCoefs:
    """``CoefClasses::CylCoefs``: cylindrical coefficient sets keyed by (rounded) time."""

    geometry = "cylinder"

    def __init__(self, name: str = ""):
        self.name = name
        self.coefs: Dict[float, CylStruct] = {}
        self.deltaT = 0.01

    def add(self, c: CylStruct) -> None:
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
        lib = h5cache._load()
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
        for k in range(0, C, stride):
            if times[k] < tmin or times[k] > tmax:
                continue
            cf = data[k, :, :, 0] + 1j * data[k, :, :, 1]
            cf[0] = cf[0].real
            self.add(CylStruct(M, N, float(times[k]), cf, ctr[k].copy(), rot[k].copy()))
        return self
