"""EXP's HDF5 basis caches: read / write the SLGridSph cache file of
``SLGridSph::WriteH5Cache`` / ``ReadH5Cache`` (exputil/SLGridMP2.cc:490-696) so that a basis
built by EXP can drive these kernels, and one built here can be handed to EXP.

The image has no h5py/HighFive, so the file is handled by a small C shim over the HDF5 C library
(``exp_amd/csrc_host/h5cache.c`` -> ``exp_amd/libexp_amd_h5.so``, built by ``make h5`` where
``hdf5.h`` exists).  The cache holds only eigenvalues and eigenfunctions; the radial grid and the
background potential/density on it are recomputed from the model, as the reference does
(``SLGridSph::init_table`` / ``:1355-1382``).
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Optional

import numpy as np

from .slgrid import SLGridSph, _xi_grid

_HERE = os.path.dirname(os.path.abspath(__file__))


class _Hdr(ctypes.Structure):
    _fields_ = [("geometry", ctypes.c_char * 64), ("forceID", ctypes.c_char * 64),
                ("version", ctypes.c_char * 32), ("model", ctypes.c_char * 512),
                ("lmax", ctypes.c_int), ("nmax", ctypes.c_int), ("numr", ctypes.c_int),
                ("cmap", ctypes.c_int), ("diverge", ctypes.c_int),
                ("rmin", ctypes.c_double), ("rmax", ctypes.c_double),
                ("rmapping", ctypes.c_double), ("dfac", ctypes.c_double)]


_lib = None


def available() -> bool:
    return os.path.exists(os.path.join(_HERE, "libexp_amd_h5.so"))


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libexp_amd_h5.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make h5` (needs hdf5.h / libhdf5)")
        _lib = ctypes.CDLL(path)
        for fn in ("exp_h5_slgrid_write", "exp_h5_slgrid_write_ex", "exp_h5_slgrid_read_header",
                   "exp_h5_slgrid_read_tables"):
            getattr(_lib, fn).restype = ctypes.c_int
    return _lib


def write_slgrid_cache(path: str, g: SLGridSph, model_name: str, diverge: int = 0,
                       dfac: float = 1.0, old_layout: bool = False) -> None:
    """``SLGridSph::WriteH5Cache`` (exputil/SLGridMP2.cc:622-696); ``model_name`` is the model
    file name the reference stores and later compares (``model`` attribute)."""
    h = _Hdr()
    h.version = b"1.0"
    h.model = model_name.encode()
    h.lmax, h.nmax, h.numr, h.cmap, h.diverge = g.lmax, g.nmax, g.numr, g.cmap, int(diverge)
    h.rmin, h.rmax, h.rmapping, h.dfac = g.rmin, g.rmax, g.rmap, float(dfac)
    ev = np.ascontiguousarray(g.ev, dtype=np.float64)
    ef = np.ascontiguousarray(g.ef, dtype=np.float64)
    rc = _load().exp_h5_slgrid_write_ex(path.encode(), ctypes.byref(h),
                                        ev.ctypes.data_as(ctypes.c_void_p),
                                        ef.ctypes.data_as(ctypes.c_void_p), int(old_layout))
    if rc:
        raise RuntimeError(f"write_slgrid_cache: HDF5 error writing <{path}>")


def read_slgrid_header(path: str) -> dict:
    h = _Hdr()
    if _load().exp_h5_slgrid_read_header(path.encode(), ctypes.byref(h)):
        raise RuntimeError(f"read_slgrid_header: <{path}> is not a readable SLGridSph cache")
    out = {k: getattr(h, k) for k, _ in _Hdr._fields_}
    for k in ("geometry", "forceID", "version", "model"):
        out[k] = out[k].decode()
    return out


def read_slgrid_cache(path: str, model, check: Optional[dict] = None) -> SLGridSph:
    """``SLGridSph::ReadH5Cache`` (exputil/SLGridMP2.cc:490-620): the tables of the file on the
    radial grid implied by its header, with p0 / d0 recomputed from ``model``.  ``check`` may
    hold wanted header values (lmax, nmax, numr, cmap, rmin, rmax, rmapping, model ...): a
    mismatch raises, where the reference silently rebuilds."""
    h = read_slgrid_header(path)
    if h["geometry"] != "sphere" or h["forceID"] != "SLGridSph":
        raise RuntimeError(f"<{path}>: geometry/forceID = {h['geometry']}/{h['forceID']}")
    for k, v in (check or {}).items():
        have = h[k]
        bad = (abs(have - v) >= 1.0e-16) if isinstance(v, float) else (have != v)
        if bad:
            raise RuntimeError(f"<{path}>: parameter {k}: wanted {v} found {have}")
    lmax, nmax, numr = h["lmax"], h["nmax"], h["numr"]
    ev = np.zeros((lmax + 1, nmax))
    ef = np.zeros((lmax + 1, nmax, numr))
    if _load().exp_h5_slgrid_read_tables(path.encode(), lmax, nmax, numr,
                                         ev.ctypes.data_as(ctypes.c_void_p),
                                         ef.ctypes.data_as(ctypes.c_void_p)):
        raise RuntimeError(f"<{path}>: Harmonic/<l>/ev|ef missing or of the wrong shape")
    xmin, xmax, dxi, xi, r = _xi_grid(h["cmap"], h["rmin"], h["rmax"], h["rmapping"], numr)
    p0 = np.asarray(model.pot(r), dtype=np.float64)
    d0 = 4.0 * math.pi * np.asarray(model.dens(r), dtype=np.float64)
    return SLGridSph(lmax=lmax, nmax=nmax, numr=numr, cmap=h["cmap"], rmin=h["rmin"], rmax=h["rmax"],
                     rmap=h["rmapping"], xmin=xmin, xmax=xmax, dxi=dxi, xi=xi, r=r, p0=p0, d0=d0,
                     ev=ev, ef=ef)


# ---- EmpCylSL cache (exputil/EmpCylSL.cc:7378-7460, :7486-7640) -----------------------------------

class _CylHdr(ctypes.Structure):
    _fields_ = [("geometry", ctypes.c_char * 64), ("forceID", ctypes.c_char * 64),
                ("version", ctypes.c_char * 32), ("model", ctypes.c_char * 128),
                ("mmax", ctypes.c_int), ("numx", ctypes.c_int), ("numy", ctypes.c_int),
                ("nmax", ctypes.c_int), ("lmaxfid", ctypes.c_int), ("nmaxfid", ctypes.c_int),
                ("neven", ctypes.c_int), ("nodd", ctypes.c_int), ("cmapr", ctypes.c_int),
                ("cmapz", ctypes.c_int), ("rmin", ctypes.c_double), ("rmax", ctypes.c_double),
                ("ascl", ctypes.c_double), ("hscl", ctypes.c_double), ("cmass", ctypes.c_double)]


def write_empcyl_cache(path: str, g, lmaxfid: int, nmaxfid: int, model: str = "Exponential",
                       neven: Optional[int] = None, nodd: int = 0, cmass: float = 0.0) -> None:
    """``EmpCylSL::WriteH5Cache``: ``g`` is an ``exp_amd.empcyl.EmpCylGrid`` WITH density tables."""
    if g.dens is None:
        raise RuntimeError("write_empcyl_cache: the grid has no density tables")
    h = _CylHdr()
    h.version, h.model = b"1.0", model.encode()
    h.mmax, h.numx, h.numy, h.nmax = g.mmax, g.numx, g.numy, g.norder
    h.lmaxfid, h.nmaxfid = int(lmaxfid), int(nmaxfid)
    h.neven, h.nodd = (g.norder - nodd if neven is None else int(neven)), int(nodd)
    h.cmapr, h.cmapz = g.cmapr, g.cmapz
    h.rmin, h.rmax, h.ascl, h.hscl, h.cmass = g.rmin, g.rmax, g.ascale, g.hscale, float(cmass)
    tab = np.ascontiguousarray(g.tab, dtype=np.float64)
    dens = np.ascontiguousarray(g.dens, dtype=np.float64)
    lib = _load()
    lib.exp_h5_cyl_write.restype = ctypes.c_int
    if lib.exp_h5_cyl_write(path.encode(), ctypes.byref(h), tab.ctypes.data_as(ctypes.c_void_p),
                            dens.ctypes.data_as(ctypes.c_void_p)):
        raise RuntimeError(f"write_empcyl_cache: HDF5 error writing <{path}>")


def read_empcyl_header(path: str) -> dict:
    h = _CylHdr()
    lib = _load()
    lib.exp_h5_cyl_read_header.restype = ctypes.c_int
    if lib.exp_h5_cyl_read_header(path.encode(), ctypes.byref(h)):
        raise RuntimeError(f"read_empcyl_header: <{path}> is not a readable EmpCylSL cache")
    out = {k: getattr(h, k) for k, _ in _CylHdr._fields_}
    for k in ("geometry", "forceID", "version", "model"):
        out[k] = out[k].decode()
    return out


def read_empcyl_cache(path: str, check: Optional[dict] = None):
    """``EmpCylSL::ReadH5Cache``: an ``EmpCylGrid`` with the file's tables; the grid geometry
    (Rtable, XMIN.., YMIN.., ``setup_table`` exputil/EmpCylSL.cc:2123-2137) follows from the header."""
    from .empcyl import EmpCylGrid, r_to_xi, z_to_y
    h = read_empcyl_header(path)
    if h["geometry"] != "cylinder" or h["forceID"] != "Cylinder":
        raise RuntimeError(f"<{path}>: geometry/forceID = {h['geometry']}/{h['forceID']}")
    for k, v in (check or {}).items():
        have = h[k]
        bad = (abs(have - v) >= 1.0e-16) if isinstance(v, float) else (have != v)
        if bad:
            raise RuntimeError(f"<{path}>: parameter {k}: wanted {v} found {have}")
    mmax, nmax, numx, numy = h["mmax"], h["nmax"], h["numx"], h["numy"]
    tab = np.zeros((6, mmax + 1, nmax, numx + 1, numy + 1))
    dens = np.zeros((2, mmax + 1, nmax, numx + 1, numy + 1))
    lib = _load()
    lib.exp_h5_cyl_read_tables.restype = ctypes.c_int
    if lib.exp_h5_cyl_read_tables(path.encode(), mmax, nmax, numx, numy,
                                  tab.ctypes.data_as(ctypes.c_void_p),
                                  dens.ctypes.data_as(ctypes.c_void_p)):
        raise RuntimeError(f"<{path}>: Cosine|Sine/<m>/<n> tables missing or of the wrong shape")
    A, H = h["ascl"], h["hscl"]
    rtable = math.sqrt(0.5) * h["rmax"]
    xmin = float(r_to_xi(h["rmin"] * A, A, h["cmapr"]))
    xmax = float(r_to_xi(rtable * A, A, h["cmapr"]))
    ymin = float(z_to_y(-rtable * A, H, h["cmapz"]))
    ymax = float(z_to_y(rtable * A, H, h["cmapz"]))
    return EmpCylGrid(mmax=mmax, norder=nmax, numx=numx, numy=numy, cmapr=h["cmapr"], cmapz=h["cmapz"],
                      ascale=A, hscale=H, rmin=h["rmin"], rmax=h["rmax"], rtable=rtable, xmin=xmin,
                      xmax=xmax, dx=(xmax - xmin) / numx, ymin=ymin, ymax=ymax, dy=(ymax - ymin) / numy,
                      tab=tab, dens=dens)


# ---- coefficient covariance store (expui/Covariance.cc, include/Covariance.H) --------------------------

def covar_append(path: str, basis_id: str, kind: int, ipar, dpar, time: float, counts, masses, mean, covr=None,
                 summed: bool = True, covar: bool = True) -> bool:
    """``SubsampleCovariance::writeCoefCovariance(fname, elem, time)``: create the file (version "1.1",
    FloatSize 8) or append one snapshot to it.  mean [T, ltot, nmax] complex, covr [T, ltot, nmax, nmax]
    complex or None; summed / covar as the constructor flags (expui/Covariance.cc:9-15).  Returns False
    when there is no data (nothing written, as the reference)."""
    lib = _load()
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    masses = np.ascontiguousarray(masses, dtype=np.float64)
    mean = np.asarray(mean, dtype=np.complex128)
    T, ltot, nmax = mean.shape
    m2 = np.ascontiguousarray(np.stack([mean.real, mean.imag], -1))
    c2 = None
    if covr is not None:
        covr = np.asarray(covr, dtype=np.complex128)
        c2 = np.ascontiguousarray(np.stack([covr.real, covr.imag], -1))
    ip = (ctypes.c_int * 2)(*[int(v) for v in ipar])
    dp = (ctypes.c_double * 5)(*([float(v) for v in dpar] + [0.0] * (5 - len(dpar))))
    lib.exp_h5_covar_append.restype = ctypes.c_int
    rc = lib.exp_h5_covar_append(path.encode(), basis_id.encode(), int(kind), ip, dp, int(summed), int(covar),
                                 ctypes.c_double(time), int(T), int(ltot), int(nmax),
                                 counts.ctypes.data_as(ctypes.c_void_p), masses.ctypes.data_as(ctypes.c_void_p),
                                 m2.ctypes.data_as(ctypes.c_void_p),
                                 None if c2 is None else c2.ctypes.data_as(ctypes.c_void_p))
    if rc == -2:
        raise RuntimeError(f"SubsampleCovariance::writeCoefCovariance: <{path}> exists and is not a covariance "
                           "file (not HDF5, or an HDF5 file with its own `count` / `snapshots`): left untouched")
    if rc < 0:
        raise RuntimeError(f"SubsampleCovariance::writeCoefCovariance: cannot write <{path}>")
    return rc == 0


def covar_set_compress(level: int, chunksize: int, shuffle: bool, szip: bool = False) -> None:
    """``SubsampleCovariance::setCovarH5Compress`` (include/Covariance.H:147-153): deflate level (0: none),
    chunk size and shuffle filter of the datasets ``covar_append`` writes from now on."""
    lib = _load()
    lib.exp_h5_covar_set_compress.restype = ctypes.c_int
    if lib.exp_h5_covar_set_compress(ctypes.c_uint(int(level)), ctypes.c_uint(int(chunksize)), int(bool(shuffle)),
                                     int(bool(szip))):
        raise RuntimeError("setCovarH5Compress: level must be 0..9, chunksize > 0, and szip is not available "
                           "in this HDF5 build")


def coef_geometry(path: str):
    """The ``geometry`` attribute of an HDF5 coefficient file, or None when ``path`` is not one
    (``Coefs::factory``, expui/Coefficients.cc:2917-2931)."""
    lib = _load()
    buf = ctypes.create_string_buffer(64)
    lib.exp_h5_coef_geometry.restype = ctypes.c_int
    if lib.exp_h5_coef_geometry(str(path).encode(), buf, 64):
        return None
    return buf.value.decode()


class SubsampleCovariance:
    """``BasisClasses::SubsampleCovariance(filename, stride)`` (expui/Covariance.cc:419-700): reads a
    covariance file back; ``Times()``, ``getCoefCovariance(time)`` -> (counts, masses, mean [T, ltot,
    nmax], covr [T, ltot, nmax, nmax]).  A summed file gives every sub-sample the total divided by the
    number of sub-samples, as the reference's reader does (:685-695)."""

    def __init__(self, filename: str, stride: int = 1):
        lib = _load()
        bid, ver = ctypes.create_string_buffer(64), ctypes.create_string_buffer(32)
        iv = [ctypes.c_int() for _ in range(8)]
        lib.exp_h5_covar_info.restype = ctypes.c_int
        if lib.exp_h5_covar_info(filename.encode(), bid, 64, ver, 32, *[ctypes.byref(v) for v in iv]):
            raise RuntimeError(f"SubsampleCovariance: cannot read <{filename}>")
        self.BasisID, version = bid.value.decode(), ver.value.decode()
        if version == "1.0":
            raise RuntimeError("SubsampleCovariance: this is an early alpha test version. Please remake your files")
        if version != "1.1":
            raise RuntimeError(f"SubsampleCovariance: unsupported file version, {version}")
        fsz, count, T, ltot, nmax, summed, has_cov, full = [v.value for v in iv]
        if fsz != 8:
            raise RuntimeError(f"SubsampleCovariance: float size {fsz} is not supported by this reader")
        self.summed, self.times, self._data = bool(summed), [], {}
        diag = nmax * (nmax + 1) // 2 if full else nmax
        lib.exp_h5_covar_read.restype = ctypes.c_int
        for k in range(0, count, max(1, int(stride))):
            t = ctypes.c_double()
            counts, masses = np.zeros(T, dtype=np.int32), np.zeros(T)
            mean = np.zeros((T, ltot, nmax, 2))
            nv = ltot * diag * (1 if summed else T)
            cre, cim = (np.zeros(nv), np.zeros(nv)) if has_cov else (None, None)
            if lib.exp_h5_covar_read(filename.encode(), k, T, ltot, nmax, int(summed), ctypes.byref(t),
                                     counts.ctypes.data_as(ctypes.c_void_p), masses.ctypes.data_as(ctypes.c_void_p),
                                     mean.ctypes.data_as(ctypes.c_void_p),
                                     None if cre is None else cre.ctypes.data_as(ctypes.c_void_p),
                                     None if cim is None else cim.ctypes.data_as(ctypes.c_void_p)):
                raise RuntimeError(f"SubsampleCovariance: cannot read snapshot {k} of <{filename}>")
            covr = None
            if has_cov:
                flat = (cre + 1j * cim).reshape((1 if summed else T), ltot, diag)
                covr = np.zeros((flat.shape[0], ltot, nmax, nmax), dtype=np.complex128)
                if full:                              # upper triangles; the reference's reader copies them
                    iu = np.triu_indices(nmax)        # into the lower ones unconjugated (:663-666)
                    covr[:, :, iu[0], iu[1]] = flat
                    lower = np.swapaxes(covr, 2, 3).copy()
                    idx = np.arange(nmax)
                    lower[:, :, idx, idx] = 0.0
                    covr = covr + lower
                else:
                    idx = np.arange(nmax)
                    covr[:, :, idx, idx] = flat
                if summed:
                    covr = np.repeat(covr / T, T, axis=0)
            tt = float(np.floor(t.value * 1e8 + 0.5) / 1e8)
            self.times.append(tt)
            self._data[tt] = (counts, masses, mean[..., 0] + 1j * mean[..., 1], covr)

    def Times(self):
        return list(self.times)

    def getCoefCovariance(self, time: float):
        key = float(np.floor(time * 1e8 + 0.5) / 1e8)
        if key not in self._data:
            raise RuntimeError("SubsampleCovariance::getCoefCovariance: time not found")
        return self._data[key]
