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
