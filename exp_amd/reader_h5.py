"""The HDF5 phase-space formats of ``pyEXP.read`` -- ``PSPhdf5`` (EXP's own OutHDF5 snapshots, both layouts) and
``GadgetHDF5`` (exputil/ParticleReader.cc:333-690, :730-1270) -- and the writer of the former (``write_psp_hdf5``:
OutHDF5::RunGadget4 / RunPSP, src/OutHDF5.cc:400-780; Component::write_HDF5 / write_H5, src/Component.cc:2456-2690).

The image has neither h5py nor HighFive: the files go through the C shim ``exp_amd/csrc_host/h5part.c`` in
``exp_amd/libexp_amd_h5.so`` (``make h5``)."""
from __future__ import annotations

import ctypes
import os
import sys
from typing import Dict, List, Optional, Sequence

import numpy as np

from .reader import GADGET_TYPES, PSP, ParticleReader, _Gadget, _two_d

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _h5():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libexp_amd_h5.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make h5` (needs hdf5.h / libhdf5)")
        _lib = ctypes.CDLL(path)
    return _lib


def _vp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def exists(path: str, obj: str) -> bool:
    return _h5().exp_h5p_exists(path.encode(), obj.encode()) == 1


def attr_numbers(path: str, obj: str, name: str) -> np.ndarray:
    out = np.zeros(64)
    n = ctypes.c_int(0)
    rc = _h5().exp_h5p_attr_f64(path.encode(), obj.encode(), name.encode(), _vp(out), 64, ctypes.byref(n))
    if rc != 0:
        raise RuntimeError(f"error reading HDF5 file, attribute <{name}> of <{obj}> in <{path}> ({rc})")
    return out[: n.value].copy()


def attr_strings(path: str, obj: str, name: str) -> List[str]:
    buf = ctypes.create_string_buffer(1 << 20)
    cnt = ctypes.c_int(0)
    out, i = [], 0
    while True:
        rc = _h5().exp_h5p_attr_str(path.encode(), obj.encode(), name.encode(), i, buf, len(buf), ctypes.byref(cnt))
        if rc != 0:
            raise RuntimeError(f"error reading HDF5 file, attribute <{name}> of <{obj}> in <{path}> ({rc})")
        if cnt.value == 0:
            break
        out.append(buf.value.decode())
        i += 1
        if i >= cnt.value:
            break
    return out


def dset_shape(path: str, dset: str):
    rank, dims, store = ctypes.c_int(0), (ctypes.c_longlong * 4)(), ctypes.c_longlong(0)
    rc = _h5().exp_h5p_dset_shape(path.encode(), dset.encode(), ctypes.byref(rank), dims, ctypes.byref(store))
    if rc != 0:
        raise RuntimeError(f"error reading HDF5 file, dataset <{dset}> in <{path}> ({rc})")
    return tuple(int(dims[k]) for k in range(rank.value)), int(store.value)


_KIND = {"d": np.float64, "f": np.float32, "i": np.int32, "u": np.uint32, "l": np.int64, "L": np.uint64}


def dset_read(path: str, dset: str, kind: str) -> np.ndarray:
    shape, _ = dset_shape(path, dset)
    out = np.zeros(shape, dtype=_KIND[kind])
    if out.size:
        rc = _h5().exp_h5p_dset_read(path.encode(), dset.encode(), ctypes.c_char(kind.encode()), _vp(out))
        if rc != 0:
            raise RuntimeError(f"error reading HDF5 file, dataset <{dset}> in <{path}> ({rc})")
    return out


class GadgetHDF5(_Gadget):
    """exputil/ParticleReader.cc:333-690: /Header {Time, MassTable, NumPart_ThisFile}, /PartType<k> {Coordinates,
    Velocities, Masses (optional: overrides the table), ParticleIDs}; everything is read as float / unsigned 32-bit,
    whatever the file holds (``H5::PredType::NATIVE_FLOAT`` / ``NATIVE_UINT32``); a ParticleIDs dataset without storage
    numbers the particles from 1."""

    _who = "GadgetHDF5"

    def _get_numbers(self) -> None:
        found = set()
        for name in self._files:
            try:
                self.time = float(attr_numbers(name, "/Header", "Time")[0])
                npart = attr_numbers(name, "/Header", "NumPart_ThisFile")
            except RuntimeError as e:
                sys.stderr.write(str(e) + "\n")             # (the reference prints the HDF5 error stack and goes on)
                continue
            found |= {GADGET_TYPES[k] for k in range(min(6, len(npart))) if npart[k] > 0}
        self.Pfound = sorted(found)

    def _count(self, name: str) -> int:
        try:
            return int(attr_numbers(name, "/Header", "NumPart_ThisFile")[self.ptype])
        except (RuntimeError, IndexError):
            return 0

    def _read_file(self, name: str) -> Dict[str, np.ndarray]:
        self.time = float(attr_numbers(name, "/Header", "Time")[0])
        table = attr_numbers(name, "/Header", "MassTable")
        npart = attr_numbers(name, "/Header", "NumPart_ThisFile")
        empty = {"mass": np.zeros(0), "pos": np.zeros((0, 3)), "vel": np.zeros((0, 3)), "indx": np.zeros(0, np.uint64)}
        if not npart[self.ptype] > 0:
            sys.stderr.write(f"GadgetHDF5:: zero pass particles for type <{GADGET_TYPES[self.ptype]}>\n")
            return empty
        grp = f"/PartType{self.ptype}"
        pos = dset_read(name, grp + "/Coordinates", "f").astype(np.float64)
        vel = dset_read(name, grp + "/Velocities", "f").astype(np.float64)
        n = len(pos)
        mass = np.full(n, float(table[self.ptype]))
        if exists(name, grp + "/Masses") and dset_shape(name, grp + "/Masses")[1] > 0:
            mass = dset_read(name, grp + "/Masses", "f").astype(np.float64).reshape(-1)
        if dset_shape(name, grp + "/ParticleIDs")[1] > 0:
            ids = dset_read(name, grp + "/ParticleIDs", "u").astype(np.uint64).reshape(-1)
        else:
            ids = np.arange(1, n + 1, dtype=np.uint64)
        s = slice(self.myid, None, self.numprocs)
        return {"mass": mass[s], "pos": pos[s], "vel": vel[s], "indx": ids[s]}


class PSPhdf5(PSP):
    """EXP's HDF5 snapshots (exputil/ParticleReader.cc:730-1270): the files of ONE snapshot (a directory is scanned for
    names ending in a digit; their number must be Header/NumFilesPerSnapshot), components by name
    (Parameters/ComponentNames), either layout (Config/PSPstyle: 0 = Gadget-4 datasets, otherwise the compound dataset
    ``particles``).  The first file supplies the metadata; particle n of a file goes to rank n mod numprocs."""

    def __init__(self, files: Sequence[str], verbose: bool = False):
        super().__init__(verbose)
        self._files = list(files)
        self._verbose = verbose
        if len(self._files) == 1:
            scan = self.scanDirectory(self._files[0])
            if scan:
                first = min(scan)                             # partial_sort: the smallest name first
                scan.remove(first)
                self._files = [first] + scan
        f0 = self._files[0]
        try:
            self.nptot = [int(v) for v in attr_numbers(f0, "/Header", "NumPart_Total")]
            self.mass = [float(v) for v in attr_numbers(f0, "/Header", "MassTable")]
            self.nfiles = int(attr_numbers(f0, "/Header", "NumFilesPerSnapshot")[0])
            self.comps = attr_strings(f0, "/Parameters", "ComponentNames")
            self.gadget4 = int(attr_numbers(f0, "/Config", "PSPstyle")[0]) == 0
            self.ntypes = int(attr_numbers(f0, "/Config", "NTYPES")[0])
            self.Niattrib = [int(v) for v in attr_numbers(f0, "/Config", "Niattrib")]
            self.Ndattrib = [int(v) for v in attr_numbers(f0, "/Config", "Ndattrib")]
            self.real4 = int(attr_numbers(f0, "/Config", "DOUBLEPRECISION")[0]) == 0
            self.time = float(attr_numbers(f0, "/Header", "Time")[0])
        except RuntimeError as e:
            raise RuntimeError("PSPhdf5: error reading HDF5 file, " + str(e)) from e
        self.totalCount = sum(self.nptot)
        self.curcomp, self.curindx = self.comps[0], 0
        if self.nfiles != len(self._files):
            raise RuntimeError("PSPhdf5: number of files does not match number expected for this snapshot")

    def NumFiles(self) -> int:
        return self.nfiles

    def GetTypes(self) -> List[str]:
        return list(self.comps)

    def SelectType(self, name: str) -> None:
        if name not in self.comps:
            sys.stderr.write(f"PSPhdf5 error: could not find particle component <{name}>\nAvailable particle components "
                             "are: " + " ".join(self.comps) + "\n")
            raise RuntimeError("PSPhdf5: non-existent component")
        self.curcomp, self.curindx = name, self.comps.index(name)
        self._sel = None

    def CurrentNumber(self) -> int:
        return self.nptot[self.curindx]

    def CurrentTime(self) -> float:
        return self.time

    def _read_file(self, name: str) -> Optional[Dict[str, np.ndarray]]:
        k = self.curindx
        self.time = float(attr_numbers(name, "/Header", "Time")[0])
        npart = [int(v) for v in attr_numbers(name, "/Header", "NumPart_ThisFile")]
        if not npart[k] > 0:
            return None
        grp = f"/PartType{k}"
        n, ni, nd = npart[k], self.Niattrib[k], self.Ndattrib[k]
        real = "f" if self.real4 else "d"
        out: Dict[str, np.ndarray] = {}
        if self.gadget4:
            out["indx"] = dset_read(name, grp + "/ParticleIDs", "L").reshape(-1)
            out["mass"] = (np.full(n, self.mass[k]) if self.mass[k] > 0
                           else dset_read(name, grp + "/Masses", real).astype(np.float64).reshape(-1))
            out["pos"] = dset_read(name, grp + "/Coordinates", real).astype(np.float64)
            out["vel"] = dset_read(name, grp + "/Velocities", real).astype(np.float64)
            out["pot"] = dset_read(name, grp + "/Potential", real).astype(np.float64).reshape(-1)
            out["potext"] = dset_read(name, grp + "/PotentialExt", real).astype(np.float64).reshape(-1)
            if ni > 0:
                out["iattrib"] = dset_read(name, grp + "/IntAttributes", "i")
            if nd > 0:
                out["dattrib"] = dset_read(name, grp + "/RealAttributes", real).astype(np.float64)
        else:
            shape, _ = dset_shape(name, grp + "/particles")
            if shape[0] != n:
                raise RuntimeError("PSPhdf5: number of particles in file does not match the number in header")
            ids = np.zeros(n, dtype=np.int64)
            mass, pos, vel = np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3))
            pot, potext = np.zeros(n), np.zeros(n)
            ia, da = np.zeros((n, ni), dtype=np.int32), np.zeros((n, nd))
            rc = _h5().exp_h5p_particles_read(name.encode(), (grp + "/particles").encode(), int(self.real4),
                                              ctypes.c_longlong(n), ni, nd, _vp(ids), _vp(mass), _vp(pos), _vp(vel),
                                              _vp(pot), _vp(potext), _vp(ia), _vp(da))
            if rc != 0:
                raise RuntimeError(f"PSPpsp: error reading HDF5 file, {name} ({rc})")
            out = {"indx": ids.astype(np.uint64), "mass": mass, "pos": pos, "vel": vel, "pot": pot, "potext": potext}
            if ni > 0:
                out["iattrib"] = ia
            if nd > 0:
                out["dattrib"] = da
        s = slice(self.myid, None, self.numprocs)
        return {key: v[s] for key, v in out.items()}

    def _load(self) -> Dict[str, np.ndarray]:
        parts = [p for p in (self._read_file(f) for f in self._files) if p is not None]
        if not parts:
            k = self.curindx
            return {"mass": np.zeros(0), "pos": np.zeros((0, 3)), "vel": np.zeros((0, 3)), "indx": np.zeros(0, np.uint64),
                    "pot": np.zeros(0), "potext": np.zeros(0),
                    **({"iattrib": np.zeros((0, self.Niattrib[k]), np.int32)} if self.Niattrib[k] else {}),
                    **({"dattrib": np.zeros((0, self.Ndattrib[k]))} if self.Ndattrib[k] else {})}
        return {key: np.concatenate([p[key] for p in parts]) for key in parts[0]}

    def PrintSummary(self, stats: bool = True, timeonly: bool = False, out=None) -> None:
        ParticleReader.PrintSummary(self, stats, timeonly, out)

    def writePSP(self, out, real4: bool) -> None:
        raise RuntimeError("PSPhdf5: writePSP works on the stanzas of a binary PSP file")


def write_psp_hdf5(path: str, time: float, comps: Sequence[dict], real4: bool = False, gadget4: bool = False,
                   nfiles: int = 1, totals: Optional[Sequence[int]] = None, ids: bool = True, chunk: int = 0,
                   compress: int = 0, shuffle: bool = True, expconfig: Optional[str] = None, all_meta: bool = True,
                   version=("", "", "")) -> None:
    """One file of an OutHDF5 snapshot (src/OutHDF5.cc:400-560 for ``gadget4``, :645-780 for the PSP style, the
    default).  ``comps``: dicts with ``name``, ``force`` (the method's id), ``fconf`` (its YAML), ``mass``, ``pos``,
    ``vel``, ``pot``, ``potext``, ``indx``, optionally ``iattrib`` / ``dattrib``; ``nfiles`` / ``totals`` describe the
    whole snapshot when this is one part of several (NumFilesPerSnapshot, NumPart_Total).

    One deliberate difference in the Gadget-4 layout: the reference enters the common mass in MassTable and passes
    ``multim`` = "all masses equal" as the flag that WRITES the Masses dataset (checkParticleMasses :783-826, :553-556), so
    a component with unequal masses gets MassTable 0 and no Masses -- which its own reader then fails to find.  Here
    Masses is written exactly when MassTable holds 0."""
    lib = _h5()
    p = path.encode()
    if lib.exp_h5p_create(p) != 0:
        raise RuntimeError(f"OutHDF5: can't open file <{path}>")

    def attr(obj, name, kind, val, scalar=False):
        a = np.ascontiguousarray(val, dtype=_KIND[kind]).reshape(-1)
        rc = lib.exp_h5p_attr_write(p, obj.encode(), name.encode(), ctypes.c_char(kind.encode()), -1 if scalar else len(a), _vp(a))
        if rc != 0:
            raise RuntimeError(f"OutHDF5: error writing HDF5 file, attribute {name} ({rc})")

    def sattr(obj, name, vals, scalar=False):
        arr = (ctypes.c_char_p * max(len(vals), 1))(*[v.encode() for v in vals])
        rc = lib.exp_h5p_attr_write_str(p, obj.encode(), name.encode(), -1 if scalar else len(vals), arr)
        if rc != 0:
            raise RuntimeError(f"OutHDF5: error writing HDF5 file, attribute {name} ({rc})")

    def dset(name, kind, a):
        a = np.ascontiguousarray(a, dtype=_KIND[kind])
        dims = (ctypes.c_longlong * 4)(*(list(a.shape) + [1] * (4 - a.ndim)))
        rc = lib.exp_h5p_dset_write(p, name.encode(), ctypes.c_char(kind.encode()), a.ndim, dims, _vp(a), int(chunk),
                                    int(compress), int(bool(shuffle)))
        if rc != 0:
            raise RuntimeError(f"OutHDF5: error writing HDF5 file, dataset {name} ({rc})")

    masses = []
    for c in comps:                                           # checkParticleMasses (:783-826)
        m = np.asarray(c["mass"], dtype=np.float64)
        lo, hi = (m.min(), m.max()) if len(m) else (np.finfo(np.float64).max, 0.0)
        masses.append(float(hi) if hi > 0 and (hi - lo) / hi < 1.0e-12 else 0.0)
    nums = [len(c["mass"]) for c in comps]
    for g in ("/Header",):
        lib.exp_h5p_group(p, g.encode())
    attr("/Header", "MassTable", "d", masses)
    attr("/Header", "NumPart_ThisFile", "L", nums)
    attr("/Header", "Time", "d", [time], scalar=True)
    dp = 0 if real4 else 1
    if all_meta:
        # (the Gadget-4 branch writes Flag_DoublePrecision = 1 whatever real4 says, :440-441; DOUBLEPRECISION, which is
        # what the reader goes by, follows it -- so a real4 Gadget-4 snapshot of the reference declares doubles and
        # holds floats, which HDF5 converts on reading.  The flag written here states what the file holds.)
        attr("/Header", "Flag_DoublePrecision", "i", [dp], scalar=True)
        for name, v in (("HubbleParam", 1.0), ("Omega0", 0.0), ("OmegaBaryon", 0.0), ("OmegaLambda", 0.0), ("Redshift", 0.0)):
            attr("/Header", name, "d", [v], scalar=True)
        attr("/Header", "NumFilesPerSnapshot", "i", [nfiles], scalar=True)
        attr("/Header", "NumPart_Total", "L", list(totals) if totals is not None else nums)
        lib.exp_h5p_group(p, b"/Config")
        attr("/Config", "PSPstyle", "i", [0 if gadget4 else 1], scalar=True)
        attr("/Config", "NTYPES", "i", [len(comps)], scalar=True)
        attr("/Config", "DOUBLEPRECISION", "i", [dp], scalar=True)
        nia = [0 if _two_d(c.get("iattrib"), len(c["mass"]), np.int32) is None else _two_d(c["iattrib"], len(c["mass"]), np.int32).shape[1] for c in comps]
        nda = [0 if _two_d(c.get("dattrib"), len(c["mass"]), np.float64) is None else _two_d(c["dattrib"], len(c["mass"]), np.float64).shape[1] for c in comps]
        attr("/Config", "Niattrib", "i", nia)
        attr("/Config", "Ndattrib", "i", nda)
        lib.exp_h5p_group(p, b"/Parameters")
        for name, v in zip(("Git_commit", "Git_branch", "Compile_date"), version):
            sattr("/Parameters", name, [v], scalar=True)
        sattr("/Parameters", "ComponentNames", [c["name"] for c in comps])
        sattr("/Parameters", "ForceMethods", [c.get("force", "") for c in comps])
        sattr("/Parameters", "ForceConfigurations", [c.get("fconf", "") for c in comps])
        if expconfig is not None:
            sattr("/Parameters", "EXPConfiguration", [expconfig], scalar=True)
    real = "f" if real4 else "d"
    for k, c in enumerate(comps):
        grp = f"/PartType{k}"
        lib.exp_h5p_group(p, grp.encode())
        n = nums[k]
        zeros = np.zeros(n)
        pot = np.asarray(c["pot"]) if c.get("pot") is not None else zeros
        pex = np.asarray(c["potext"]) if c.get("potext") is not None else zeros
        vel = np.asarray(c["vel"]) if c.get("vel") is not None else np.zeros((n, 3))
        idx = np.asarray(c["indx"]) if c.get("indx") is not None else np.arange(1, n + 1)
        ia, da = _two_d(c.get("iattrib"), n, np.int32), _two_d(c.get("dattrib"), n, np.float64)
        ia = None if ia is None else np.ascontiguousarray(ia)
        da = None if da is None else np.ascontiguousarray(da)
        if gadget4:
            if masses[k] == 0.0:
                dset(grp + "/Masses", real, c["mass"])
            if ids:
                dset(grp + "/ParticleIDs", "l", idx.astype(np.int64))
            dset(grp + "/Coordinates", real, np.asarray(c["pos"]).reshape(n, 3))
            dset(grp + "/Velocities", real, vel.reshape(n, 3))
            dset(grp + "/Potential", real, pot)
            dset(grp + "/PotentialExt", real, pex)
            if ia is not None and ia.shape[1]:
                dset(grp + "/IntAttributes", "i", ia)
            if da is not None and da.shape[1]:
                dset(grp + "/RealAttributes", real, da)
        else:
            f64 = lambda a, shape: np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(shape))
            rc = lib.exp_h5p_particles_write(p, (grp + "/particles").encode(), int(bool(real4)), ctypes.c_longlong(n),
                                             0 if ia is None else ia.shape[1], 0 if da is None else da.shape[1],
                                             _vp(np.ascontiguousarray(idx, dtype=np.int64)), _vp(f64(c["mass"], n)),
                                             _vp(f64(c["pos"], (n, 3))), _vp(f64(vel, (n, 3))), _vp(f64(pot, n)),
                                             _vp(f64(pex, n)), _vp(ia), _vp(da), int(chunk), int(compress),
                                             int(bool(shuffle)))
            if rc != 0:
                raise RuntimeError(f"OutHDF5: error writing HDF5 file, {grp}/particles ({rc})")
