"""``pyEXP.read``: the phase-space files either side of the path (include/ParticleReader.H, exputil/ParticleReader.cc;
pyEXP/ParticleReaderWrappers.cc) -- what ``Basis.createFromReader`` and the particle histograms of ``FieldGenerator``
are handed, and what the n-body side writes and restarts from.

Readers (``ParticleReader.createReader(type, files)``):

* ``PSPout``       EXP's monolithic phase-space dump ``OUT.runtag.nnnnn`` (src/OutPSN.cc:130-170, src/Component.cc:2385-2454,
                   exputil/Particle.cc:333-388): MasterHeader {double time; int ntot; int ncomp}, then per component an
                   unsigned long magic 0xadbfabc0 + sizeof(real), ComponentHeader {int nbod, niatr, ndatr, ninfochar;
                   char info[ninfochar]} (exputil/header.cc:7-19) whose info string is the component's YAML stanza, and nbod
                   packed records [unsigned long indx]? real mass, pos[3], vel[3], pot; int iattrib[niatr]; real
                   dattrib[ndatr];
* ``PSPspl``       the split dump ``SPL.runtag.nnnnn`` (src/OutPSQ.cc): the same master file with, per component, the
                   number of part files and their 1024-byte names after the magic; each part starts with its unsigned
                   particle count;
* ``GadgetNative`` Gadget-2 binary snapshots (include/gadget.H: 256-byte header between Fortran record marks; float
                   positions, velocities, int ids, float masses where the mass table holds 0);
* ``TipsyNative``, ``Bonsai``, ``Bonsai1``  Tipsy native files (include/tipsy.H: 32-byte header, gas / dark / star structs of
                   floats; Bonsai keeps a 64-bit id in (eps, phi), Bonsai1 a 32-bit one in phi);
* ``GadgetHDF5``, ``PSPhdf5``  through the HDF5 shim (``exp_amd/libexp_amd_h5.so``), see ``reader_h5.py``.

The reference hands out one ``Particle`` at a time; ``firstParticle`` / ``nextParticle`` do the same here, and
``arrays()`` returns the whole selection of this rank at once (what the device path consumes).  With
``torch.distributed`` initialised the particles are dealt over the ranks as the reference deals them over MPI ranks:
PSP and Gadget files round-robin (particle n goes to rank n mod numprocs), Tipsy files in contiguous blocks.

Writers: ``write_psp`` (the OUT file of OutPSN / OutCHKPT), ``write_spl`` (OutPSQ), ``write_bodies_ascii`` /
``read_bodies_ascii`` (the body file a component starts from: src/Component.cc:1462-1560, exputil/Particle.cc:469-526)."""
from __future__ import annotations

import ctypes
import io
import math
import os
import struct
import sys
from typing import Dict, List, Optional, Sequence

import numpy as np

PSP_MAGIC = 0xadbfabc0          # include/ParticleReader.H:340-342, src/Component.H (magic), src/Component.cc:2426
PSP_MMASK = 0xf
PSP_NMASK = ~PSP_MMASK & 0xffffffffffffffff
DEFAULT_INFO_SIZE = 1024        # ComponentHeader::defaultInfoSize (exputil/header.cc:5)
SPL_NAME_SIZE = 1024            # PBUF_SIZ (exputil/ParticleReader.cc:1587)


class Particle:
    """include/Particle.H: the fields a reader fills."""
    __slots__ = ("mass", "pos", "vel", "acc", "pot", "potext", "iattrib", "dattrib", "level", "indx")

    def __init__(self):
        self.mass = 0.0
        self.pos = np.zeros(3)
        self.vel = np.zeros(3)
        self.acc = np.zeros(3)
        self.pot = self.potext = 0.0
        self.iattrib: List[int] = []
        self.dattrib: List[float] = []
        self.level = 0
        self.indx = 0


def _ranks():
    """(numprocs, myid): ParticleReader() asks MPI (include/ParticleReader.H:50-62); here the process group."""
    dist = sys.modules.get("torch.distributed")               # (a process with a group has imported it; no import here:
    try:                                                       # the first `import torch` of a process takes seconds)
        if dist is not None and dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
    except Exception:
        pass
    return 1, 0


_hostlib = False


def _host_lib():
    """libexp_amd.so for its host-side record unpacking (exp_amd_host_psp_unpack); None when it is not built -- the
    files are then unpacked by numpy, same values, a tenth of the rate"""
    global _hostlib
    if _hostlib is False:
        try:
            from ._lib import load
            _hostlib = load()
            if not hasattr(_hostlib, "exp_amd_host_psp_unpack"):
                _hostlib = None
        except Exception:
            _hostlib = None
    return _hostlib


class P2Quantile:
    """include/P2Quantile.H: the P-square streaming quantile estimator (Jain & Chlamtac 1985) behind the medians of
    the generic ``PrintSummary``."""

    def __init__(self, prob: float = 0.5):
        self.p, self.count = prob, 0
        self.q = [0.0] * 5
        self.n = [0] * 5
        self.ns = [0.0] * 5
        self.dns = [0.0] * 5

    def addValue(self, x: float) -> None:
        q, n, ns, dns, p = self.q, self.n, self.ns, self.dns, self.p
        if self.count < 5:
            q[self.count] = x
            self.count += 1
            if self.count == 5:
                q.sort()
                for i in range(5):
                    n[i] = i
                ns[:] = [0, 2 * p, 4 * p, 2 + 2 * p, 4]
                dns[:] = [0, p / 2, p, (1 + p) / 2, 1]
            return
        if x < q[0]:
            q[0] = x; k = 0
        elif x < q[1]: k = 0
        elif x < q[2]: k = 1
        elif x < q[3]: k = 2
        elif x < q[4]: k = 3
        else:
            q[4] = x; k = 3
        for i in range(k + 1, 5):
            n[i] += 1
        for i in range(5):
            ns[i] += dns[i]
        for i in range(1, 4):
            d = ns[i] - n[i]
            if (d >= 1 and n[i + 1] - n[i] > 1) or (d <= -1 and n[i - 1] - n[i] < -1):
                di = -1 if d < 0 else 1
                qs = q[i] + di / (n[i + 1] - n[i - 1]) * (
                    (n[i] - n[i - 1] + di) * (q[i + 1] - q[i]) / (n[i + 1] - n[i]) +
                    (n[i + 1] - n[i] - di) * (q[i] - q[i - 1]) / (n[i] - n[i - 1]))
                if q[i - 1] < qs < q[i + 1]:
                    q[i] = qs
                else:
                    q[i] = q[i] + di * (q[i + di] - q[i]) / (n[i + di] - n[i])
                n[i] += di
        self.count += 1

    def getQuantile(self) -> float:
        if self.count == 0:
            raise RuntimeError("Sequence contains no elements")
        if self.count <= 5:
            s = sorted(self.q[: self.count])
            self.q[: self.count] = s
            # std::round: half away from zero
            return s[int(math.floor((self.count - 1) * self.p + 0.5))]
        return self.q[2]


def _g(v: float) -> str:
    """operator<< of a double with the stream's default flags"""
    return "%g" % float(v)


class ParticleReader:
    """Base class (include/ParticleReader.H:30-101)."""

    readerTypes = ["PSPout", "PSPspl", "GadgetNative", "GadgetHDF5", "PSPhdf5",
                   "TipsyNative", "TipsyXDR", "Bonsai1", "Bonsai"]          # exputil/ParticleReader.cc:1932-1934

    def __init__(self):
        self.numprocs, self.myid = _ranks()
        self._sel: Optional[Dict[str, np.ndarray]] = None     # the loaded selection of this rank
        self._it = 0

    # -- the interface of the reference ----------------------------------------------------------------------
    def SelectType(self, name: str) -> None:
        raise NotImplementedError

    def CurrentNumber(self) -> int:
        raise NotImplementedError

    def GetTypes(self) -> List[str]:
        raise NotImplementedError

    def CurrentTime(self) -> float:
        raise NotImplementedError

    def _load(self) -> Dict[str, np.ndarray]:
        """mass [n], pos [n, 3], vel [n, 3], indx [n] (+ pot, iattrib, dattrib where the format has them) of the
        selected type, this rank's share, in file order"""
        raise NotImplementedError

    def arrays(self) -> Dict[str, np.ndarray]:
        if self._sel is None:
            self._sel = self._load()
        return self._sel

    def _particle(self, i: int) -> Particle:
        a = self._sel
        p = Particle()
        p.mass = float(a["mass"][i])
        p.pos = a["pos"][i].astype(np.float64)
        p.vel = a["vel"][i].astype(np.float64)
        p.indx = int(a["indx"][i])
        if "pot" in a:
            p.pot = float(a["pot"][i])
        if "potext" in a:
            p.potext = float(a["potext"][i])
        if "iattrib" in a:
            p.iattrib = [int(v) for v in a["iattrib"][i]]
        if "dattrib" in a:
            p.dattrib = [float(v) for v in a["dattrib"][i]]
        return p

    def firstParticle(self) -> Optional[Particle]:
        self.arrays()
        self._it = 0
        return self.nextParticle()

    def nextParticle(self) -> Optional[Particle]:
        a = self.arrays()
        if self._it >= len(a["mass"]):
            return None
        self._it += 1
        return self._particle(self._it - 1)

    def __iter__(self):
        p = self.firstParticle()
        while p is not None:
            yield p
            p = self.nextParticle()

    def PrintSummary(self, stats: bool = True, timeonly: bool = False, out=None) -> None:
        """The generic summary (exputil/ParticleReader.cc:2298-2407); ``stats`` is ignored there too."""
        out = out or sys.stdout
        out.write(f"   Time                : {_g(self.CurrentTime())}\n")
        if timeonly:
            return
        out.write(f"   Number of particles : {self.CurrentNumber()}\n")
        a = self.arrays()
        m, pos, vel = a["mass"].astype(np.float64), a["pos"].astype(np.float64), a["vel"].astype(np.float64)
        mtot = m.sum()
        rows = {}
        if len(m) == 0:
            return
        for tag, x in (("p", pos), ("v", vel)):
            med = []
            for k in range(3):
                est = P2Quantile()
                for v in x[:, k]:
                    est.addValue(float(v))
                med.append(est.getQuantile())
            avg = (m[:, None] * x).sum(axis=0) / mtot
            var = (m[:, None] * x * x).sum(axis=0) / mtot
            rows[tag] = (x.min(axis=0), med, avg, np.sqrt(np.abs(var - avg * avg)), x.max(axis=0))

        def block(title, labels, r):
            out.write("\n" + title.rjust(20) + "".join(s.rjust(15) for s in labels) + "\n")
            for name, vals in zip(("Min :: ", "Med :: ", "Avg :: ", "Std :: ", "Max :: "), r):
                out.write(name.rjust(20) + "".join(_g(v).rjust(15) for v in vals) + "\n")
        block("*** Position", "XYZ", rows["p"])
        block("*** Velocity", "UVW", rows["v"])

    # -- file lists (exputil/ParticleReader.cc:1937-2034) ------------------------------------------------------
    @staticmethod
    def parseFileList(file: str, delimit: str):
        files: List[str] = []
        try:
            with open(file) as f:
                files = f.read().split()
        except OSError:
            sys.stderr.write(f"Error opening file <{file}>\n")
        return ParticleReader.parseStringList(files, delimit)

    @staticmethod
    def _all_directories(files: Sequence[str]) -> bool:
        d = sum(1 for f in files if os.path.isdir(f))
        if d > 0 and d < len(files):
            raise RuntimeError("ParticleReader::parseDirectoryList: cannot mix directories and files")
        return d > 0

    @staticmethod
    def parseStringList(infiles: Sequence[str], delimit: str):
        """Batches of files that share everything before the LAST occurrence of any character of ``delimit``
        (``find_last_of``): the parts of one snapshot.  The list is sorted first; directories are one batch each."""
        files = sorted(infiles)
        if ParticleReader._all_directories(files):
            return [[d] for d in files]
        batches, batch, templ = [], [], ""
        for f in files:
            found = max((f.rfind(ch) for ch in delimit), default=-1)
            if found < 0:
                batch.append(f)
                batches.append(batch)
                batch = []
                continue
            trimmed = f[:found]
            if not batch:
                templ = trimmed
                batch.append(f)
            elif trimmed == templ:
                batch.append(f)
            else:
                batches.append(batch)
                batch = [f]
                templ = trimmed
        if batch:
            batches.append(batch)
        return batches

    @staticmethod
    def scanDirectory(d: str) -> List[str]:
        """regular files of a directory whose name ends in a digit (the parts of a split snapshot, :691-728)"""
        if not os.path.isdir(d):
            return []
        return [os.path.join(d, e) for e in os.listdir(d)
                if os.path.isfile(os.path.join(d, e)) and e[-1:].isdigit()]

    @staticmethod
    def getReaders() -> List[str]:
        return list(ParticleReader.readerTypes)

    @staticmethod
    def createReader(reader: str, file, myid: int = 0, verbose: bool = False) -> "ParticleReader":
        """exputil/ParticleReader.cc:2036-2089: the type is matched as a PREFIX, in this order."""
        files = [file] if isinstance(file, str) else list(file)
        if reader.startswith("PSPout"):
            return PSPout(files, verbose)
        if reader.startswith("PSPspl"):
            return PSPspl(files, verbose)
        if reader.startswith("PSPhdf5"):
            from .reader_h5 import PSPhdf5
            return PSPhdf5(files, verbose)
        if reader.startswith("GadgetNative"):
            return GadgetNative(files, verbose)
        if reader.startswith("GadgetHDF5"):
            from .reader_h5 import GadgetHDF5
            return GadgetHDF5(files, verbose)
        if reader.startswith("TipsyNative"):
            return Tipsy(files, "native", verbose)
        if reader.startswith("TipsyXDR"):
            # a reference build without RPC/XDR prints this and exits (:2059-2069)
            raise RuntimeError("ParticleReader: this build does not have RPC/XDR support so Tipsy standard reading "
                               "is not available.  Use Tipsy native format.")
        if reader.startswith("Bonsai1"):
            return Tipsy(files, "bonsai1", verbose)
        if reader.startswith("Bonsai"):
            return Tipsy(files, "bonsai", verbose)
        raise RuntimeError(f"ParticleReader: I don't know about reader <{reader}>\nAvailable readers are: "
                           + " ".join(ParticleReader.readerTypes))


# ---------------------------------------------------------------------------------------------------------------
# PSP
# ---------------------------------------------------------------------------------------------------------------
class PSPstanza:
    """include/ParticleReader.H:238-259"""

    def __init__(self):
        self.nbod = self.niatr = self.ndatr = self.ninfochar = 0
        self.info = b""
        self.name = self.id = self.cparam = self.fparam = ""
        self.index_size = 0
        self.r_size = 8
        self.pspos = 0
        self.nparts: List[str] = []


def psp_record_dtype(r_size: int, indexed: bool, niatr: int, ndatr: int) -> np.dtype:
    """One particle of a PSP file, packed (exputil/Particle.cc:333-388; PParticle::read, include/ParticleReader.H:276-315)."""
    real = "<f4" if r_size == 4 else "<f8"
    f = []
    if indexed:
        f.append(("indx", "<u8"))
    f += [("mass", real), ("pos", real, (3,)), ("vel", real, (3,)), ("pot", real)]
    if niatr:
        f.append(("iattrib", "<i4", (niatr,)))
    if ndatr:
        f.append(("dattrib", real, (ndatr,)))
    return np.dtype(f, align=False)


def _flow(node) -> str:
    """a YAML node in flow style, as `out << node` prints one whose style was set to Flow"""
    import yaml
    if node is None:
        return "~"
    return yaml.dump(node, default_flow_style=True, width=1 << 20).strip()


def _string_tok(text: str, delim: str, count: int) -> List[str]:
    """``StringTok<string> tokens(text); trim_copy(tokens(delim))`` ``count`` times (include/StringTok.H,
    exputil/Sutils.cc): like strtok, a token starts at the first character that is NOT a delimiter, so empty fields are
    skipped, and once the text is used up every further token is empty"""
    out, pos = [], 0
    for _ in range(count):
        token = ""
        if pos is not None:
            first = next((i for i in range(pos, len(text)) if text[i] not in delim), None)
            if first is not None:
                end = next((i for i in range(first, len(text)) if text[i] in delim), None)
                token = text[first:end] if end is not None else text[first:]
                pos = None if end is None or end + 1 >= len(text) else end + 1
        out.append(token.strip(" \t\n\r\f\v"))
    return out


def _parse_info(st: PSPstanza, allow_old: bool) -> Optional[str]:
    """The stanza's info string -> name, id, cparam, fparam, index_size (exputil/ParticleReader.cc:1346-1439)."""
    import yaml
    text = st.info.split(b"\0", 1)[0].decode("utf-8", "replace")
    err = None
    conf = None
    try:
        conf = yaml.safe_load(io.StringIO(text))
        if not isinstance(conf, dict):
            raise yaml.YAMLError("the info string is not a map")
    except yaml.YAMLError as e:
        if not allow_old:
            raise RuntimeError("Error parsing component config") from e
        err = f"Error parsing component config.  Trying old-style PSP\n{e}\n"
        conf = None
    if conf is not None:
        cconf, fconf = conf.get("parameters"), conf.get("force")
        if "name" not in conf:
            raise RuntimeError("Error parsing component config: the stanza has no <name>")
        st.name = str(conf["name"])
        st.id = str(fconf["id"]) if isinstance(fconf, dict) else "<undefined>"
        st.cparam = _flow(cconf)
        st.fparam = _flow(fconf.get("parameters")) if isinstance(fconf, dict) else "<undefined>"
        st.index_size = 8 if isinstance(cconf, dict) and bool(cconf.get("indexing", False)) else 0
        return err
    # old style: name : id : cparam : fparam, "indexing=1" in cparam (:1405-1437)
    tok = _string_tok(text, ":", 4)
    st.name, st.id, st.cparam, st.fparam = tok[0], tok[1], tok[2], tok[3]
    st.index_size = 0
    p1 = st.cparam.find("indexing")
    if p1 >= 0:
        p2 = st.cparam.find("=", p1)
        if p2 < 0:
            raise RuntimeError("Bad syntax in component parameter string")
        p3 = st.cparam.find(",", p2)
        val = st.cparam[p2 + 1:] if p3 < 0 else st.cparam[p2 + 1:p3]
        try:
            on = int(val.strip().split()[0]) if val.strip() else 0          # atoi
        except ValueError:
            on = 0
        if on:
            st.index_size = 8
    return err


def _read_component_header(f, st: PSPstanza) -> None:
    """ComponentHeader::read (exputil/header.cc:85-110)"""
    raw = f.read(16)
    if len(raw) < 16:
        raise RuntimeError("Error reading component header")
    st.nbod, st.niatr, st.ndatr, st.ninfochar = struct.unpack("<4i", raw)
    st.info = f.read(st.ninfochar)
    if len(st.info) < st.ninfochar:
        raise RuntimeError("Error reading component header")


class PSP(ParticleReader):
    """include/ParticleReader.H:333-432"""

    def __init__(self, verbose: bool = False):
        super().__init__()
        self.VERBOSE = verbose
        self.time, self.ntot, self.ncomp = 0.0, 0, 0
        self.stanzas: List[PSPstanza] = []
        self._cur = 0

    def CurrentTime(self) -> float:
        return self.time

    def GetTypes(self) -> List[str]:
        return [s.name for s in self.stanzas]

    def GetNamed(self, name: str) -> Optional[PSPstanza]:
        for i, s in enumerate(self.stanzas):
            if s.name == name:
                self._cur, self._sel = i, None
                return s
        return None

    def SelectType(self, name: str) -> None:
        if self.GetNamed(name) is None:
            print(f"PSP error: no particle type <{name}>")
            raise RuntimeError("PSP error: non-existent particle type")

    def CurrentNumber(self) -> int:
        return self.stanzas[self._cur].nbod

    def GetStanza(self) -> Optional[PSPstanza]:
        self._cur, self._sel = 0, None
        return self.stanzas[0] if self.stanzas else None

    def NextStanza(self) -> Optional[PSPstanza]:
        self._cur += 1
        self._sel = None
        return self.stanzas[self._cur] if self._cur < len(self.stanzas) else None

    def _dtype(self, st: PSPstanza) -> np.dtype:
        return psp_record_dtype(st.r_size, st.index_size > 0, st.niatr, st.ndatr)

    def _finish(self, st: PSPstanza, rec: np.ndarray, first: int) -> Dict[str, np.ndarray]:
        """records of this rank (global sequence numbers first, first + numprocs, ...) -> double arrays.  The fields are
        taken out of the packed records one scalar column at a time (a strided 1-D copy each): numpy's copy of a
        sub-array field out of a packed structured array runs at a fifth of that rate."""
        n = len(rec)
        seq = first + self.numprocs * np.arange(n, dtype=np.uint64)
        lib = _host_lib()
        if lib is not None and n:
            # rec may be a strided view of the file's records (this rank's share): the byte distance between them is the stride
            base = rec if rec.flags.c_contiguous or rec.strides[0] % rec.dtype.itemsize == 0 else np.ascontiguousarray(rec)
            indx = np.empty(n, np.uint64) if st.index_size else seq
            mass, pos, vel, pot = np.empty(n), np.empty((n, 3)), np.empty((n, 3)), np.empty(n)
            ia = np.empty((n, st.niatr), np.int32) if st.niatr else None
            da = np.empty((n, st.ndatr)) if st.ndatr else None
            vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
            rc = lib.exp_amd_host_psp_unpack(n, ctypes.c_void_p(base.__array_interface__["data"][0]), int(base.strides[0]),
                                             int(st.r_size), int(bool(st.index_size)), int(st.niatr), int(st.ndatr),
                                             vp(indx) if st.index_size else None, vp(mass), vp(pos), vp(vel), vp(pot), vp(ia), vp(da))
            if rc != 0:
                raise RuntimeError(f"PSP: exp_amd_host_psp_unpack failed ({rc})")
            out = {"mass": mass, "pos": pos, "vel": vel, "pot": pot, "indx": indx}
            if ia is not None:
                out["iattrib"] = ia
            if da is not None:
                out["dattrib"] = da
            return out
        rec = np.ascontiguousarray(rec)
        dt, raw, size = rec.dtype, rec.view(np.uint8).reshape(-1), rec.dtype.itemsize

        def column(name, k, width, out_dtype):
            base, off = dt.fields[name][0].base, dt.fields[name][1]
            out = np.empty((n, width) if width > 1 or k is None else n, dtype=out_dtype)
            for j in range(width):
                col = np.ndarray(shape=(n,), dtype=base, buffer=raw, offset=off + j * base.itemsize, strides=(size,))
                if out.ndim == 2:
                    out[:, j] = col
                else:
                    out[:] = col
            return out
        if n == 0:
            out = {"mass": np.zeros(0), "pos": np.zeros((0, 3)), "vel": np.zeros((0, 3)), "pot": np.zeros(0), "indx": seq}
            if st.niatr:
                out["iattrib"] = np.zeros((0, st.niatr), np.int32)
            if st.ndatr:
                out["dattrib"] = np.zeros((0, st.ndatr))
            return out
        out = {"mass": column("mass", 0, 1, np.float64), "pos": column("pos", None, 3, np.float64),
               "vel": column("vel", None, 3, np.float64), "pot": column("pot", 0, 1, np.float64),
               "indx": column("indx", 0, 1, np.uint64) if st.index_size else seq}      # indx = pcount otherwise (:283)
        if st.niatr:
            out["iattrib"] = column("iattrib", None, st.niatr, np.int32)
        if st.ndatr:
            out["dattrib"] = column("dattrib", None, st.ndatr, np.float64)
        return out

    def PrintSummary(self, stats: bool = True, timeonly: bool = False, out=None) -> None:
        """PSP::PrintSummary (exputil/ParticleReader.cc:1605-1658).  As there, the statistics printed under EVERY
        stanza are those of the stanza currently selected (ComputeStats works on ``spos``)."""
        out = out or sys.stdout
        out.write(f"Time={_g(self.time)}\n")
        if timeonly:
            return
        out.write(f"   Total particle number: {self.ntot}\n   Number of components:  {self.ncomp}\n")
        for cnt, s in enumerate(self.stanzas, 1):
            out.write("-" * 60 + "\n" + f"--- Component #{cnt:2d}\n")
            for key, val in (("name", s.name), ("id", s.id), ("cparam", s.cparam), ("fparam", s.fparam),
                             ("nbod", s.nbod), ("niatr", s.niatr), ("ndatr", s.ndatr), ("rsize", s.r_size)):
                out.write(f" {key} :: ".rjust(20) + f"{val}\n")
            out.write("-" * 60 + "\n")
            if stats and len(self.arrays()["mass"]):
                a = self.arrays()
                n = self.stanzas[self._cur].nbod
                mid = min(int(math.floor(0.5 * n + 0.5)), len(a["mass"]) - 1)   # (the reference indexes one past the end for n = 1)
                for title, labels, x in (("*** Position", ("X", "Y", "Z"), a["pos"]),
                                         ("*** Velocity", ("U", "Vn", "W"), a["vel"])):
                    srt = np.sort(x, axis=0)
                    out.write("\n" + title.rjust(20) + "".join(t.rjust(15) for t in labels) + "\n")
                    for name, row in (("Min :: ", srt[0]), ("Med :: ", srt[mid]), ("Max :: ", srt[-1])):
                        out.write(name.rjust(20) + "".join(_g(v).rjust(15) for v in row) + "\n")

    def writePSP(self, out, real4: bool) -> None:
        """PSP::writePSP (exputil/ParticleReader.cc:1883-1930): a copy of the snapshot.  The magic says ``real4`` or
        not, the records are ALWAYS written as floats (``part->writeBinary(sizeof(float), ...)``, :1921) -- a copy made
        with real4 = false is unreadable; reproduced, because it is what the reference writes."""
        out.write(struct.pack("<dii", self.time, self.ntot, self.ncomp))
        keep = self._cur
        for i, st in enumerate(self.stanzas):
            self._cur, self._sel = i, None
            out.write(struct.pack("<Q", PSP_MAGIC + (4 if real4 else 8)))
            out.write(struct.pack("<4i", st.nbod, st.niatr, st.ndatr, st.ninfochar) + st.info)
            a = self.arrays()
            _write_records(out, 4, st.index_size > 0, a["indx"], a["mass"], a["pos"], a["vel"], a["pot"],
                           a.get("iattrib"), a.get("dattrib"))
        self._cur, self._sel = keep, None


class PSPout(PSP):
    """A monolithic PSP file (exputil/ParticleReader.cc:1298-1469, :1680-1735)."""

    def __init__(self, infile: Sequence[str], verbose: bool = False):
        super().__init__(verbose)
        self.file = infile[0]
        try:
            f = open(self.file, "rb")
        except OSError as e:
            raise RuntimeError(f"Could not open PSP file <{self.file}>") from e
        with f:
            raw = f.read(16)
            if len(raw) < 16:
                raise RuntimeError(f"Could not read master header for <{self.file}>")
            self.time, self.ntot, self.ncomp = struct.unpack("<dii", raw)
            for _ in range(self.ncomp):
                st = PSPstanza()
                raw = f.read(8)
                if len(raw) < 8:
                    raise RuntimeError(f"Error reading magic for <{self.file}>")
                ret, = struct.unpack("<Q", raw)
                st.r_size = (ret & PSP_MMASK) if (ret & PSP_NMASK) == PSP_MAGIC else 8
                _read_component_header(f, st)
                st.pspos = f.tell()
                msg = _parse_info(st, allow_old=True)
                if msg and verbose:
                    print(msg, end="")
                # (seeking past the end of a truncated file does not fail, there or here: the stanza is kept, the NEXT read
                # -- of a magic number here, of the particles in _load -- is what fails)
                f.seek(st.nbod * (st.index_size + 8 * st.r_size + st.niatr * 4 + st.ndatr * st.r_size), os.SEEK_CUR)
                self.stanzas.append(st)

    def _load(self) -> Dict[str, np.ndarray]:
        st = self.stanzas[self._cur]
        dt = self._dtype(st)
        try:
            rec = np.memmap(self.file, dtype=dt, mode="r", offset=st.pspos, shape=(st.nbod,)) if st.nbod else np.zeros(0, dt)
        except (ValueError, OSError) as e:                     # shorter than its header says
            raise RuntimeError(f"PSPout: <{self.file}> ends inside component <{st.name}>") from e
        return self._finish(st, rec[self.myid::self.numprocs], self.myid)   # stagger by myid, stride numprocs (:1689-1735)


class PSPspl(PSP):
    """A split PSP file (exputil/ParticleReader.cc:1471-1603, :1737-1838): ``files[0]`` is the master."""

    def __init__(self, master: Sequence[str], verbose: bool = False):
        super().__init__(verbose)
        self.file = master[0]
        try:
            f = open(self.file, "rb")
        except OSError as e:
            raise RuntimeError(f"Error opening master SPL file <{self.file}>") from e
        with f:
            raw = f.read(16)
            if len(raw) < 16:
                raise RuntimeError(f"Could not read master header for <{self.file}>")
            self.time, self.ntot, self.ncomp = struct.unpack("<dii", raw)
            for i in range(self.ncomp):
                st = PSPstanza()
                raw = f.read(12)
                if len(raw) < 12:
                    raise RuntimeError(f"Error reading magic info for Comp #{i} from <{self.file}>")
                cmagic, number = struct.unpack("<Qi", raw)
                st.r_size = (cmagic & PSP_MMASK) if (cmagic & PSP_NMASK) == PSP_MAGIC else 8
                _read_component_header(f, st)
                try:
                    _parse_info(st, allow_old=False)
                except RuntimeError as e:
                    raise RuntimeError(f"Error parsing component config in Comp #{i} from <{self.file}>") from e
                for _ in range(number):
                    buf = f.read(SPL_NAME_SIZE)
                    st.nparts.append(buf.split(b"\0", 1)[0].decode())
                self.stanzas.append(st)

    def _load(self) -> Dict[str, np.ndarray]:
        st = self.stanzas[self._cur]
        dt = self._dtype(st)
        parts = []
        got = 0
        for name in st.nparts:
            if got >= st.nbod:
                break
            # the reference opens the names as written, i.e. relative to the working directory (the run's outdir); a
            # master read from elsewhere finds its parts beside itself
            if not os.path.exists(name):
                name = os.path.join(os.path.dirname(self.file), os.path.basename(name))
            try:
                with open(name, "rb") as f:
                    raw = f.read(4)
                    if len(raw) < 4:
                        raise RuntimeError(f"Could not get particle count from <{name}>")
                    n, = struct.unpack("<I", raw)
                    n = min(n, st.nbod - got)
                    parts.append(np.fromfile(f, dtype=dt, count=n))
                    if len(parts[-1]) < n:
                        raise RuntimeError(f"SPL blob <{name}> is shorter than its particle count")
            except OSError as e:
                raise RuntimeError(f"Could not open SPL blob <{name}>") from e
            got += n
        rec = np.concatenate(parts) if parts else np.zeros(0, dt)
        return self._finish(st, rec[self.myid::self.numprocs], self.myid)


# ---------------------------------------------------------------------------------------------------------------
# Gadget-2 native
# ---------------------------------------------------------------------------------------------------------------
GADGET_HEADER = np.dtype([("npart", "<i4", (6,)), ("mass", "<f8", (6,)), ("time", "<f8"), ("redshift", "<f8"),
                          ("flag_sfr", "<i4"), ("flag_feedback", "<i4"), ("npartTotal", "<u4", (6,)),
                          ("flag_cooling", "<i4"), ("num_files", "<i4"), ("BoxSize", "<f8"), ("Omega0", "<f8"),
                          ("OmegaLambda", "<f8"), ("HubbleParam", "<f8"), ("flag_stellarage", "<i4"),
                          ("flag_metals", "<i4"), ("npartTotalHighWord", "<u4", (6,)), ("fill", "S64")],
                         align=False)                          # include/gadget.H: 256 bytes
assert GADGET_HEADER.itemsize == 256
GADGET_TYPES = ["Gas", "Halo", "Disk", "Bulge", "Stars", "Bndry"]


class _Gadget(ParticleReader):
    """What GadgetNative and GadgetHDF5 share (include/ParticleReader.H:103-234): six fixed types, "Halo" selected at
    construction, ONE file loaded at a time -- ``CurrentNumber`` is the count of the type in the file being read
    (``totalCount``), and the iteration runs through the files of the snapshot in turn."""

    _who = "Gadget"

    def __init__(self, files: Sequence[str], verbose: bool = False):
        super().__init__()
        self._files = list(files)
        self._verbose = verbose
        if len(self._files) == 1:
            scan = self.scanDirectory(self._files[0])
            if scan:
                self._files = scan
        self.ptype = 1
        self.time = 0.0
        self.totalCount = 0
        self.Pfound: List[str] = []
        self._get_numbers()
        if not self._files:
            sys.stderr.write(f"{self._who}: no files found\n")
        else:
            self.totalCount = self._count(self._files[0])

    def SelectType(self, name: str) -> None:
        if name not in GADGET_TYPES:
            sys.stderr.write(f"{self._who} error: no particle type <{name}>\nValid {self._who} types are: "
                             + " ".join(GADGET_TYPES) + "\n")
            raise RuntimeError(f"{self._who}: non-existent particle type")
        self.ptype = GADGET_TYPES.index(name)
        self._sel = None
        if self._files:
            self.totalCount = self._count(self._files[0])

    def CurrentNumber(self) -> int:
        return int(self.totalCount)

    def GetTypes(self) -> List[str]:
        return list(self.Pfound)

    def CurrentTime(self) -> float:
        return float(self.time)

    def _load(self) -> Dict[str, np.ndarray]:
        parts = [self._read_file(f) for f in self._files]
        if self._files:
            self.totalCount = self._count(self._files[-1])      # the last file read_and_load touched
        keys = ("mass", "pos", "vel", "indx")
        return {k: np.concatenate([p[k] for p in parts]) if parts else np.zeros((0, 3) if k in ("pos", "vel") else 0)
                for k in keys}


class GadgetNative(_Gadget):
    """exputil/ParticleReader.cc:29-318.  One deliberate difference: with several ranks the reference reads the
    mass block sequentially for its own particles only (:273-287 has no seek for the others), so rank r gets the masses
    of particles 0, 1, 2, ... instead of r, r + numprocs, ...; here every rank gets the masses of its own particles."""

    _who = "GadgetNative"

    def _header(self, f) -> np.ndarray:
        blk1 = f.read(4)
        h = np.frombuffer(f.read(256), dtype=GADGET_HEADER, count=1)[0]
        blk2 = f.read(4)
        if blk1 != blk2:
            print(f"GadgetNative header read: blk1={struct.unpack('<i', blk1)[0]} != blk2={struct.unpack('<i', blk2)[0]}")
        return h

    def _get_numbers(self) -> None:
        found = set()
        for name in self._files:
            try:
                f = open(name, "rb")
            except OSError as e:
                sys.stderr.write(f"Error opening file: {name}\n")
                raise RuntimeError("GadgetNative::getNumbers: open file error") from e
            with f:
                h = self._header(f)
            self.time = float(h["time"])
            found |= {GADGET_TYPES[k] for k in range(6) if h["npart"][k] > 0}
        self.Pfound = sorted(found)

    def _count(self, name: str) -> int:
        with open(name, "rb") as f:
            return int(self._header(f)["npart"][self.ptype])

    def _read_file(self, name: str) -> Dict[str, np.ndarray]:
        try:
            f = open(name, "rb")
        except OSError as e:
            raise RuntimeError(f"Error opening file: {name}") from e
        with f:
            h = self._header(f)
            self.time = float(h["time"])
            npart = [int(v) for v in h["npart"]]
            before, n = sum(npart[: self.ptype]), npart[self.ptype]
            total = sum(npart)

            def block(dtype, width, label):
                """one Fortran record holding `width` values per particle for all six types -> this type's rows"""
                b1, = struct.unpack("<i", f.read(4))
                start = f.tell()
                f.seek(before * width * 4, os.SEEK_CUR)
                x = np.fromfile(f, dtype=dtype, count=n * width)
                f.seek(start + total * width * 4)
                b2, = struct.unpack("<i", f.read(4))
                if b1 != b2:
                    print(f"GadgetNative {label} block read: blk1={b1} != blk2={b2}")
                return x.reshape(n, width) if width > 1 else x
            pos = block("<f4", 3, "position")
            vel = block("<f4", 3, "velocity")
            ids = block("<i4", 1, "id")
            # (the mass block exists when some populated type has a zero table entry, and holds those types only)
            if h["mass"][self.ptype] == 0 and n:
                b1, = struct.unpack("<i", f.read(4))
                f.seek(sum(npart[k] for k in range(self.ptype) if h["mass"][k] == 0) * 4, os.SEEK_CUR)
                mass = np.fromfile(f, dtype="<f4", count=n).astype(np.float64)
            else:
                mass = np.full(n, float(h["mass"][self.ptype]))
        s = slice(self.myid, None, self.numprocs)
        return {"mass": mass[s], "pos": pos[s].astype(np.float64), "vel": vel[s].astype(np.float64),
                "indx": ids[s].astype(np.int64).astype(np.uint64)}     # `indx = temp` (int -> unsigned long: sign-extended)


# ---------------------------------------------------------------------------------------------------------------
# Tipsy native
# ---------------------------------------------------------------------------------------------------------------
TIPSY_HEADER = np.dtype([("time", "<f8"), ("nbodies", "<i4"), ("ndim", "<i4"), ("nsph", "<i4"), ("ndark", "<i4"),
                         ("nstar", "<i4"), ("pad", "<i4")], align=False)       # include/tipsy.H:112-123: 32 bytes
TIPSY_GAS = np.dtype([("mass", "<f4"), ("pos", "<f4", (3,)), ("vel", "<f4", (3,)), ("rho", "<f4"), ("temp", "<f4"),
                      ("hsmooth", "<f4"), ("metals", "<f4"), ("phi", "<f4")], align=False)
TIPSY_DARK = np.dtype([("mass", "<f4"), ("pos", "<f4", (3,)), ("vel", "<f4", (3,)), ("eps", "<f4"), ("phi", "<f4")],
                      align=False)
TIPSY_STAR = np.dtype([("mass", "<f4"), ("pos", "<f4", (3,)), ("vel", "<f4", (3,)), ("metals", "<f4"),
                       ("tform", "<f4"), ("eps", "<f4"), ("phi", "<f4")], align=False)
assert (TIPSY_HEADER.itemsize, TIPSY_GAS.itemsize, TIPSY_DARK.itemsize, TIPSY_STAR.itemsize) == (32, 48, 36, 44)
TIPSY_TYPES = {"Gas": ("nsph", TIPSY_GAS), "Dark": ("ndark", TIPSY_DARK), "Star": ("nstar", TIPSY_STAR)}


class Tipsy(ParticleReader):
    """exputil/ParticleReader.cc:2091-2296, include/tipsy.H.  No type is selected at construction (``curName`` is empty:
    ``firstParticle`` before ``SelectType`` raises, as there).  Ranks take contiguous blocks of nsize / numprocs
    particles, the last one the remainder (``ios_psize``); the index of a native file's particle is its position in
    its group + 1, a Bonsai file carries its own."""

    class TipsyType:
        """the enum pyEXP exports (pyEXP/ParticleReaderWrappers.cc:563-567; ``export_values``: also ``Tipsy.native`` ...)"""
        native, xdr, bonsai1, bonsai = "native", "xdr", "bonsai1", "bonsai"

    native, xdr, bonsai = TipsyType.native, TipsyType.xdr, TipsyType.bonsai

    def __init__(self, files, ttype: str = "native", verbose: bool = False):
        super().__init__()
        if ttype == "xdr":
            # (a reference build without RPC/XDR falls back to the native reader for this type, :2117-2121; refusing is
            # the safer reading of a request for XDR data)
            raise RuntimeError("Tipsy: this build does not have RPC/XDR support; use Tipsy native format")
        if ttype not in ("native", "bonsai", "bonsai1"):
            raise RuntimeError(f"Tipsy: unknown file type <{ttype}>")
        self.files = [files] if isinstance(files, str) else list(files)
        if not isinstance(files, str) and len(self.files) == 1:
            scan = self.scanDirectory(self.files[0])
            if scan:
                self.files = scan
        self.ttype = ttype
        self.curName = ""
        self.Ngas = self.Ndark = self.Nstar = 0
        self.time = 0.0
        types = set()
        for name in self.files:
            h = self._header(name)
            for key, cnt in (("Gas", "nsph"), ("Dark", "ndark"), ("Star", "nstar")):
                if h[cnt]:
                    types.add(key)
            self.Ngas += int(h["nsph"]); self.Ndark += int(h["ndark"]); self.Nstar += int(h["nstar"])
            self.time = float(h["time"])
        self.curTypes = sorted(types)
        if not self.files:
            sys.stderr.write("Tipsy: no files found\n")
        else:
            self.time = float(self._header(self.files[0])["time"])

    @staticmethod
    def _header(name: str):
        try:
            with open(name, "rb") as f:
                raw = f.read(32)
        except OSError as e:
            raise RuntimeError(f"TipsyFile native error opening file <{name}>: {e}") from e
        if len(raw) < 32:
            raise RuntimeError("TipsyFile native: could not read a valid header")
        return np.frombuffer(raw, dtype=TIPSY_HEADER, count=1)[0]

    def GetTypes(self) -> List[str]:
        return list(self.curTypes)

    def SelectType(self, name: str) -> None:
        if name not in self.curTypes:
            raise RuntimeError(f"Tipsy error: no particle type <{name}>")
        self.curName = name
        self._sel = None

    def CurrentNumber(self) -> int:
        return {"Gas": self.Ngas, "Dark": self.Ndark, "Star": self.Nstar}.get(self.curName, 0)

    def CurrentTime(self) -> float:
        return self.time

    def _load(self) -> Dict[str, np.ndarray]:
        if self.curName not in TIPSY_TYPES:
            raise RuntimeError("Tipsy error: particle type must be one of Gas, Dark, Star. You selected ["
                               + self.curName + "]")
        parts = []
        for name in self.files:
            h = self._header(name)
            self.time = float(h["time"])
            off = 32
            for key in ("Gas", "Dark", "Star"):
                cnt, dt = TIPSY_TYPES[key]
                nsize = int(h[cnt])
                if key == self.curName:
                    psize = nsize // self.numprocs
                    first = psize * self.myid
                    if self.myid == self.numprocs - 1:
                        psize = nsize - first
                    rec = np.fromfile(name, dtype=dt, count=psize, offset=off + first * dt.itemsize) if psize else np.zeros(0, dt)
                    if key != "Gas" and self.ttype == "bonsai":
                        w = np.ascontiguousarray(np.stack([rec["eps"], rec["phi"]], axis=1)).view("<u8").reshape(-1)
                        indx = w.astype(np.uint64)                                        # ID2(): (eps, phi) as one uint64
                    elif key != "Gas" and self.ttype == "bonsai1":
                        indx = np.ascontiguousarray(rec["phi"]).view("<i4").astype(np.int64).astype(np.uint64)   # ID()
                    else:
                        indx = (first + 1 + np.arange(psize)).astype(np.uint64)           # getIndexOffset + pcount + 1
                    parts.append({"mass": rec["mass"].astype(np.float64), "pos": rec["pos"].astype(np.float64),
                                  "vel": rec["vel"].astype(np.float64), "indx": indx})
                off += nsize * dt.itemsize
        return {k: np.concatenate([p[k] for p in parts]) for k in ("mass", "pos", "vel", "indx")}


# ---------------------------------------------------------------------------------------------------------------
# writers
# ---------------------------------------------------------------------------------------------------------------
def _two_d(a, n: int, dtype):
    """attribute array -> [n, k] (k kept for an empty component handed a (0, k) array), or None"""
    if a is None:
        return None
    a = np.asarray(a, dtype=dtype)
    k = a.shape[1] if a.ndim == 2 else (a.size // n if n else 0)
    return a.reshape(n, k) if k else None


def _write_records(out, r_size: int, indexing: bool, indx, mass, pos, vel, pot, iattrib=None, dattrib=None) -> None:
    """Particle::writeBinary for a whole component (exputil/Particle.cc:333-388): doubles are narrowed to float with
    static_cast when rsize is 4."""
    n = len(mass)
    ia, da = _two_d(iattrib, n, np.int32), _two_d(dattrib, n, np.float64)
    dt = psp_record_dtype(r_size, indexing, 0 if ia is None else ia.shape[1], 0 if da is None else da.shape[1])
    rec = np.zeros(n, dtype=dt)
    if indexing:
        rec["indx"] = np.asarray(indx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        rec["mass"], rec["pos"], rec["vel"], rec["pot"] = mass, pos, vel, pot
        if ia is not None:
            rec["iattrib"] = ia
        if da is not None:
            rec["dattrib"] = da
    out.write(rec.tobytes())


def component_info(name: str, force_id: Optional[str] = None, force_parameters: Optional[dict] = None,
                   parameters: Optional[dict] = None, extra: Optional[dict] = None) -> str:
    """The YAML stanza a component carries in its PSP header (``outs << conf << std::endl``, src/Component.cc:2395-2396):
    name, parameters, bodyfile, force {id, parameters} as in the run's configuration file."""
    import yaml
    conf: dict = {"name": name, "parameters": dict(parameters or {})}
    conf.update(extra or {})
    if force_id is not None:
        conf["force"] = {"id": force_id, "parameters": dict(force_parameters or {})}
    return yaml.dump(conf, default_flow_style=False, sort_keys=False)


def _component_block(comp: dict, real4: bool):
    n = len(comp["mass"])
    info = comp["info"]
    if not isinstance(info, (bytes, bytearray)):
        info = (info if info.endswith("\n") else info + "\n").encode()
    ninfo = max(DEFAULT_INFO_SIZE, len(info))                   # grown when the stanza is longer (:2399-2408)
    ia, da = _two_d(comp.get("iattrib"), n, np.int32), _two_d(comp.get("dattrib"), n, np.float64)
    niatr = 0 if ia is None else ia.shape[1]
    ndatr = 0 if da is None else da.shape[1]
    import yaml
    try:
        conf = yaml.safe_load(info.decode())
        indexing = bool(((conf or {}).get("parameters") or {}).get("indexing", False))
    except yaml.YAMLError:
        indexing = False
    if comp.get("indexing") is not None and bool(comp["indexing"]) != indexing:
        raise RuntimeError("write_psp: the `indexing` of a component must be the one its info stanza states "
                           "(the readers take it from there)")
    head = struct.pack("<4i", n, niatr, ndatr, ninfo) + info.ljust(ninfo, b"\0")
    indx = comp.get("indx")
    if indexing and indx is None:
        indx = np.arange(1, n + 1, dtype=np.uint64)             # the sequence numbers of a body file (:1522-1526)
    pot = comp.get("pot")
    if pot is None:
        pot = np.zeros(n)
    if comp.get("potext") is not None:
        pot = np.asarray(pot) + np.asarray(comp["potext"])      # pot0 = pot + potext (:369)
    vel = comp.get("vel")
    if vel is None:
        vel = np.zeros((n, 3))
    return head, (4 if real4 else 8, indexing, indx, comp["mass"], comp["pos"], vel, pot, ia, da)


def write_psp(path: str, time: float, comps: Sequence[dict], real4: bool = False) -> None:
    """A monolithic PSP file (OutPSN::Run, src/OutPSN.cc:130-170 -> Component::write_binary, src/Component.cc:2385-2454).
    ``comps``: dicts with ``info`` (the YAML stanza, see ``component_info``), ``mass``, ``pos``, optionally ``vel``,
    ``pot``, ``potext``, ``indx``, ``iattrib``, ``dattrib``."""
    with open(path, "wb") as out:
        out.write(struct.pack("<dii", float(time), int(sum(len(c["mass"]) for c in comps)), len(comps)))
        for c in comps:
            head, rec = _component_block(c, real4)
            out.write(struct.pack("<Q", PSP_MAGIC + rec[0]))
            out.write(head)
            _write_records(out, *rec)


def write_spl(master: str, time: float, comps: Sequence[dict], nparts: int = 2, real4: bool = False) -> List[str]:
    """A split PSP file (OutPSQ::Run, src/OutPSQ.cc:160-215 -> Component::write_binary_header, src/Component.cc:2698-2760,
    and write_binary_particles, :2764-2785): the master holds, per component, magic, the number of part files, the
    header and the 1024-byte part names ``<master file name>_<component number>-<k>`` (stored WITHOUT the output
    directory, as there); part k -- one per process in the reference, ``nparts`` equal slices here -- starts with its
    unsigned particle count."""
    written = []
    base, outdir = os.path.basename(master), os.path.dirname(master)
    with open(master, "wb") as out:
        out.write(struct.pack("<dii", float(time), int(sum(len(c["mass"]) for c in comps)), len(comps)))
        for j, c in enumerate(comps):
            head, rec = _component_block(c, real4)
            n = len(c["mass"])
            edges = [(n * k) // nparts for k in range(nparts + 1)]
            names = [f"{base}_{j}-{k}" for k in range(nparts)]
            out.write(struct.pack("<Qi", PSP_MAGIC + rec[0], nparts))
            out.write(head)
            for name in names:
                out.write(name.encode().ljust(SPL_NAME_SIZE, b"\0"))
            r_size, indexing, indx, mass, pos, vel, pot, ia, da = rec
            for k, name in enumerate(names):
                s = slice(edges[k], edges[k + 1])
                with open(os.path.join(outdir, name), "wb") as part:
                    part.write(struct.pack("<I", edges[k + 1] - edges[k]))
                    _write_records(part, r_size, indexing, None if indx is None else np.asarray(indx)[s],
                                   np.asarray(mass)[s], np.asarray(pos)[s], np.asarray(vel)[s], np.asarray(pot)[s],
                                   None if ia is None else ia[s], None if da is None else da[s])
                written.append(os.path.join(outdir, name))
    return written


def read_bodies_ascii(path: str, aindex: bool = False) -> Dict[str, np.ndarray]:
    """The body file of a component (Component::read_bodies_and_distribute_ascii, src/Component.cc:1462-1560;
    Particle::readAscii, exputil/Particle.cc:469-505): first line ``nbodies niattrib ndattrib``, then per particle
    ``[indx] mass x y z u v w [iattrib ...] [dattrib ...]``; attributes a line does not hold are zero; without ``aindex``
    the index is the line's sequence number, from 1."""
    with open(path) as f:
        head = f.readline().split()
        if not head:
            raise RuntimeError("Error reading nbodies_tot . . . quitting")
        try:
            n = int(head[0])
        except ValueError as e:
            raise RuntimeError("Error reading nbodies_tot . . . quitting") from e
        ni = int(head[1]) if len(head) > 1 else 0
        nd = int(head[2]) if len(head) > 2 else 0
        mass, pos, vel = np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3))
        indx = np.arange(1, n + 1, dtype=np.uint64)
        ia, da = np.zeros((n, ni), dtype=np.int32), np.zeros((n, nd))
        for i in range(n):
            tok = f.readline().split()
            k = 0
            if aindex:
                indx[i] = int(tok[0]); k = 1
            vals = [float(t) for t in tok[k:k + 7]] + [0.0] * 7
            mass[i], pos[i], vel[i] = vals[0], vals[1:4], vals[4:7]
            rest = tok[k + 7:]
            for j in range(ni):
                try:
                    ia[i, j] = int(rest[j])
                except (IndexError, ValueError):
                    break                                        # `if (!ins) it = 0`: once the stream fails all that follow are 0
            for j in range(nd):
                try:
                    da[i, j] = float(rest[ni + j])
                except (IndexError, ValueError):
                    break
    out = {"mass": mass, "pos": pos, "vel": vel, "indx": indx}
    if ni:
        out["iattrib"] = ia
    if nd:
        out["dattrib"] = da
    return out


def write_bodies_ascii(path: str, mass, pos, vel, iattrib=None, dattrib=None, indx=None) -> None:
    """A body file ``read_bodies_ascii`` (and EXP) reads; 17 significant digits so that doubles round-trip."""
    n = len(mass)
    ia = None if iattrib is None else np.asarray(iattrib).reshape(n, -1)
    da = None if dattrib is None else np.asarray(dattrib).reshape(n, -1)
    with open(path, "w") as f:
        f.write(f"{n} {0 if ia is None else ia.shape[1]} {0 if da is None else da.shape[1]}\n")
        for i in range(n):
            row = ([str(int(indx[i]))] if indx is not None else []) + ["%.17g" % mass[i]]
            row += ["%.17g" % v for v in pos[i]] + ["%.17g" % v for v in vel[i]]
            if ia is not None:
                row += [str(int(v)) for v in ia[i]]
            if da is not None:
                row += ["%.17g" % v for v in da[i]]
            f.write(" ".join(row) + "\n")
