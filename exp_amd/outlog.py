"""``OutLog`` (src/OutLog.cc, src/OutLog.H): the run log ``OUTLOG.<runtag>`` -- one row of global and per-component
conserved quantities every ``nint`` steps: mass, bodies, centre of mass and of velocity, angular momentum, the expansion
centre, kinetic and potential energy, the Clausius virial, their sum and the virial ratio -2T/VC, the wall-clock time per
step and the number of particles the force used.  The reference's own N-body acceptance test reads it
(tests/Halo/check.py: the mean of column 17, 2T/VC, within 5.5 % of 1).

PE is 0.5 m pot + m potext (src/OutLog.cc:490-491, :589, :642).  The cross forces between components add to ``pot`` in
the reference as here (``Component::AddPot``, src/SphericalBasis.cc:1652, src/Cylinder.cc:1416); ``potext`` only ever
receives the External force plug-ins (``AddPotExt``), which are outside this build: it is zero, and the store keeps no
array for it.

The sums over the particles are the device's (``exp_amd_comp_log_sums``); the file is the reference's, character for
character: a six-line header, then ``setw(10 + precision)`` columns in scientific notation joined by ``|``."""
from __future__ import annotations

import os
import time as _time
from typing import List, Optional, Sequence

import numpy as np

LAB_GLOBAL = ["Time", "Mass", "Bodies", "R(x)", "R(y)", "R(z)", "V(x)", "V(y)", "V(z)", "L(x)", "L(y)", "L(z)",
              "KE", "PE", "VC", "E", "2T/VC", "Clock", "# used"]                                  # :11-31
LAB_COMPONENT = ["mass", "bodies", "R(x)", "R(y)", "R(z)", "V(x)", "V(y)", "V(z)", "L(x)", "L(y)", "L(z)",
                 "C(x)", "C(y)", "C(z)", "KE", "PE", "VC", "E", "2T/VC", "# used"]                # :33-54


def row_from_sums(tnow: float, sums: Sequence[dict], centers: Sequence[Sequence[float]], used: Sequence[int],
                  wtime: float, precision: int = 10) -> str:
    """The data row of OutLog::Run from the reduced sums (src/OutLog.cc:480-590)."""
    cwid = 10 + precision
    f = lambda v: f"{float(v):{cwid}.{precision}e}"
    d = lambda v: f"{int(v):{cwid}d}"
    mtot0 = 0.0
    for s in sums:
        mtot0 += s["mtot"]
    nb0 = sum(int(s["nbodies"]) for s in sums)
    com0, cov0, angm0 = np.zeros(3), np.zeros(3), np.zeros(3)
    for s in sums:                                            # com_system off: the inertial sums are the local ones
        com0 += s["com"]; cov0 += s["cov"]; angm0 += s["angm"]
    cols = [f(tnow), f(mtot0), d(nb0)]
    cols += [f(com0[j] / mtot0 if mtot0 > 0.0 else 0.0) for j in range(3)]
    cols += [f(cov0[j] / mtot0 if mtot0 > 0.0 else 0.0) for j in range(3)]
    cols += [f(angm0[j]) for j in range(3)]
    ek0 = ep0 = cl0 = 0.0
    for s in sums:
        ek0 += s["ektot"]
    for s in sums:
        ep0 += s["eptot"] + s.get("eptotx", 0.0)
    for s in sums:
        cl0 += s["clausius"]
    cols += [f(ek0), f(ep0), f(cl0), f(ek0 + cl0), f(-2.0 * ek0 / cl0 if cl0 != 0.0 else 0.0), f(wtime),
             d(sum(int(u) for u in used))]
    for s, ctr, u in zip(sums, centers, used):
        m = s["mtot"]
        cols += [f(m), d(s["nbodies"])]
        cols += [f(s["com"][j] / m if m > 0.0 else 0.0) for j in range(3)]
        cols += [f(s["cov"][j] / m if m > 0.0 else 0.0) for j in range(3)]
        cols += [f(s["angm"][j]) for j in range(3)] + [f(ctr[j]) for j in range(3)]
        vbar2 = 0.0                                           # kinetic energy in the centre-of-velocity frame (:563-570)
        if m > 0.0:
            for j in range(3):
                vbar2 += s["cov"][j] * s["cov"][j]
            vbar2 /= m * m
        ek = s["ektot"]
        if s["nbodies"] > 1:
            ek -= 0.5 * m * vbar2
        ep = s["eptot"] + s.get("eptotx", 0.0)
        cl = s["clausius"]
        cols += [f(ek), f(ep), f(cl), f(ek + cl), f(-2.0 * ek / cl if cl != 0.0 else 0.0), d(u)]
    return "|".join(cols) + "\n"


def header(names: Sequence[str], ids: Sequence[str], precision: int = 10) -> str:
    """The six header lines (src/OutLog.cc:278-356)."""
    cwid = 10 + precision
    ng, nc = len(LAB_GLOBAL), len(LAB_COMPONENT)
    blank, dash = " " * cwid, "-" * cwid
    rule = dash + ("+" + dash) * (ng - 1 + nc * len(names)) + "\n"
    out = "Global stats".rjust(cwid, "-") + ("|" + blank) * (ng - 1)
    for cid in ids:
        # (the fill character is still '-' here when there is a single global column; with 19 it is ' ')
        out += "|" + cid.rjust(cwid) + ("|" + blank) * (nc - 1)
    out += "\n" + rule
    out += LAB_GLOBAL[0].rjust(cwid) + "".join("|" + s.rjust(cwid) for s in LAB_GLOBAL[1:])
    for name in names:
        for s in LAB_COMPONENT:
            label = name + " " + s
            out += "|" + (label.rjust(cwid) if len(label) <= cwid else label)
    out += "\n" + rule
    count = ng + nc * len(names)
    out += "[1]".rjust(cwid) + "".join("|" + f"[{k}]".rjust(cwid) for k in range(2, count + 1)) + "\n" + rule
    return out


_ATOF = __import__("re").compile(r"^[ \t]*([+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)")


def _atof(tok: str) -> float:
    """C's atof: the longest leading numeric prefix, 0.0 when there is none (`1.5|` -> 1.5, `abc` -> 0.0)"""
    m = _ATOF.match(tok)
    return float(m.group(1)) if m else 0.0


class OutLog:
    """``OutLog(filename | outdir + runtag, nint, precision)``; ``run(n, tnow, last)`` appends a row when ``n % nint == 0``
    (or ``last``).  Components are registered with their name, force id, particle store and force (for ``Used()``)."""

    def __init__(self, filename: Optional[str] = None, nint: int = 1, precision: int = 10, outdir: str = "",
                 runtag: str = "run0", restart: bool = False):
        self.filename = filename or os.path.join(outdir, "OUTLOG." + runtag)
        self.nint, self.precision, self.restart = max(int(nint), 1), int(precision), bool(restart)
        self.comps: List[tuple] = []
        self.firstime = True
        self.laststep, self.lastwtime = -1, _time.time()

    def add_component(self, name: str, force_id: str, comp, force) -> None:
        self.comps.append((name, force_id, comp, force))

    def _first(self, tnow: float) -> None:
        self.firstime = False
        if not self.restart:
            with open(self.filename, "a") as out:
                out.write(header([c[0] for c in self.comps], [c[1] for c in self.comps], self.precision))
            return
        # restart (:210-275): the old log becomes <filename>.bak; header and the rows up to the current time are kept
        backup = self.filename + ".bak"
        os.replace(self.filename, backup)
        with open(backup) as src, open(self.filename, "w") as out:
            # std::getline semantics (src/OutLog.cc:271-287): a last row WITHOUT a trailing newline is still a row (characters
            # were extracted, only eofbit is set); the empty fragment behind a trailing newline is not
            lines = src.read().split("\n")
            if lines and lines[-1] == "":
                lines.pop()
            k = 0
            while k < len(lines):
                out.write(lines[k] + "\n")
                k += 1
                if any(ch in lines[k - 1] for ch in "Time"):   # find_first_of("Time"): ANY of the four letters
                    break
            while k < len(lines):
                tok = [t for t in lines[k].split(" ") if t]
                ttim = _atof(tok[0]) if tok else 0.0
                if tnow < ttim:
                    break
                out.write(lines[k] + "\n")
                k += 1

    def run(self, n: int, tnow: float, last: bool = False) -> Optional[str]:
        if self.firstime:
            self._first(tnow)
        if n % self.nint and not last:
            return None
        wtime = 0.0
        if n > self.laststep:
            cur = _time.time()
            wtime = (cur - self.lastwtime) / (n - self.laststep)
            self.lastwtime, self.laststep = cur, n
        sums = [c[2].log_sums() for c in self.comps]
        centers = [getattr(c[2], "center", None) if getattr(c[2], "center", None) is not None else np.zeros(3)
                   for c in self.comps]
        used = [int(c[3].Used()) if c[3] is not None else 0 for c in self.comps]
        row = row_from_sums(tnow, sums, centers, used, wtime, self.precision)
        with open(self.filename, "a") as out:
            out.write(row)
        return row
