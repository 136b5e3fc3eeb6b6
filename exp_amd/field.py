"""``pyEXP.field.FieldGenerator`` (expui/FieldGenerator.H, .cc; pyEXP/FieldWrappers.cc:94-470): density, potential and
force fields of a basis over the coefficient sets of a ``Coefs`` container -- on a planar slice, a line probe, an
arbitrary point mesh or a volume -- and the files the reference writes from them.

The reference walks the pixels with an OpenMP loop and one ``(*basis)(x1, x2, x3, ctype)`` call each; here the points
of a frame go to the device in ONE ``exp_amd_sph_fields`` / ``exp_amd_cyl_fields`` launch (``basis.__call__`` with
arrays), in the basis' own coordinate type exactly as the reference converts them (``r + 1e-18``, ``R + 1e-18``).
Results are float32 like the reference's ``Eigen::MatrixXf`` / ``VectorXf`` / ``Tensor<float, 3>``.

The three particle histograms (``histo2d``, ``histo1d``, ``histo1dlog``) take a ``ParticleReader`` (``exp_amd.reader``);
their float accumulators are filled one particle at a time in reader order (``exp_amd_host_binsum_f32``), which is what
the result depends on in the last bits.  With several ranks each reads its share and the float sums are added, as the
reference's MPI_Reduce(MPI_FLOAT, MPI_SUM) does."""
from __future__ import annotations

import os
from typing import Dict, Sequence

import numpy as np


def _nan(v: float) -> str:
    """a NaN as the C library prints it: with its sign (0/0 on x86 is the NEGATIVE quiet NaN: a one-point axis of a grid
    reads `-nan` in the reference's files); Python's % drops it"""
    return "-nan" if np.signbit(v) else "nan"


def _fmt_general(v: float) -> str:
    """operator<< of a float with the stream's default flags (6 significant digits, %g)."""
    return _nan(v) if v != v else "%g" % float(v)


def _fmt_sci8(v: float) -> str:
    return _nan(v) if v != v else "%.8e" % float(v)


class _VtrGrid:
    """The rectilinear-grid file of a build without the VTK library (exputil/VtkGrid.cc:140-289): ASCII ``.vtr`` XML,
    point data in name order with ``<`` / ``>`` spelled out, values six to a line in ``%.8e``; coordinates X, Y, Z."""

    def __init__(self, nx, ny, nz, xmin, xmax, ymin, ymax, zmin, zmax):
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.coord = {"X": (xmin + (xmax - xmin) * np.arange(self.nx) / np.float64(self.nx - 1)).astype(np.float32),
                          "Y": (ymin + (ymax - ymin) * np.arange(self.ny) / np.float64(self.ny - 1)).astype(np.float32)}
            if self.nz > 1:
                self.coord["Z"] = (zmin + (zmax - zmin) * np.arange(self.nz) / np.float64(self.nz - 1)).astype(np.float32)
            else:
                self.coord["Z"] = np.zeros(1, np.float32)
        self.data: Dict[str, np.ndarray] = {}

    def Add(self, data: np.ndarray, name: str) -> None:
        """data[(k*ny + j)*nx + i]"""
        self.data[name.replace(">", ".gt.").replace("<", ".lt.")] = np.asarray(data, np.float64).astype(np.float32)

    def Write(self, name: str) -> str:
        path = name + ".vtr"
        try:
            fout = open(path, "w")
        except OSError:
            raise RuntimeError(f"VtkGrid::Write: could not open file <{path}>")
        fmt = _fmt_general                    # the stream turns scientific with the first value written, and stays so
        with fout:
            ext = f"0 {self.nx - 1} 0 {self.ny - 1} 0 {self.nz - 1}"
            fout.write('<?xml version="1.0"?>\n')
            fout.write('<VTKFile type="RectilinearGrid" version="0.1" byte_order="LittleEndian" header_type="UInt32">\n')
            fout.write(f'  <RectilinearGrid WholeExtent="{ext}">\n')
            fout.write(f'  <Piece Extent="{ext}">\n')
            fout.write("    <PointData>\n")
            for key in sorted(self.data):
                v = self.data[key]
                fout.write(f'      <DataArray type="Float32" Name="{key}"  format="ascii" RangeMin="{fmt(v.min())}" '
                           f'RangeMax="{fmt(v.max())}">\n')
                fmt = _fmt_sci8
                self._values(fout, v, "         ")
                fout.write("      </DataArray>\n")
            fout.write("    </PointData>\n    <CellData>\n    </CellData>\n    <Coordinates>\n")
            for key in sorted(self.coord):
                v = self.coord[key]
                fout.write(f'      <DataArray type="Float32" Name="{key}" format="ascii" RangeMin="{fmt(v.min())}" '
                           f'RangeMax="{fmt(v.max())}">\n')
                fmt = _fmt_sci8
                self._values(fout, v, "        ")
                fout.write("      </DataArray>\n")
            fout.write("    </Coordinates>\n  </Piece>\n  </RectilinearGrid>\n</VTKFile>\n")
        return path

    @staticmethod
    def _values(fout, v, indent):
        for c in range(0, len(v), 6):
            fout.write(indent + "".join(_fmt_sci8(x) + " " for x in v[c:c + 6]) + "\n")


class FieldGenerator:
    """``FieldGenerator(times, lower, upper, gridsize)`` for slices and volumes (a slice has exactly one zero in
    ``gridsize``), ``FieldGenerator(times, mesh)`` with an [N, 3] array for ``points``."""

    def __init__(self, times: Sequence[float] = (), lower=None, upper=None, gridsize=None, mesh=None):
        self.times = [float(t) for t in times]
        self.pmin = self.pmax = self.grid = None
        self.mesh = None
        if lower is not None and upper is None and gridsize is None and mesh is None:
            mesh, lower = lower, None                     # the two-argument overload
        if mesh is not None:
            self.mesh = np.asarray(mesh, dtype=np.float64)
            if self.mesh.ndim != 2 or self.mesh.shape[1] != 3:
                raise RuntimeError("FieldGenerator: bad mesh specification. The mesh must be an Nx3 array where the columns "
                                   "are Cartesian points")
        if lower is not None:
            self.pmin = [float(v) for v in lower]
            self.pmax = [float(v) for v in upper]
            self.grid = [int(v) for v in gridsize]
        self.midplane = False
        self.colheight = 4.0

    # -- particle histograms (expui/FieldGenerator.cc:776-1009; pyEXP/FieldWrappers.cc:273-350) --
    @staticmethod
    def _binsum(bins: np.ndarray, vals: np.ndarray, nbins: int) -> np.ndarray:
        from ._lib import load
        import ctypes
        lib = load()
        out = np.zeros(nbins, dtype=np.float32)
        b = np.ascontiguousarray(bins, dtype=np.int32)
        v = np.ascontiguousarray(vals, dtype=np.float64)
        rc = lib.exp_amd_host_binsum_f32(len(b), b.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p),
                                         int(nbins), out.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise RuntimeError("FieldGenerator: exp_amd_host_binsum_f32 failed")
        return out

    @staticmethod
    def _reduce_f32(a: np.ndarray) -> np.ndarray:
        """MPI_Reduce(MPI_FLOAT, MPI_SUM) to the root: every rank gets the sum here"""
        import sys
        dist = sys.modules.get("torch.distributed")           # (only a process that made a group has it; no import here)
        if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            import torch
            t = torch.from_numpy(np.ascontiguousarray(a))
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.all_reduce(t)
            return t.cpu().numpy()
        return a

    def histogram2d(self, reader, center=(0.0, 0.0, 0.0)) -> Dict[str, np.ndarray]:
        """Surface density of the reader's particles on the generator's grid, projected along each axis whose two
        others have a positive grid size: {"xy": [g0, g1], "xz": [g0, g2], "yz": [g1, g2]}, float32."""
        if self.grid is None:
            raise RuntimeError("FieldGenerator::histogram2d: no grid was given to the constructor")
        a = reader.arrays()
        ctr = np.asarray(center, dtype=np.float64).reshape(3)
        pmin, pmax, grid = np.array(self.pmin), np.array(self.pmax), self.grid
        dl = np.array([(pmax[k] - pmin[k]) / grid[k] if grid[k] > 0 else 0.0 for k in range(3)])
        pp = a["pos"].astype(np.float64) - ctr
        bb = (pp >= pmin) & (pp < pmax) & (dl > 0.0)
        with np.errstate(divide="ignore", invalid="ignore"):
            idx = np.floor((pp - pmin) / np.where(dl > 0.0, dl, 1.0))
        idx = np.where(np.isfinite(idx), idx, -1).astype(np.int64)
        ret = {}
        for key, (i, j) in (("xy", (0, 1)), ("xz", (0, 2)), ("yz", (1, 2))):
            if not (grid[i] > 0 and grid[j] > 0):
                continue
            fac = 1.0 / (dl[i] * dl[j])
            ok = bb[:, i] & bb[:, j] & (idx[:, i] >= 0) & (idx[:, i] < grid[i]) & (idx[:, j] >= 0) & (idx[:, j] < grid[j])
            flat = np.where(ok, idx[:, i] * grid[j] + idx[:, j], -1)
            ret[key] = self._reduce_f32(self._binsum(flat, a["mass"] * fac, grid[i] * grid[j])).reshape(grid[i], grid[j])
        return ret

    def histogram1d(self, reader, rmax: float, nbins: int, proj: str, center=(0.0, 0.0, 0.0)) -> np.ndarray:
        """Density in ``nbins`` linear bins out to ``rmax``: cylindrical rings of the plane ``proj`` in ("xy", "xz", "yz"),
        spherical shells for "r".  Any other string leaves the reference's projection variable unset (its error is
        built and dropped, :871-876); it is refused here."""
        axes = {"xy": (0, 1), "xz": (0, 2), "yz": (1, 2), "r": (0, 1, 2)}
        if proj not in axes:
            raise RuntimeError(f'FieldGenerator::histogram1d: error parsing projection <{proj}>.  Must be one of '
                               '"xy", "xz", "yz, "r".')
        a = reader.arrays()
        ctr = np.asarray(center, dtype=np.float64).reshape(3)
        dl = float(rmax) / nbins
        pp = a["pos"].astype(np.float64) - ctr
        rad = np.zeros(len(pp))
        for k in axes[proj]:                                  # summed in the order k = 0, 1, 2
            rad = rad + pp[:, k] * pp[:, k]
        q = np.floor(np.sqrt(rad) / dl)
        bins = np.where(np.isfinite(q) & (q >= 0) & (q < nbins), q, -1).astype(np.int64)
        ret = self._reduce_f32(self._binsum(bins, a["mass"].astype(np.float64), nbins))
        i = np.arange(nbins)
        pi = 3.14159265358979323846
        if proj == "r":
            ret = (ret.astype(np.float64) / (4.0 * pi / 3.0 * dl * dl * dl * (3 * i * (i + 1) + 1))).astype(np.float32)
        else:
            ret = (ret.astype(np.float64) / (pi * dl * dl * (2 * i + 1))).astype(np.float32)
        return ret

    def histo1dlog(self, reader, rmin: float, rmax: float, nbins: int, center=(0.0, 0.0, 0.0)):
        """Spherical shells, logarithmic in radius -> (bin-centre radii, density, velocity dispersion), float32."""
        if rmin <= 0.0:
            raise RuntimeError("FieldGenerator::histo1dlog: rmin must be > 0.0")
        if rmax <= rmin:
            raise RuntimeError("FieldGenerator::histo1dlog: rmax must be > rmin")
        import math
        a = reader.arrays()
        ctr = np.asarray(center, dtype=np.float64).reshape(3)
        lrmin, lrmax = math.log(rmin), math.log(rmax)
        dl = (lrmax - lrmin) / nbins
        pp = a["pos"].astype(np.float64) - ctr
        r2 = (pp[:, 0] * pp[:, 0] + pp[:, 1] * pp[:, 1]) + pp[:, 2] * pp[:, 2]
        with np.errstate(divide="ignore"):
            q = np.floor((np.log(np.sqrt(r2)) - lrmin) / dl)
        bins = np.where(np.isfinite(q) & (q >= 0) & (q < nbins), q, -1).astype(np.int64)
        m, v = a["mass"].astype(np.float64), a["vel"].astype(np.float64)
        ret = self._reduce_f32(self._binsum(bins, m, nbins))
        vc1 = np.stack([self._reduce_f32(self._binsum(bins, m * v[:, k], nbins)) for k in range(3)], axis=1)
        vc2 = np.stack([self._reduce_f32(self._binsum(bins, m * v[:, k] * v[:, k], nbins)) for k in range(3)], axis=1)
        i = np.arange(nbins)
        rf = 4.0 * 3.14159265358979323846 / 3.0 * (math.exp(3.0 * dl) - 1.0)
        rad = np.exp(lrmin + dl * (0.5 + i)).astype(np.float32)
        has = ret > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            c1 = (vc1 / ret[:, None]).astype(np.float32)                      # float / float
            c2 = (vc2 / ret[:, None]).astype(np.float32)
            sig = np.zeros(nbins)
            for k in range(3):
                sig = sig + (c2[:, k] - c1[:, k] * c1[:, k]).astype(np.float64)   # double += float - float * float
            dens = (ret.astype(np.float64) / (np.exp(3.0 * (lrmin + dl * i)) * rf)).astype(np.float32)
            vel = np.sqrt(np.abs(sig)).astype(np.float32)
        return rad, np.where(has, dens, np.float32(0)), np.where(has, vel, np.float32(0))

    histo2d, histo1d = histogram2d, histogram1d               # the names pyEXP binds (pyEXP/FieldWrappers.cc:273, :295)

    # -- expui/FieldGenerator.H:150-156 --
    def setMidplane(self, value: bool) -> None:
        self.midplane = bool(value)

    def setColumnHeight(self, value: float) -> None:
        self.colheight = float(value)

    def _check_times(self, coefs) -> None:
        have = coefs.Times()
        for t in self.times:
            if t not in have:
                raise RuntimeError(f"FieldGenerator: requested time <{t:g}> not in DB\n")

    @staticmethod
    def _eval(basis, x, y, z) -> np.ndarray:
        """[n, nlabels] at Cartesian points, through the basis' coordinate type as the reference converts them."""
        ctype = basis.coordinates
        if ctype == "spherical":
            r = np.sqrt(x * x + y * y + z * z) + 1.0e-18
            return np.atleast_2d(basis(r, z / r, np.arctan2(y, x), ctype))
        if ctype == "cylindrical":
            R = np.sqrt(x * x + y * y) + 1.0e-18
            return np.atleast_2d(basis(R, z, np.arctan2(y, x), ctype))
        return np.atleast_2d(basis(x, y, z, "cartesian"))

    def _frames(self, basis, coefs, x, y, z, shape, extra=None):
        labels = basis.getFieldLabels(basis.coordinates)
        ret = {}
        for T in self.times:
            cs = coefs.getCoefStruct(T)
            if cs is None:
                print(f"Could not find time={T:g}, continuing")
                continue
            basis.set_coefs(cs)
            v = self._eval(basis, x, y, z)
            frame = dict(extra) if extra else {}
            for n, s in enumerate(labels):
                frame[s] = v[:, n].astype(np.float32).reshape(shape)
            ret[T] = frame
        return ret

    # -- expui/FieldGenerator.cc:1011-1175 --
    def points(self, basis, coefs):
        if self.mesh is None or self.mesh.size == 0:
            raise RuntimeError("FieldGenerator::points: bad mesh specification.  Did you call the mesh constructor?")
        basis.setMidplane(self.midplane)
        basis.setColumnHeight(self.colheight)
        self._check_times(coefs)
        m = self.mesh
        return self._frames(basis, coefs, m[:, 0].copy(), m[:, 1].copy(), m[:, 2].copy(), (m.shape[0],))

    def _slice_axes(self, who):
        i1 = i2 = i3 = -1
        for i, g in enumerate(self.grid or []):
            if g > 0:
                if i1 < 0:
                    i1 = i
                elif i2 < 0:
                    i2 = i
            else:
                i3 = i
        if i1 < 0 or i2 < 0 or i3 < 0:
            raise RuntimeError(f"FieldGenerator::{who}: bad grid specification")
        return i1, i2, i3

    # -- expui/FieldGenerator.cc:328-509 --
    def slices(self, basis, coefs):
        basis.setMidplane(self.midplane)
        basis.setColumnHeight(self.colheight)
        self._check_times(coefs)
        i1, i2, _ = self._slice_axes("slices")
        n1, n2 = self.grid[i1], self.grid[i2]
        d1 = (self.pmax[i1] - self.pmin[i1]) / max(n1 - 1, 1)
        d2 = (self.pmax[i2] - self.pmin[i2]) / max(n2 - 1, 1)
        pp = [np.full(n1 * n2, self.pmin[k]) for k in range(3)]
        ii, jj = np.divmod(np.arange(n1 * n2), n2)                    # pixel k -> (i, j), j fastest
        pp[i1] = self.pmin[i1] + d1 * ii
        pp[i2] = self.pmin[i2] + d2 * jj
        return self._frames(basis, coefs, pp[0], pp[1], pp[2], (n1, n2))

    # -- expui/FieldGenerator.cc:87-253 --
    def lines(self, basis, coefs, beg, end, num: int):
        beg, end = [float(v) for v in beg], [float(v) for v in end]
        if len(beg) != 3 or len(end) != 3:
            raise RuntimeError("FieldGenerator::lines: vectors beg and end must have rank 3")
        if num < 1:
            raise RuntimeError("FieldGenerator::lines: number of evaluation points must be > 0")
        self._check_times(coefs)
        with np.errstate(divide="ignore", invalid="ignore"):
            dd = [(np.float64(end[k]) - beg[k]) / np.float64(num - 1) for k in range(3)]
            dlen = np.sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2])
            n = np.arange(num)
            x, y, z = beg[0] + dd[0] * n, beg[1] + dd[1] * n, beg[2] + dd[2] * n
            extra = {"x": x.astype(np.float32), "y": y.astype(np.float32), "z": z.astype(np.float32),
                     "arc": (dlen * n).astype(np.float32)}
        return self._frames(basis, coefs, x, y, z, (num,), extra)

    # -- expui/FieldGenerator.cc:566-722 --
    def volumes(self, basis, coefs):
        g = self.grid
        if g is None or len(g) != 3:
            raise RuntimeError("FieldGenerator::volumes: bad grid specification")
        d = [(self.pmax[k] - self.pmin[k]) / max(g[k] - 1, 1) for k in range(3)]
        n = np.arange(g[0] * g[1] * g[2])
        i = n // (g[1] * g[2])
        j = (n - i * g[1] * g[2]) // g[2]
        k = n - (i * g[1] + j) * g[2]
        return self._frames(basis, coefs, self.pmin[0] + d[0] * i, self.pmin[1] + d[1] * j, self.pmin[2] + d[2] * k,
                            (g[0], g[1], g[2]))

    # -- files --
    @staticmethod
    def _need_dir(who, outdir):
        if not os.path.isdir(outdir):
            raise RuntimeError(f"FieldGenerator::{who}: directory <{outdir}> does not exist")

    def file_lines(self, basis, coefs, beg, end, num: int, prefix: str, outdir: str = ".") -> None:
        """One ``<prefix>_probe_<k>.txt`` per time (expui/FieldGenerator.cc:255-325): header lines, then 16-wide columns
        in label order."""
        self._need_dir("file_lines", outdir)
        db = self.lines(basis, coefs, beg, end, num)
        for icnt, T in enumerate(sorted(db)):
            frame = db[T]
            keys = sorted(frame)
            path = os.path.join(outdir, f"{prefix}_probe_{icnt}.txt")
            try:
                out = open(path, "w")
            except OSError:
                raise RuntimeError(f"FieldGenerator::file_lines: couldn't open <{path}>")
            with out:
                out.write(f"# T={T:g}\n")
                out.write("".join(("#%15s" if c == 0 else "%16s") % (k + " ") for c, k in enumerate(keys)) + "\n")
                out.write("".join(("#%15s" if c == 0 else "%16s") % f"[{c + 1}] " for c in range(len(keys))) + "\n")
                out.write("".join(("#%15s" if c == 0 else "%16s") % ("-" * 10) for c in range(len(keys))) + "\n")
                for i in range(num):
                    out.write("".join("%16s" % _fmt_general(frame[k][i]) for k in keys) + "\n")

    def file_slices(self, basis, coefs, prefix: str, outdir: str = ".") -> None:
        """One ``<prefix>_surface_<k>.vtr`` per time (expui/FieldGenerator.cc:512-564)."""
        self._need_dir("file_slices", outdir)
        db = self.slices(basis, coefs)
        i1, i2, _ = self._slice_axes("file_slices")
        for icnt, T in enumerate(sorted(db)):
            dg = _VtrGrid(self.grid[i1], self.grid[i2], 1, self.pmin[i1], self.pmax[i1], self.pmin[i2], self.pmax[i2], 0, 0)
            for key, v in db[T].items():
                dg.Add(np.asarray(v, np.float64).T.reshape(-1), key)          # tmp[j*n1 + i] = v(i, j)
            dg.Write(os.path.join(outdir, f"{prefix}_surface_{icnt}"))

    def file_volumes(self, basis, coefs, prefix: str, outdir: str = ".") -> None:
        """One ``<prefix>_volume_<k>.vtr`` per time (expui/FieldGenerator.cc:725-774)."""
        self._need_dir("file_volumes", outdir)
        db = self.volumes(basis, coefs)
        g = self.grid
        for icnt, T in enumerate(sorted(db)):
            dg = _VtrGrid(g[0], g[1], g[2], self.pmin[0], self.pmax[0], self.pmin[1], self.pmax[1], self.pmin[2], self.pmax[2])
            for key, v in db[T].items():
                dg.Add(np.asarray(v, np.float64).transpose(2, 1, 0).reshape(-1), key)   # tmp[(k*n1 + j)*n0 + i] = v(i, j, k)
            dg.Write(os.path.join(outdir, f"{prefix}_volume_{icnt}"))
