"""Loaders for the committed golden fixtures (tests/golden/*.npz)."""
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_sph():
    from exp_amd.slgrid import SLGridSph
    z = np.load(os.path.join(HERE, "sph_small.npz"))
    g = SLGridSph(lmax=int(z["lmax"]), nmax=int(z["nmax"]), numr=int(z["numr"]), cmap=int(z["cmap"]),
                  rmin=float(z["rmin"]), rmax=float(z["rmax"]), rmap=float(z["rmap"]),
                  xmin=float(z["xmin"]), xmax=float(z["xmax"]), dxi=float(z["dxi"]), xi=z["xi"],
                  r=z["r"], p0=z["p0"], d0=z["d0"], ev=z["ev"], ef=z["ef"])
    return g, z


def load_cyl():
    from exp_amd.empcyl import EmpCylGrid
    z = np.load(os.path.join(HERE, "cyl_small.npz"))
    g = EmpCylGrid(mmax=int(z["mmax"]), norder=int(z["norder"]), numx=int(z["numx"]),
                   numy=int(z["numy"]), cmapr=int(z["cmapr"]), cmapz=int(z["cmapz"]),
                   ascale=float(z["ascale"]), hscale=float(z["hscale"]), rmin=float(z["rmin"]),
                   rmax=float(z["rmax"]), rtable=float(z["rtable"]), xmin=float(z["xmin"]),
                   xmax=float(z["xmax"]), dx=float(z["dx"]), ymin=float(z["ymin"]),
                   ymax=float(z["ymax"]), dy=float(z["dy"]), tab=np.ascontiguousarray(z["tab"]))
    return g, z
