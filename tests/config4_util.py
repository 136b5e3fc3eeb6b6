"""BASELINE config 4 at test size (disk + halo, SphericalSL + EmpCylSL, multistep 4, both self forces
and both cross forces): inputs, the oracle run and the frozen golden vector
tests/golden/config4_small.npz (written by tests/golden/make_golden.py config4)."""
import os

import numpy as np

from tests.golden_util import HERE, load_cyl, load_sph
from tests.oracle_lib import NBodyOracle

# dynfracD, V, S, A, P (src/global.cc:76-80 defaults); dtime puts ~4-5 levels under both components
DYN = [1000.0, 0.01, 0.01, 0.03, 0.05]
MULTISTEP = 4
DTIME = 5.0e-4
NSTEPS = 2


def config4_inputs(n_halo=600, n_disk=600):
    """The disk sits in a Plummer halo of scale 3 a (so that the orbital frequency falls by ~8 across
    the disk and the disk populates four levels), carries a fiftieth of its mass and moves on
    near-circular orbits in the halo's monopole field; the halo reaches far beyond the cylinder's
    table (monopole branch of the cross force), the disk lies inside the halo's window."""
    from exp_amd.models import PlummerModel, sample_disk, sample_sphere
    cg, _ = load_cyl()
    model = PlummerModel(1.0, 1.0, 1e-3, 50.0)
    m, pos, vel = sample_sphere(model, n_halo, seed=41, rlim=45.0)
    pos[:, 2] *= 0.8
    sc = 3.0 * cg.ascale
    pos = pos * sc
    vel = vel / np.sqrt(sc)
    dm, dpos, dvel = sample_disk(n_disk, 42, a=cg.ascale, h=cg.hscale, mass=0.02,
                                 vcirc=lambda R: R / (R * R + sc * sc) ** 0.75)
    dpos[:, 0] *= 1.15
    return dict(scale=sc, halo_mass=m, halo_pos=pos, halo_vel=vel, disk_mass=dm, disk_pos=dpos,
                disk_vel=dvel)


def grids():
    g, _ = load_sph()            # lmax 4, nmax 6, numr 200 Plummer tables
    cg, _ = load_cyl()           # mmax 2, nmax 3, 16 x 8 EmpCylSL tables
    return g, cg


def sph_window(g, sc):
    return dict(scale=sc, rmin=g.rmin * sc, rmax=g.rmax * sc)


def oracle_run(oracle, inp, nsteps=NSTEPS, multistep=MULTISTEP, dtime=DTIME, dyn=DYN, pass0_only=False):
    g, cg = grids()
    sc = float(inp["scale"])
    prm = oracle.params(**sph_window(g, sc))
    nb = NBodyOracle(oracle, multistep, dtime, dyn)
    i1 = nb.add_sphere(g, prm, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
    i2 = nb.add_cylinder(cg, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
    nb.add_interaction(i1, i2)
    nb.add_interaction(i2, i1)
    nb.init(pass0_only=pass0_only)
    nsw = [0, 0]
    if not pass0_only:
        for _ in range(nsteps):
            nsw = [a + b for a, b in zip(nsw, nb.step())]
    return nb, nsw


def snapshot(nb, nsw):
    out = {"nswitch": np.array(nsw), "disk_cylmass": nb.cylmass(1), "halo_used": nb.used(0),
           "disk_used": nb.used(1)}
    for k, name in enumerate(("halo", "disk")):
        s = nb.state[k]
        out[name + "_level"] = s["level"].copy()
        for key in ("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "pot", "coefN", "coefL", "coef"):
            out[f"{name}_{key}"] = s[key].copy()
    return out


def load_golden():
    return np.load(os.path.join(HERE, "config4_small.npz"))
