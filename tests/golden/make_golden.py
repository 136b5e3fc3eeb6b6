#!/usr/bin/env python3
"""Generates the golden input/output vectors under tests/golden/ (run from the repo root):

    python tests/golden/make_golden.py

sph_small.npz : a small SLGridSph (Plummer, lmax 4, nmax 6, numr 200) + 300 particles (incl. edge
                cases) + the CPU oracle's coefficients, accelerations, potentials and one KDK step.
cyl_small.npz : a small EmpCylSL table set (mmax 2, nmax 3, 16x8 grid) + 300 disk particles + the
                oracle's cos/sin coefficients, cylmass, accelerations and potentials.

The reference cannot be built or imported in this image, so these vectors
come from the oracle (oracle/bfe_oracle.c, oracle/cyl_oracle.c), whose correspondence with the
reference is argued line by line there and pinned by tests/test_oracle_kat.py.  They freeze the
oracle's answers so that any later change to oracle OR device code that moves a result is caught.
sph_fields.npz: see make_fields() (python tests/golden/make_golden.py fields).
extras.npz    : see make_extras() (python tests/golden/make_golden.py extras).
config4_small.npz: see make_config4() (python tests/golden/make_golden.py config4).
SLGridSph.model is the reference's own data file (tests/Halo/SLGridSph.model), copied verbatim.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from exp_amd.empcyl import build_empcyl          # noqa: E402
from exp_amd.models import PlummerModel, sample_disk, sample_sphere   # noqa: E402
from exp_amd.slgrid import build_slgrid          # noqa: E402
from tests.oracle_lib import Oracle              # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    orc = Oracle()
    # ---- spherical -------------------------------------------------------------------------
    model = PlummerModel(1.0, 1.0, 1e-3, 50.0)
    g = build_slgrid(model, 4, 6, numr=200, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0, nel=24, P=8)
    m, pos, vel = sample_sphere(model, 290, seed=20261002)
    pos[:, 2] *= 0.7
    pos[:, 0] += 0.05
    extra = np.array([[0, 0, 0], [0, 0, 0.5], [60.0, 1, 2], [0, 70.0, 0], [1e-4, 0, 0],
                      [g.rmax * (1 - 1e-12), 0, 0], [0, 2e-3, 0], [-3, 0, 0], [0, -3, 0],
                      [0.1, 0.1, -40.0]])
    pos = np.concatenate([pos, extra])
    vel = np.concatenate([vel, np.zeros((len(extra), 3))])
    m = np.concatenate([m, np.full(len(extra), m[0])])
    prm = orc.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    coef, used = orc.sph_accumulate(g, prm, pos, m)
    acc, pot = orc.sph_accel(g, prm, pos, coef)
    p1, v1, a1, pt1, c1 = orc.sph_step(g, prm, 0.01, pos, vel, acc, m)
    np.savez_compressed(os.path.join(HERE, "sph_small.npz"),
                        lmax=g.lmax, nmax=g.nmax, numr=g.numr, cmap=g.cmap, rmin=g.rmin, rmax=g.rmax,
                        rmap=g.rmap, xmin=g.xmin, xmax=g.xmax, dxi=g.dxi, xi=g.xi, r=g.r, p0=g.p0,
                        d0=g.d0, ev=g.ev, ef=g.ef, mass=m, pos=pos, vel=vel, coef=coef, used=used,
                        acc=acc, pot=pot, dt=0.01, step_pos=p1, step_vel=v1, step_acc=a1,
                        step_pot=pt1, step_coef=c1)
    # ---- cylindrical -----------------------------------------------------------------------
    cg = build_empcyl(mmax=2, norder=3, numx=16, numy=8, lmaxfid=8, nmaxfid=6, numr=300, rnum=40,
                      tnum=20)
    cm, cpos, _ = sample_disk(292, 777, a=cg.ascale, h=cg.hscale)
    cpos[:, 0] *= 1.2
    Rt = cg.rtable * cg.ascale
    cextra = np.array([[1.2 * Rt, 0, 0], [0, 0.9 * Rt, 0.05 * Rt], [0.8 * Rt, 0.1 * Rt, 0],
                       [1e-5 * cg.ascale, 2e-5 * cg.ascale, 0], [0.999 * Rt, 0, 0],
                       [25 * cg.ascale, 0, 0], [0, 0.5 * cg.ascale, 0.5 * Rt],
                       [-2 * cg.ascale, 0, -3 * cg.hscale]])
    cpos = np.concatenate([cpos, cextra])
    cm = np.concatenate([cm, np.full(len(cextra), cm[0])])
    cc, ss, cused, cmass = orc.cyl_accumulate(cg, cpos, cm)
    cacc, cpot = orc.cyl_accel(cg, cpos, cc, ss, cmass)
    np.savez_compressed(os.path.join(HERE, "cyl_small.npz"),
                        mmax=cg.mmax, norder=cg.norder, numx=cg.numx, numy=cg.numy, cmapr=cg.cmapr,
                        cmapz=cg.cmapz, ascale=cg.ascale, hscale=cg.hscale, rmin=cg.rmin,
                        rmax=cg.rmax, rtable=cg.rtable, xmin=cg.xmin, xmax=cg.xmax, dx=cg.dx,
                        ymin=cg.ymin, ymax=cg.ymax, dy=cg.dy, tab=cg.tab, mass=cm, pos=cpos,
                        cos=cc, sin=ss, used=cused, cylmass=cmass, acc=cacc, pot=cpot)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])


def make_fields():
    """sph_fields.npz : pyEXP field evaluation (getFields) at 60 points in the three coordinate
    systems + a fix_positions vector, for the grid / coefficients / particles frozen in
    sph_small.npz (which is NOT regenerated here)."""
    from tests.golden_util import load_sph
    orc = Oracle()
    g, z = load_sph()
    prm = orc.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    rng = np.random.default_rng(4242)
    pts = rng.standard_normal((60, 3)) * np.array([2.0, 1.5, 0.8])
    pts[:4] *= 30.0                                   # beyond rmax: pyEXP extrapolates the tables
    x, y, zz = pts.T
    R, ph, r = np.hypot(x, y), np.arctan2(y, x), np.linalg.norm(pts, axis=1)
    crt = orc.sph_fields(g, prm, z["coef"], x, y, zz, "cartesian")
    cyl = orc.sph_fields(g, prm, z["coef"], R, zz, ph, "cylindrical")
    sph = orc.sph_fields(g, prm, z["coef"], r, zz / r, ph, "spherical")
    sums = np.zeros((1, 10))
    com = orc.fix_positions(z["mass"], z["pos"], z["vel"], z["acc"], np.zeros(len(z["mass"]), np.int32),
                            0, 0, sums)
    np.savez_compressed(os.path.join(HERE, "sph_fields.npz"), points=pts, crt=crt, cyl=cyl, sph=sph,
                        fix_positions=com)
    print("wrote sph_fields.npz")


def make_extras():
    """extras.npz : Orient (three calls: selection, histories, regression, rotations), the pseudo-
    acceleration fit, and the sub-sample covariances of both bases, for the particles / tables frozen
    in sph_small.npz and cyl_small.npz (NOT regenerated here)."""
    from tests.golden_util import load_cyl, load_sph
    orc = Oracle()
    g, z = load_sph()
    prm = orc.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    o = orc.orient(2, 120, 3, 2, 0.0, 0.8)
    rows, states = [], []
    pos, vel = z["pos"].copy(), z["vel"].copy()
    for k in range(4):
        orc.orient_accumulate(o, 0.1 * k, 0.1, z["mass"], pos, vel, z["pot"])
        states.append([o.Ecurr, o.used, *o.center[:], *o.axis[:], *o.center1[:], *o.axis1[:], o.sigC, o.sigA])
        rows.append([0.1 * k, *o.center1[:], *o.axis1[:]])
        pos = pos + 0.1 * vel
    acc, om, dom = orc.pseudo_accel_fit(np.array(rows))
    cov = orc.pyexp_sph_covariance(g, prm, z["pos"], z["mass"], 5)
    cg, cz = load_cyl()
    ccov = orc.cyl_covariance(cg, cz["pos"], cz["mass"], 4)
    np.savez_compressed(os.path.join(HERE, "extras.npz"), orient_states=np.array(states),
                        orient_body=np.array(o.body[:]), pseudo=np.concatenate([acc, om, dom]),
                        sph_cov_counts=cov["counts"], sph_cov_masses=cov["masses"], sph_cov_mean=cov["mean"],
                        sph_cov_covr=cov["covr"], cyl_cov_counts=ccov["counts"], cyl_cov_mean=ccov["mean"],
                        cyl_cov_covr=ccov["covr"])
    print("wrote extras.npz")


def make_config4():
    """config4_small.npz : BASELINE config 4 at test size -- inputs (600 + 600 bodies on the tables of
    sph_small.npz / cyl_small.npz, which are NOT regenerated here) and the n-body oracle's state after
    begin_run and after two multistep-4 master steps (oracle/nbody_oracle.c)."""
    from tests import config4_util as c4
    orc = Oracle()
    inp = c4.config4_inputs()
    nb0, _ = c4.oracle_run(orc, inp, nsteps=0)
    init = {"init_" + k: v for k, v in c4.snapshot(nb0, [0, 0]).items()}
    nb, nsw = c4.oracle_run(orc, inp)
    snap = c4.snapshot(nb, nsw)
    print("levels halo", np.bincount(snap["halo_level"], minlength=5), "disk",
          np.bincount(snap["disk_level"], minlength=5), "switches", nsw)
    np.savez_compressed(os.path.join(HERE, "config4_small.npz"), multistep=c4.MULTISTEP, dtime=c4.DTIME,
                        dynfrac=np.array(c4.DYN), nsteps=c4.NSTEPS, **inp, **init, **snap)
    print("wrote config4_small.npz")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "config4":
        make_config4()
    elif len(sys.argv) > 1 and sys.argv[1] == "fields":
        make_fields()
    elif len(sys.argv) > 1 and sys.argv[1] == "extras":
        make_extras()
    else:
        main()
        make_fields()
        make_extras()
        make_config4()
