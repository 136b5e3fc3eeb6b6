#!/usr/bin/env python3
"""tests/golden/adaptor_case.bin : the input + expected output of tests/cpp/test_potaccel.cpp (the
g++-built test that drives the C++ adaptor include/exp_amd_potaccel.hpp with no Python in the process):
the tables / particles of sph_small.npz, the oracle's coefficients, accelerations and one KDK step
(already frozen there), plus begin_run and one multistep-2 master step of the same particles from the
oracle (oracle/bfe_oracle.c).  Little-endian: 8-byte magic, int32[6], float64 scalars and arrays in
the order written below.

    python tests/golden/make_adaptor_case.py"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden_util import load_sph          # noqa: E402
from tests.oracle_lib import Oracle             # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MS, DTIME, DYN = 2, 0.05, [1000.0, 0.01, 0.01, 0.03, 0.05]


def main():
    orc = Oracle()
    g, z = load_sph()
    prm = orc.params(scale=1.0, rmin=g.rmin, rmax=g.rmax)
    n = len(z["mass"])
    st = orc.sph_multistep_init(g, prm, MS, DTIME, DYN, 0, z["pos"], z["vel"], z["mass"])
    nsw = orc.sph_multistep_step(g, prm, st)
    with open(os.path.join(HERE, "adaptor_case.bin"), "wb") as f:
        f.write(b"EXPAMD01")
        f.write(struct.pack("<6i", g.lmax, g.nmax, g.numr, g.cmap, n, MS))
        f.write(struct.pack("<8d", g.rmap, 1.0, g.rmin, g.rmax, g.xmin, g.dxi, float(z["dt"]), DTIME))
        f.write(struct.pack("<5d", *DYN))
        f.write(struct.pack("<2q", nsw, int(z["used"])))
        for a in (g.xi, g.p0, g.ev, g.ef, z["mass"], z["pos"], z["vel"], z["coef"], z["acc"], z["pot"],
                  z["step_pos"], z["step_vel"], z["step_acc"], z["step_pot"], z["step_coef"]):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        f.write(np.ascontiguousarray(st["level"], dtype="<i4").tobytes())
        for a in (np.stack([st["x"], st["y"], st["z"]], 1), np.stack([st["vx"], st["vy"], st["vz"]], 1),
                  np.stack([st["ax"], st["ay"], st["az"]], 1), st["coef"]):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
    print("wrote adaptor_case.bin", os.path.getsize(os.path.join(HERE, "adaptor_case.bin")), "bytes; levels",
          np.bincount(st["level"], minlength=MS + 1), "switches", nsw)


# ---- two components through the adaptor: cylinder + sphere, both cross forces (SetExternal), and the keys that default to
# off (rtrunc / com0, ton / toff / twid, FIX_L0, mlim, self_consistent) ---------------------------------------------------
MS2, DTIME2 = 2, 1.5e-4
SCENARIOS = [
    # (halo options, disk options): tests/oracle_lib.py NBodyOracle.set_options
    ({}, {}),
    (dict(rtrunc=0.35, com0=(0.004, -0.003, 0.002), fix_l0=True), dict(adiabatic=(0.5 * DTIME2, 1.0e20, DTIME2), mlim=1)),
    (dict(self_consistent=False), dict(rtrunc=0.03, self_consistent=False)),
]


def _opt_record(o):
    """[has_rtrunc, rtrunc, com0 x3, adiabatic, ton, toff, twid, self_consistent, fix_l0, mlim]"""
    ad = o.get("adiabatic")
    return [1.0 if o.get("rtrunc") is not None else 0.0, float(o.get("rtrunc") or 1e20), *[float(v) for v in o.get("com0", (0, 0, 0))],
            1.0 if ad else 0.0, *(ad if ad else (-1e20, 1e20, 0.1)), 1.0 if o.get("self_consistent", True) else 0.0,
            1.0 if o.get("fix_l0") else 0.0, float(o["mlim"]) if o.get("mlim") is not None else -1.0]


def main2():
    from tests import config4_util as c4
    from tests.oracle_lib import NBodyOracle
    orc = Oracle()
    g, cg = c4.grids()
    inp = c4.config4_inputs(n_halo=400, n_disk=400)
    sc = float(inp["scale"])
    win = c4.sph_window(g, sc)
    prm = orc.params(**win)
    path = os.path.join(HERE, "adaptor_case2.bin")
    with open(path, "wb") as f:
        f.write(b"EXPAMD02")
        f.write(struct.pack("<12i", g.lmax, g.nmax, g.numr, g.cmap, cg.mmax, cg.norder, cg.numx, cg.numy, cg.cmapr,
                            cg.cmapz, MS2, len(SCENARIOS)))
        f.write(struct.pack("<2i", len(inp["halo_mass"]), len(inp["disk_mass"])))
        f.write(struct.pack("<6d", g.rmap, win["scale"], win["rmin"], win["rmax"], g.xmin, g.dxi))
        f.write(struct.pack("<8d", cg.ascale, cg.hscale, cg.rtable, cg.xmin, cg.dx, cg.ymin, cg.dy, cg.rmax))
        f.write(struct.pack("<6d", DTIME2, *DYN))
        for a in (g.xi, g.p0, g.ev, g.ef, cg.tab, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"], inp["disk_mass"],
                  inp["disk_pos"], inp["disk_vel"]):
            f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
        for oh, od in SCENARIOS:
            nb = NBodyOracle(orc, MS2, DTIME2, DYN)
            i1 = nb.add_sphere(g, prm, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
            i2 = nb.add_cylinder(cg, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
            nb.add_interaction(i1, i2)
            nb.add_interaction(i2, i1)
            if oh:
                nb.set_options(i1, **oh)
            if od:
                nb.set_options(i2, **od)
            nb.init()
            init = [{k: nb.state[c][k].copy() for k in ("ax", "ay", "az", "pot", "coef", "level")} for c in (0, 1)]
            nsw = nb.step()
            f.write(np.array(_opt_record(oh) + _opt_record(od), dtype="<f8").tobytes())
            f.write(struct.pack("<2q", *nsw))
            f.write(struct.pack("<1d", nb.cylmass(1)))
            for c in (0, 1):
                st = nb.state[c]
                f.write(np.ascontiguousarray(init[c]["level"], dtype="<i4").tobytes())
                f.write(np.ascontiguousarray(st["level"], dtype="<i4").tobytes())
                for a in (np.stack([init[c]["ax"], init[c]["ay"], init[c]["az"]], 1), init[c]["pot"], init[c]["coef"],
                          np.stack([st["x"], st["y"], st["z"]], 1), np.stack([st["vx"], st["vy"], st["vz"]], 1),
                          np.stack([st["ax"], st["ay"], st["az"]], 1), st["pot"], st["coef"]):
                    f.write(np.ascontiguousarray(a, dtype="<f8").tobytes())
            frozen = [int((np.linalg.norm(np.stack([nb.state[c]["x"], nb.state[c]["y"], nb.state[c]["z"]], 1) -
                                          np.array(o.get("com0", (0, 0, 0))), axis=1) > o["rtrunc"]).sum()) if o.get("rtrunc") else 0
                      for c, o in ((0, oh), (1, od))]
            print("scenario", oh, od, "-> switches", nsw, "levels halo", np.bincount(nb.state[0]["level"], minlength=MS2 + 1),
                  "disk", np.bincount(nb.state[1]["level"], minlength=MS2 + 1), "frozen now", frozen, "cylmass", nb.cylmass(1))
    print("wrote adaptor_case2.bin", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
    main2()
