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


if __name__ == "__main__":
    main()
