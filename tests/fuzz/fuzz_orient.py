"""Randomised campaign for the centre / orientation estimator (GPU): exp_amd/csrc/orient.hip against the oracle's
restatement of src/Orient.cc over random particle sets (sizes from 1 to 1e5, ties and signed zeros in the binding
energies), `keep`, `many`, flags, the kinetic-energy mode, spacing and damping, several calls in a row with the
component moved in between.  The energy threshold must be the same double, the count the same integer, centre / axis /
rotations equal to round-off.    python tests/fuzz/fuzz_orient.py [trials=60] [seed=1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from exp_amd.runtime import Component, Context, Orient
from tests.oracle_lib import Oracle

_argv = sys.argv if __name__ == "__main__" else [""]   # imported by tests/test_fuzz_gpu.py: defaults
trials = int(_argv[1]) if len(_argv) > 1 else 60
seed0 = int(_argv[2]) if len(_argv) > 2 else 1
orc = Oracle()
ctx = Context(0)


def one(t, rng):
    n = int(rng.choice([1, 2, 63, 64, 65, 777, 5000, 100000]))
    keep = int(rng.integers(1, 6))
    many = int(rng.choice([1, 5, max(1, n // 3), n - 1 if n > 1 else 1, n, 2 * n + 3]))
    flags = int(rng.choice([Orient.AXIS, Orient.CENTER, Orient.AXIS | Orient.CENTER]))
    cfl = Orient.KE if rng.random() < 0.4 else 0
    dT = float(rng.choice([0.0, 0.0, 0.05]))
    damp = float(rng.choice([1.0, 0.7, 0.3]))
    m = rng.uniform(0.5, 1.5, n) / n
    pos = rng.standard_normal((n, 3)) * rng.uniform(0.2, 3.0) + rng.normal(0, 0.3, 3)
    vel = rng.standard_normal((n, 3)) * 0.5 + 0.3 * np.cross([0.1, 0.2, 1.0], pos)
    pot = -1.0 / np.sqrt(0.1 + (pos ** 2).sum(axis=1))
    if n > 20 and rng.random() < 0.5:
        # +0 / -0 and exact ties AT THE TOP (both sides exclude the threshold energy itself).  Exact ties BELOW the threshold
        # are a documented deviation: the reference's std::set keeps one particle per energy and its
        # trimming turns order-dependent; FUZZ_ORIENT_TIES=1 plants them to see it.
        pot[::7] = 0.0
        pot[7::14] = -0.0
        if os.environ.get("FUZZ_ORIENT_TIES"):
            pot[3::11] = pot[3]
    c = Component.from_arrays(ctx, m, pos, vel)
    c.upload_acc(np.zeros((n, 3)), pot)
    o = Orient(ctx, keep, many, flags, cfl, dT=dT, damping=damp)
    ref = orc.orient(keep, many, flags, cfl, dT, damp)
    status, detail = "ok", ""
    calls = int(rng.integers(1, keep + 4))
    for k in range(calls):
        tnow = 0.03 * k
        o.accumulate(tnow, c, 0.03)
        d = c.download(("mass", "pos", "vel", "pot"))
        orc.orient_accumulate(ref, tnow, 0.03, d["mass"], d["pos"], d["vel"], d["pot"])
        st = o.state()
        bad = []
        if st["Ecurr"] != ref.Ecurr and not (np.isnan(st["Ecurr"]) and np.isnan(ref.Ecurr)):
            bad.append(f"Ecurr {st['Ecurr']!r} vs {ref.Ecurr!r}")
        if st["used"] != ref.used:
            bad.append(f"used {st['used']} vs {ref.used}")
        for key in ("center", "axis", "axis1", "center1", "center0"):
            r = np.array(getattr(ref, key)[:])
            if not np.allclose(st[key], r, rtol=0, atol=1e-11 * max(1.0, np.abs(r).max()), equal_nan=True):
                bad.append(f"{key} {st[key]} vs {r}")
        # (an axis that is rounding noise around zero -- a one-entry history of one particle -- has no direction to compare)
        if np.linalg.norm(np.array(ref.axis[:])) > 1e-10 * max(1.0, np.abs(np.array(ref.axis1[:])).max()) and \
                not np.allclose(st["body"], np.array(ref.body[:]).reshape(3, 3), rtol=0, atol=1e-9, equal_nan=True):
            bad.append(f"body {np.round(st['body'].ravel(), 6).tolist()} vs {np.round(np.array(ref.body[:]), 6).tolist()} axis "
                       f"{st['axis'].tolist()} vs {list(ref.axis[:])}")
        if bad:
            status, detail = "MISMATCH", f"call {k}: " + "; ".join(bad)
            break
        # move the component rigidly and let it rotate a little before the next call
        c.incr_position(0.03)
    print(f"{t:3d} n {n} keep {keep} many {many} flags {flags} ke {bool(cfl)} dT {dT} damp {damp} calls {calls}: {status} {detail}", flush=True)
    o.close(); c.close()
    return status == "ok"


def main():
    t0 = time.time()
    bad = sum(0 if one(t, np.random.default_rng([seed0, t])) else 1 for t in range(trials))
    print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
