"""Randomised campaign for the coefficient containers and their files (CPU): random spherical / cylindrical coefficient
sets (orders, number of times, times with more digits than the container's time key keeps, centre and rotation) through
every route -- HDF5 write + factory read, extension of an existing file, native stream write + factory read, stride and
time-window reads, deepcopy, setMatrix / getMatrix, Power -- with the values required back bit for bit (times rounded the
way the reference's map key rounds them) and Power recomputed from the definition.
    python tests/fuzz/fuzz_coefs.py [trials=200] [seed=1]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from exp_amd import coefs as C
from exp_amd.basis import CylStruct, SphStruct

_argv = sys.argv if __name__ == "__main__" else [""]   # imported by tests/test_fuzz_cpu.py: defaults
trials = int(_argv[1]) if len(_argv) > 1 else 200
seed0 = int(_argv[2]) if len(_argv) > 2 else 1
tmp = tempfile.mkdtemp(prefix="fuzz_coefs_")
UNITS = (("mass", "Msun", 1e10), ("length", "kpc", 1.0), ("time", "Myr", 1.0), ("G", "mixed", 4.3e-6))


def make(rng, sphere, order, nmax, t, scale):
    ctr = rng.normal(size=3) if rng.random() < 0.5 else np.zeros(3)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    rot = q if rng.random() < 0.5 else np.eye(3)
    if sphere:
        ldim = (order + 1) * (order + 2) // 2
        cf = rng.normal(size=(ldim, nmax)) + 1j * rng.normal(size=(ldim, nmax))
        k = 0
        for l in range(order + 1):            # m = 0 rows are real
            cf[k] = cf[k].real
            k += l + 1
        return SphStruct(order, nmax, scale, t, cf, ctr, rot)
    cf = rng.normal(size=(order + 1, nmax)) + 1j * rng.normal(size=(order + 1, nmax))
    cf[0] = cf[0].real
    return CylStruct(order, nmax, t, cf, ctr, rot)


def one(t, rng):
    sphere = bool(rng.random() < 0.5)
    order, nmax = int(rng.integers(0, 7)), int(rng.integers(1, 13))
    ntimes = int(rng.integers(1, 9))
    times = np.sort(rng.uniform(0, 10, ntimes) + rng.uniform(0, 1e-9, ntimes))
    scale = float(rng.choice([1.0, 0.05]))                   # one per container: the file holds one
    sets = [make(rng, sphere, order, nmax, float(tt), scale) for tt in times]
    bad = []
    cont = C.Coefs.makecoefs(sets[0], "comp")
    for s in sets:
        cont.add(s)
    keys = [C.round_time(float(tt)) for tt in times]
    if cont.Times() != sorted(set(keys)):
        bad.append("Times")
    for name, unit, value in UNITS:
        cont.setUnits(name, unit, value)
    path = os.path.join(tmp, f"c{t}.h5")
    first = int(rng.integers(1, ntimes + 1))
    part = C.Coefs.makecoefs(sets[0], "comp")
    for s in sets[:first]:
        part.add(s)
    for name, unit, value in UNITS:
        part.setUnits(name, unit, value)
    part.WriteH5Coefs(path)
    if first < ntimes:
        rest = C.Coefs.makecoefs(sets[first], "comp")
        for s in sets[first:]:
            rest.add(s)
        rest.ExtendH5Coefs(path)
    back = C.Coefs.factory(path)
    if type(back) is not type(cont) or back.getName() != "comp" or back.Times() != cont.Times():
        bad.append("h5 factory")
    else:
        for tt in back.Times():
            a, b = back.getCoefStruct(tt), cont.getCoefStruct(tt)
            if not (np.array_equal(a.coefs, b.coefs) and np.array_equal(np.asarray(a.ctr), np.asarray(b.ctr))
                    and np.array_equal(np.asarray(a.rot).reshape(-1), np.asarray(b.rot).reshape(-1))):
                bad.append(f"h5 values at {tt}")
                break
        if not np.array_equal(back.getAllCoefs(), cont.getAllCoefs()) or not cont.CompareStanzas(back):
            bad.append("h5 getAllCoefs / CompareStanzas")
        if [u[:2] for u in back.getUnits()] != [u[:2] for u in cont.getUnits()]:
            bad.append("units")
    stride = int(rng.integers(1, 4))
    lo, hi = sorted(rng.uniform(-1, 11, 2))
    sub = C.Coefs.factory(path, stride, lo, hi)
    # every stride-th snapshot OF THE FILE, then the window (expui/Coefficients.cc:257-300: `n += stride`, `continue` outside)
    want = [tt for tt in cont.Times()[::stride] if lo <= tt <= hi]
    if sub.Times() != want:
        bad.append(f"stride / window {sub.Times()} vs {want}")
    npath = os.path.join(tmp, f"c{t}.native")
    cont.writeNativeCoefs(npath)
    nat = C.Coefs.factory(npath)
    if nat.Times() != cont.Times() or not np.array_equal(nat.getAllCoefs(), cont.getAllCoefs()):
        bad.append("native")
    cp = cont.deepcopy()
    t0 = cont.Times()[0]
    mat = cont.getMatrix(t0).copy()
    cp.setMatrix(t0, mat * 2.0)
    if not np.array_equal(cont.getMatrix(t0), mat) or not np.array_equal(cp.getMatrix(t0), mat * 2.0):
        bad.append("deepcopy / setMatrix")
    allc = cont.getAllCoefs()                                   # [rows, n, times]
    p = cont.Power()
    if sphere:
        ref = np.zeros((len(cont.Times()), order + 1))
        k = 0
        for l in range(order + 1):
            for m in range(l + 1):
                ref[:, l] += (np.abs(allc[k]) ** 2).sum(axis=0)
                k += 1
    else:
        ref = (np.abs(allc) ** 2).sum(axis=1).T
    if p.shape != ref.shape or not np.allclose(p, ref, rtol=1e-13, atol=0):
        bad.append("Power")
    for f in (path, npath):
        os.remove(f)
    print(f"{t:3d} {'sph' if sphere else 'cyl'} order {order} nmax {nmax} times {ntimes} first {first} stride {stride}: "
          f"{'ok' if not bad else 'MISMATCH ' + '; '.join(bad)}", flush=True)
    return not bad


def main():
    t0 = time.time()
    bad = sum(0 if one(t, np.random.default_rng([seed0, t])) else 1 for t in range(trials))
    print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
