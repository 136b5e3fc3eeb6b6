"""Randomised campaign for the pyEXP.basis surface (GPU): `Basis.factory` configurations of both bases with random
orders, grids, maps, windows (N1 / N2), flags and vertical-parity splits; `createFromArray` with a random centre, rotation,
array layout (posvelrows, extra velocity columns) against the literal pyEXP twins of the oracle (accumulate at the
transformed positions); `getAccel` and `getFields` at points inside and far outside the tables; `set_coefs` round trip of
the returned structure.    python tests/fuzz/fuzz_pyexp.py [trials=40] [seed=1]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from exp_amd.basis import Basis
from tests.oracle_lib import Oracle

_argv = sys.argv if __name__ == "__main__" else [""]   # imported by tests/test_fuzz_gpu.py: defaults
trials = int(_argv[1]) if len(_argv) > 1 else 40
seed0 = int(_argv[2]) if len(_argv) > 2 else 1
orc = Oracle()
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
tmp = tempfile.mkdtemp(prefix="fuzz_pyexp_")
_bases = {}


def rotation(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return q * np.sign(np.linalg.det(q))


def layout(rng, pos):
    """the array createFromArray is handed: [N, 3], [3, N] (posvelrows False) or [N, 6] / [6, N] with velocities"""
    vel = rng.standard_normal(pos.shape)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        return pos.T.copy(), False                      # 3 x N: pyEXP's default layout (columns are particles)
    if kind == 1:
        return pos.copy(), True                         # N x 3 with posvelrows
    if kind == 2:
        return np.concatenate([pos, vel], axis=1).T.copy(), False
    return np.concatenate([pos, vel], axis=1), True


def sph_basis(rng):
    Lmax, nmax = int(rng.integers(0, 7)), int(rng.integers(2, 13))
    numr, cmap = int(rng.choice([400, 1000])), int(rng.choice([1, 1, 2]))
    key = ("sph", Lmax, nmax, numr, cmap)
    if key not in _bases:
        _bases[key] = f"""
id : sphereSL
parameters :
  numr: {numr}
  rmin: 0.0001
  rmax: 1.95
  Lmax: {Lmax}
  nmax: {nmax}
  cmap: {cmap}
  rmapping : 0.0667
  modelname: {os.path.join(GOLD, 'SLGridSph.model')}
  cachename: {os.path.join(tmp, 'sl.' + '_'.join(map(str, key[1:])))}
"""
    extra = ""
    flags = {}
    for k in ("NO_L0", "NO_L1", "EVEN_L", "EVEN_M", "M0_ONLY"):
        if rng.random() < 0.15:
            extra += f"  {k}: true\n"
            flags["M0_only" if k == "M0_ONLY" else k] = True
    n1 = n2 = None
    if rng.random() < 0.3:
        n1, n2 = bool(rng.random() < 0.5), bool(rng.random() < 0.5)
        extra += f"  N1: {str(n1).lower()}\n  N2: {str(n2).lower()}\n"
    return key, Basis.factory(_bases[key] + extra), flags, (n1, n2)


def cyl_basis(rng):
    mmax, nmax = int(rng.integers(0, 5)), int(rng.integers(1, 7))
    nodd = int(rng.integers(0, nmax + 2))
    key = ("cyl", mmax, nmax, nodd, int(rng.choice([24, 40])))
    cfg = f"""
id: cylinder
parameters:
  acyl: 0.01
  hcyl: 0.001
  mmax: {mmax}
  nmax: {nmax}
  ncylodd: {nodd}
  ncylnx: {key[4]}
  ncylny: {key[4] // 2}
  ncylr: 400
  lmaxfid: {max(10, mmax + 6)}
  nmaxfid: 8
  rnum: 30
  tnum: 20
  cachename: {os.path.join(tmp, 'eof.' + '_'.join(map(str, key[1:])))}
"""
    return key, Basis.factory(cfg)


def field_err(got, ref, floors=(0.0, 0.0, 0.0)):
    """worst error of the density, potential and force columns, each against the largest value of its group (or the floor
    given for it: what a unit coefficient would produce, for sets whose coefficients vanish by symmetry)"""
    e = 0.0
    for (a, b), floor in zip(((0, 3), (3, 6), (6, 9)), floors):
        sc = max(np.abs(ref[:, a:b]).max(), floor)
        if sc > 1e-250:
            e = max(e, np.abs(got[:, a:b] - ref[:, a:b]).max() / sc)
    return e


def trial_sph(t, rng):
    key, basis, flags, (n1, n2) = sph_basis(rng)
    n = int(rng.choice([1, 64, 1000, 8000]))
    pos = rng.normal(0, 0.25, (n, 3)) * np.array([1.0, 1.0, rng.uniform(0.3, 1.2)])
    pos[: min(n, 4)] = np.array([[0, 0, 0], [0, 0, 0.3], [5.0, 0, 0], [1e-5, 1e-5, -2e-5]])[: min(n, 4)]
    m = rng.uniform(0.5, 1.5, n) / n
    ctr, rot = (rng.normal(0, 0.1, 3), rotation(rng)) if rng.random() < 0.5 else (np.zeros(3), np.eye(3))
    if ctr.any():
        # (through a frame, what the accumulation sees of a particle ON the axis is rounding dust -- or a zero whose sign,
        # the azimuth atan2(+-0, +-0), depends on the signs of the rotation's entries: not a property of this code)
        pos[: min(n, 2)] += np.array([3e-4, 1e-4, 0.0])
    world = pos @ rot + ctr                                  # what the caller holds: accumulate sees (world - ctr) rot^T = pos
    arr, pvr = layout(rng, world)
    cs = basis.createFromArray(m, arr, time=0.25, center=ctr, rot=rot, posvelrows=pvr)
    prm = orc.params(scale=1.0, rmin=basis.rmin, rmax=basis.rmax, N1=0 if n1 is None else int(n1),
                     N2=-1 if n2 is None else int(n2), **flags)
    c_ref, used_ref = orc.pyexp_sph_accumulate(basis.grid, prm, pos, m)
    got = basis.expcoef.copy()
    cmax = max(np.abs(c_ref).max(), 1e-300)
    e_c = np.abs(got - c_ref).max() / cmax
    # (a ladder of polar angles from 1e-9 to 1e-2 about either pole: the reference's arithmetic is ill-conditioned there
    # and parity is with its values -- see tests/test_sph_gpu.py::test_polar_axis_lanes)
    lad = []
    for _ in range(10):
        th, ph, rr = 10.0 ** rng.uniform(-9, -2), rng.uniform(0, 2 * np.pi), np.exp(rng.uniform(np.log(0.02), np.log(3.0)))
        lad.append([rr * np.sin(th) * np.cos(ph), rr * np.sin(th) * np.sin(ph), rng.choice([-1.0, 1.0]) * rr * np.cos(th)])
    test = np.concatenate([rng.normal(0, 0.4, (200, 3)), np.array([[6.0, 1.0, -2.0], [0.0, 30.0, 4.0], [2e-5, 1e-5, 3e-5]]),
                           np.array(lad)])
    basis.set_coefs(cs)
    a_ref = orc.pyexp_sph_accel(basis.grid, prm, basis.force.get_coefs(), test)
    acc = basis.getAccel(test)
    fin = np.isfinite(a_ref).all(axis=1)
    asc_ = max(np.linalg.norm(a_ref[fin], axis=1).max(), 1e-300)
    e_a = np.abs(acc[fin] - a_ref[fin]).max() / asc_
    # (every point against its own acceleration, floored at 1e-2 of the largest)
    e_a = max(e_a, 0.1 * (np.linalg.norm(acc[fin] - a_ref[fin], axis=1) / np.maximum(np.linalg.norm(a_ref[fin], axis=1), 1e-2 * asc_)).max())
    same = np.array_equal(np.isfinite(acc).all(axis=1), fin)
    x, y, z = test.T
    with np.errstate(all="ignore"):
        fr = orc.sph_fields(basis.grid, prm, basis.force.get_coefs(), x, y, z, "cartesian")
    fg = basis.getFields(x, y, z)
    ff = np.isfinite(fr).all(axis=1)
    e_f = field_err(fg[ff], fr[ff])
    ok = (e_c <= 1e-10 and e_a <= 1e-9 and e_f <= 1e-9 and same and cs.time == 0.25 and np.array_equal(cs.ctr, ctr)
          and np.array_equal(cs.rot, rot))
    print(f"sph {t:3d} {key[1:]} flags {list(flags)} N1N2 {(n1, n2)} n {n} layout {arr.shape} pvr {pvr} ctr {bool(np.any(ctr))}: "
          f"coef {e_c:.1e} acc {e_a:.1e} fields {e_f:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    return ok


def trial_cyl(t, rng):
    key, basis = cyl_basis(rng)
    n = int(rng.choice([1, 64, 1000, 8000]))
    A, H = basis.acyl, basis.hcyl
    R = -A * np.log(rng.random(n) * rng.random(n))
    ph = rng.uniform(0, 2 * np.pi, n)
    pos = np.stack([R * np.cos(ph), R * np.sin(ph), 2 * H * np.arctanh(rng.uniform(-0.99, 0.99, n))], axis=1)
    pos[: min(n, 3)] = np.array([[0, 0, 0], [A, 0, 0], [50 * A, 0, 0.0]])[: min(n, 3)]
    m = rng.uniform(0.5, 1.5, n) / n
    ctr, rot = (rng.normal(0, 0.003, 3), rotation(rng)) if rng.random() < 0.5 else (np.zeros(3), np.eye(3))
    if ctr.any():
        pos[0] += np.array([3e-3 * A, 1e-3 * A, 0.0])           # (see trial_sph: nothing exactly on the axis through a frame)
    world = pos @ rot + ctr
    arr, pvr = layout(rng, world)
    cs = basis.createFromArray(m, arr, time=1.5, center=ctr, rot=rot, posvelrows=pvr)
    cc, ss, _ = orc.pyexp_cyl_accumulate(basis.grid, pos, m)
    floor = 1e-3 * m.sum() * np.abs(basis.grid.tab[0]).max()
    cmax = max(np.abs(cc).max(), floor)
    e_c = max(np.abs(cs.coefs.real - cc).max(), np.abs(cs.coefs.imag - ss).max()) / cmax
    Rt = basis.grid.rtable * A
    # (a ladder of distances from the axis, 1e-12 a to 1e-3 a: the force is projected with 1/R and 1/R^2 there)
    lad = []
    for _ in range(10):
        Rr, ph = A * 10.0 ** rng.uniform(-12, -3), rng.uniform(0, 2 * np.pi)
        lad.append([Rr * np.cos(ph), Rr * np.sin(ph), H * rng.normal(0, 1.5)])
    test = np.concatenate([np.random.default_rng(t).normal(0, 0.03, (200, 3)) * np.array([1, 1, 0.1]),
                           np.array([[1.2 * Rt, 0, 0], [0, 0.9 * Rt, 0.3 * Rt], [1e-8, 0, 0.001]]), np.array(lad)])
    basis.set_coefs(cs)
    a_ref = orc.pyexp_cyl_accel(basis.grid, cc, ss, test)
    acc = basis.getAccel(test)
    fin = np.isfinite(a_ref).all(axis=1)
    asc = max(np.linalg.norm(a_ref[fin], axis=1).max(), 1e-3 * m.sum() * np.abs(basis.grid.tab[1]).max())
    e_a = np.abs(acc[fin] - a_ref[fin]).max() / asc
    if n >= 64:     # (a lone particle's field near itself is what symmetry leaves of its terms: no scale of its own)
        e_a = max(e_a, 0.1 * (np.linalg.norm(acc[fin] - a_ref[fin], axis=1) / np.maximum(np.linalg.norm(a_ref[fin], axis=1), 1e-2 * asc)).max())
    same = np.array_equal(np.isfinite(acc).all(axis=1), fin)
    x, y, z = test.T
    with np.errstate(all="ignore"):
        fr = orc.cyl_fields(basis.grid, cc, ss, x, y, z, "cartesian")
    fg = basis.getFields(x, y, z)
    ff = np.isfinite(fr).all(axis=1)
    unit = max(np.abs(cc).max(), np.abs(ss).max(), floor / max(np.abs(basis.grid.tab[0]).max(), 1e-300))
    e_f = field_err(fg[ff], fr[ff], (unit * np.abs(basis.grid.dens).max(), unit * np.abs(basis.grid.tab[0]).max(),
                                     unit * max(np.abs(basis.grid.tab[1]).max(), np.abs(basis.grid.tab[2]).max())))
    if os.environ.get("FUZZ_VERBOSE") or e_f > 1e-9:
        for a, b in ((0, 3), (3, 6), (6, 9)):
            d = np.abs(fg[ff][:, a:b] - fr[ff][:, a:b])
            k = np.unravel_index(np.argmax(d), d.shape)
            print(f"    cols {a}:{b} worst |d| {d.max():.3e} at point {test[ff][k[0]]} col {a + k[1]} ref {fr[ff][k[0], a + k[1]]:.6e} "
                  f"got {fg[ff][k[0], a + k[1]]:.6e} group max {np.abs(fr[ff][:, a:b]).max():.3e}")
    ok = e_c <= 1e-10 and e_a <= 1e-9 and e_f <= 1e-9 and same and cs.time == 1.5
    print(f"cyl {t:3d} {key[1:]} n {n} layout {arr.shape} pvr {pvr} ctr {bool(np.any(ctr))}: coef {e_c:.1e} acc {e_a:.1e} "
          f"fields {e_f:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
    return ok


def main():
    t0 = time.time()
    bad = 0
    for t in range(trials):
        for kind, fn in (("sph", trial_sph), ("cyl", trial_cyl)):
            if not fn(t, np.random.default_rng([seed0, t, 0 if kind == "sph" else 1])):
                bad += 1
    print(f"{trials} trials of each basis, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
