"""Randomised parity campaign (GPU): device against oracle over random basis orders, grids, maps, flags, scales, windows,
centres and ADVERSARIAL particle sets (origin, polar axis, exactly on the window edges, far outside, denormal offsets,
duplicates, zero masses, one particle, a handful) for both force methods, single level.  Every trial prints one line;
exit code 1 on the first mismatch with the seed that reproduces it.

    python tests/fuzz/fuzz_parity.py [trials=60] [seed=1] [sph|cyl|both]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from exp_amd.empcyl import build_empcyl
from exp_amd.models import NFWModel, PlummerModel, sample_sphere
from exp_amd.runtime import Component, Context, Cylinder, SphereSL
from exp_amd.slgrid import build_slgrid
from tests.oracle_lib import Oracle

_argv = sys.argv if __name__ == "__main__" else [""]   # imported by tests/test_fuzz_gpu.py: defaults
trials = int(_argv[1]) if len(_argv) > 1 else 60
seed0 = int(_argv[2]) if len(_argv) > 2 else 1
which = _argv[3] if len(_argv) > 3 else "both"
orc = Oracle()
ctx = Context(0)
COEF_TOL, ACC_TOL = 1e-10, 1e-9
_sl, _cy = {}, {}


def field_err(got, ref, floor=0.0):
    """density, potential and force columns each against the largest value of THEIR group over the points (a column that is
    identically zero -- the azimuthal force on the axis -- has no scale of its own); `floor`: one number or one per group"""
    e = 0.0
    floors = floor if isinstance(floor, tuple) else (floor, floor, floor)
    for (a, b), floor in zip(((0, 3), (3, 6), (6, 9)), floors):
        # (floor: what the particles COULD have contributed -- an all-antisymmetric basis and a particle in the plane leave
        # coefficients, and fields, that are rounding noise around zero on both sides)
        sc = max(np.abs(ref[:, a:b]).max(), floor)
        if sc < 1e-250:
            continue
        e = max(e, np.abs(got[:, a:b] - ref[:, a:b]).max() / sc)
    return e


def sl_grid(rng):
    kind = rng.choice(["plummer", "nfw"])
    lmax, nmax = int(rng.integers(0, 13)), int(rng.integers(1, 21))
    numr, cmap = int(rng.choice([100, 257, 400, 800, 1500])), int(rng.choice([1, 1, 2]))
    key = (kind, lmax, nmax, numr, cmap)
    if key not in _sl:
        model = PlummerModel(1.0, 1.0, 1e-3, 50.0) if kind == "plummer" else NFWModel(1.0, 20.0, 6.0, 1e-3, 50.0)
        _sl[key] = (model, build_slgrid(model, lmax, nmax, numr=numr, rmin=1e-3, rmax=49.5, cmap=cmap, rmap=1.0, nel=24, P=6))
    return key, _sl[key]


def cyl_grid(rng):
    mmax, norder = int(rng.integers(0, 8)), int(rng.integers(1, 9))
    numx, numy = int(rng.choice([16, 33, 48])), int(rng.choice([8, 17, 24]))
    cm = (int(rng.choice([1, 2])), int(rng.choice([1, 2, 3])))
    nodd = int(rng.integers(-1, norder + 1))
    key = (mmax, norder, numx, numy, cm, nodd)
    if key not in _cy:
        _cy[key] = build_empcyl(mmax=mmax, norder=norder, numx=numx, numy=numy, lmaxfid=max(10, mmax + 6), nmaxfid=8,
                                numr=300, rnum=30, tnum=20, cmapr=cm[0], cmapz=cm[1], nodd=None if nodd < 0 else nodd)
    return key, _cy[key]


def nasty_sphere(rng, model, g, scale, rmin, rmax, ctr):
    n = int(rng.choice([1, 2, 7, 63, 64, 65, 255, 1000, 5000, 20000]))
    m, pos, _ = sample_sphere(model, n, seed=int(rng.integers(1 << 30)))
    pos = pos * scale
    pos[:, 2] *= rng.uniform(0.2, 1.5)
    pos[:, 0] += rng.uniform(-0.2, 0.2) * scale
    m = m * rng.uniform(0.1, 3.0, n)
    # (near the polar axis the reference's arithmetic is ill-conditioned and parity is with ITS values: a ladder of polar
    # angles from 1e-9 to 1e-2 about either pole, and the axis itself with a negative zero x -- phi = pi there)
    ladder = []
    for _ in range(8):
        th, ph, rr = 10.0 ** rng.uniform(-9, -2), rng.uniform(0, 2 * np.pi), scale * np.exp(rng.uniform(np.log(0.02), np.log(5.0)))
        ladder.append(np.array([rr * np.sin(th) * np.cos(ph), rr * np.sin(th) * np.sin(ph), rng.choice([-1.0, 1.0]) * rr * np.cos(th)]))
    nlad = len(ladder)
    # (... and of radii towards the centre, 1e-3 to 1e-14 of the scale: r = sqrt(...) + 1e-16 there, src/SphericalBasis.cc:1545)
    for _ in range(6):
        u = rng.normal(0, 1, 3)
        ladder.append(u / np.linalg.norm(u) * scale * 10.0 ** rng.uniform(-14, -3))
    ladder += [np.array([-0.0, 0.0, 0.4 * scale]), np.array([-0.0, -0.0, -0.7 * scale])]
    k = min(n, 32)
    idx = rng.choice(n, k, replace=False)
    special = ladder + [np.zeros(3), np.array([0, 0, 0.3 * scale]), np.array([0, 0, -2.0 * scale]), np.array([1e-300, 0, 0]),
               np.array([rmax, 0, 0]), np.array([0, rmax * (1 - 1e-15), 0]), np.array([0, 0, rmax * (1 + 1e-15)]),
               np.array([rmin, 0, 0]), np.array([rmin * (1 - 1e-12), 0, 0]), np.array([3 * rmax, rmax, -5 * rmax]),
               np.array([1e-9 * scale, 1e-9 * scale, scale]), np.array([scale, 0, 1e-200]), np.array([-scale, 1e-17, 0]),
               pos[0] - ctr, pos[0] - ctr, np.array([1e-4, -1e-4, 1e-4]) * scale]
    for j, i in enumerate(idx):
        pos[i] = special[j] + ctr if ctr.any() else special[j]         # (-0 + 0 would be +0)
    if n > 3 and rng.random() < 0.5:
        m[rng.choice(n, 2, replace=False)] = 0.0
    return m, pos, idx[:min(k, 14)]         # (the slots of the two ladders)


def trial_sph(t, rng):
    key, (model, g) = sl_grid(rng)
    scale = float(rng.choice([1.0, 1.0, 0.05, 7.0]))
    rmin = g.rmin * scale * float(rng.choice([1.0, 1.0, 30.0]))
    rmax = g.rmax * scale * float(rng.choice([1.0, 1.0, 0.2]))
    flags = {k: bool(rng.random() < 0.2) for k in ("NO_L0", "NO_L1", "EVEN_L", "EVEN_M", "M0_only")}
    ctr = rng.normal(0, 0.3, 3) * scale if rng.random() < 0.5 else np.zeros(3)
    m, pos, lad = nasty_sphere(rng, model, g, scale, rmin, rmax, ctr)
    prm = orc.params(scale=scale, rmin=rmin, rmax=rmax, **flags)
    # (round 6) one trial in five with the "ssfrac" key on: a sub-sample of the caller's order, 1-5 threads
    # (src/SphericalBasis.cc:437-473; its own generator, so that the other trials of a seed stay what they were)
    srng = np.random.default_rng([int(t), 6])
    ssfrac, nthrds = (float(srng.uniform(0.05, 0.95)), int(srng.integers(1, 6))) if srng.random() < 0.2 else (None, 1)
    with orc.call_opts(ssfrac=ssfrac, nthrds=nthrds):
        c_ref, used_ref = orc.sph_accumulate(g, prm, pos, m, center=ctr)
    a_ref, p_ref = orc.sph_accel(g, prm, pos, c_ref, center=ctr)
    f = SphereSL(ctx, g, scale=scale, rmin=rmin, rmax=rmax, **flags)
    if ssfrac is not None:
        f.set_subset(ssfrac, nthrds)
        key = key + (f"ssfrac {ssfrac:.3f}/{nthrds}",)
    c = Component.from_arrays(ctx, m, pos)
    c.set_center(ctr)
    f.determine_coefficients(c)
    coef, used = f.get_coefs(), f.Used()
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot", "pos"))
    # pyEXP's field evaluation of the same coefficient set at (some of) the same points, in a random coordinate type
    sub = rng.choice(len(m), min(len(m), 300), replace=False)
    q = pos[sub] - ctr
    ctype = str(rng.choice(["cartesian", "cylindrical", "spherical"]))
    if ctype == "cartesian":
        args = (q[:, 0], q[:, 1], q[:, 2])
    elif ctype == "cylindrical":
        args = (np.hypot(q[:, 0], q[:, 1]), q[:, 2], np.arctan2(q[:, 1], q[:, 0]))
    else:
        rq = np.linalg.norm(q, axis=1)
        with np.errstate(all="ignore"):
            args = (rq, np.where(rq > 0, q[:, 2] / rq, 0.0), np.arctan2(q[:, 1], q[:, 0]))
    f.set_coefs(c_ref)
    with np.errstate(all="ignore"):
        fg, fr = f.fields(*args, ctype), orc.sph_fields(g, prm, c_ref, *args, ctype)
    finf = np.isfinite(fr).all(axis=1)
    e_f = field_err(fg[finf], fr[finf]) if finf.any() else 0.0
    same_f = np.array_equal(np.isfinite(fg).all(axis=1), finf)
    c.close(); f.close()
    cs = max(np.abs(c_ref).max(), 1e-300)
    ok = used == used_ref and np.array_equal(out["pos"], pos) and same_f and e_f <= ACC_TOL
    e_c = np.abs(coef - c_ref).max() / cs
    fin = np.isfinite(a_ref).all(axis=1) & np.isfinite(p_ref)
    same_nan = np.array_equal(np.isfinite(out["acc"]).all(axis=1) & np.isfinite(out["pot"]), fin)
    asc = max(np.linalg.norm(a_ref[fin], axis=1).max() if fin.any() else 0.0, 1e-300)
    psc = max(np.abs(p_ref[fin]).max() if fin.any() else 0.0, 1e-300)
    e_a = np.abs(out["acc"][fin] - a_ref[fin]).max() / asc if fin.any() else 0.0
    e_p = np.abs(out["pot"][fin] - p_ref[fin]).max() / psc if fin.any() else 0.0
    # ... and the particles of the polar ladder each against its OWN acceleration (floored at 1e-2 of the largest: deep in the centre the force is what is left of the n-sums)
    lf = lad[fin[lad]]
    e_o = (np.linalg.norm(out["acc"][lf] - a_ref[lf], axis=1) / np.maximum(np.linalg.norm(a_ref[lf], axis=1), 1e-2 * asc)).max() if len(lf) else 0.0
    ok = ok and same_nan and e_c <= COEF_TOL and e_a <= ACC_TOL and e_p <= ACC_TOL and e_o <= ACC_TOL
    print(f"sph {t:3d} {key} scale {scale} flags {[k for k, v in flags.items() if v]} n {len(m)}: coef {e_c:.1e} acc {e_a:.1e} own {e_o:.1e} pot {e_p:.1e} "
          f"fields[{ctype[:3]}] {e_f:.1e}{'' if same_f else ' NAN-PATTERN'} "
          f"used {used}/{used_ref} {'ok' if ok else 'MISMATCH'}", flush=True)
    if not ok and fin.any():
        d = np.abs(out["acc"] - a_ref).max(axis=1)
        d[~fin] = 0.0
        for i in np.argsort(d)[::-1][:3]:
            rr = np.linalg.norm(pos[i] - ctr)
            print(f"    worst particle {i}: pos - ctr {pos[i] - ctr} r/rmax {rr / rmax:.17g} r/rmin {rr / rmin:.6g} acc {out['acc'][i]} ref {a_ref[i]} "
                  f"pot {out['pot'][i]:.17g} ref {p_ref[i]:.17g}", flush=True)
    return ok


def trial_cyl(t, rng):
    key, g = cyl_grid(rng)
    n = int(rng.choice([1, 3, 64, 65, 500, 4000, 20000]))
    A, H = g.ascale, g.hscale
    R = -A * np.log(rng.random(n) * rng.random(n))
    ph = rng.uniform(0, 2 * np.pi, n)
    z = 2 * H * np.arctanh(rng.uniform(-0.999, 0.999, n)) * rng.uniform(0.3, 3.0)
    pos = np.stack([R * np.cos(ph), R * np.sin(ph), z], axis=1)
    Rt = g.rtable * A
    special = [np.array([-0.0, 0.0, -0.6 * H]), np.array([-0.0, -0.0, 0.2 * H]),       # (on the axis, phi = +-pi: IEEE atan2)
               np.zeros(3), np.array([0, 0, H]), np.array([1e-300, 0, 0]), np.array([A, 0, 0.0]), np.array([Rt * 0.999999, 0, 0]),
               np.array([Rt * 1.000001, 0, 0]), np.array([0, 0.3 * Rt, 0.69 * Rt]), np.array([0.1 * Rt, 0, -0.71 * Rt]),
               np.array([5 * Rt, Rt, 0]), np.array([g.rmin * A * 0.5, 0, 0]), np.array([A, 1e-17, 1e-200]), np.array([-A, 0, -H])]
    # (a ladder of distances from the axis, 1e-12 a to 1e-3 a: the force is projected with 1/R and 1/R^2 there)
    for _ in range(8):
        Rr, pp = A * 10.0 ** rng.uniform(-12, -3), rng.uniform(0, 2 * np.pi)
        special.append(np.array([Rr * np.cos(pp), Rr * np.sin(pp), H * rng.normal(0, 1.5)]))
    nlad = min(n, len(special))
    lad = rng.choice(n, nlad, replace=False)
    for j, i in enumerate(lad):
        pos[i] = special[j]
    lad = lad[12:] if nlad > 12 else lad[:0]                    # (the slots of that ladder)
    m = np.full(n, 1.0 / n) * rng.uniform(0.1, 3.0, n)
    even_m = bool(rng.random() < 0.2)
    kw = dict(EVEN_M=even_m) if even_m else {}
    c_ref, s_ref, used_ref, mass_ref = orc.cyl_accumulate(g, pos, m, **kw)
    a_ref, p_ref = orc.cyl_accel(g, pos, c_ref, s_ref, mass_ref, **kw)
    f = Cylinder(ctx, g, EVEN_M=even_m)
    c = Component.from_arrays(ctx, m, pos)
    f.determine_coefficients(c)
    cc, ss = f.get_coefs()
    used, cm = f.Used(), f.cylmass
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pot"))
    # pyEXP's field evaluation (accumulated_eval + accumulated_dens_eval) at some of the points, random coordinate type
    sub = rng.choice(n, min(n, 300), replace=False)
    q = pos[sub]
    ctype = str(rng.choice(["cartesian", "cylindrical", "spherical"]))
    if ctype == "cartesian":
        args = (q[:, 0], q[:, 1], q[:, 2])
    elif ctype == "cylindrical":
        args = (np.hypot(q[:, 0], q[:, 1]), q[:, 2], np.arctan2(q[:, 1], q[:, 0]))
    else:
        rq = np.linalg.norm(q, axis=1)
        with np.errstate(all="ignore"):
            args = (rq, np.where(rq > 0, q[:, 2] / rq, 0.0), np.arctan2(q[:, 1], q[:, 0]))
    f.set_coefs(c_ref, s_ref)
    with np.errstate(all="ignore"):
        fg, fr = f.fields(*args, ctype), orc.cyl_fields(g, c_ref, s_ref, *args, ctype, **kw)
    finf = np.isfinite(fr).all(axis=1)
    # (floors: a thousandth of what a coefficient of the size the particles could produce gives through the largest entry of
    # the density, potential and force tables -- a lone particle at the origin, evaluated at the origin, has a force that
    # vanishes by symmetry and is interpolation rounding, 1e-16 of that scale, on both sides)
    unit = 1e-3 * np.abs(m).sum() * np.abs(g.tab[0]).max()
    floors = (unit * np.abs(g.dens).max(), unit * np.abs(g.tab[0]).max(), unit * max(np.abs(g.tab[1]).max(), np.abs(g.tab[2]).max()))
    e_f = field_err(fg[finf], fr[finf], floors) if finf.any() else 0.0
    same_f = np.array_equal(np.isfinite(fg).all(axis=1), finf)
    if e_f > ACC_TOL:
        for a_, b_ in ((0, 3), (3, 6), (6, 9)):
            d_ = np.abs(fg[finf][:, a_:b_] - fr[finf][:, a_:b_])
            k_ = np.unravel_index(np.argmax(d_), d_.shape)
            print(f"    cols {a_}:{b_} worst |d| {d_.max():.3e} at args {[float(v[finf][k_[0]]) for v in args]} col {a_ + k_[1]} ref "
                  f"{fr[finf][k_[0], a_ + k_[1]]:.6e} got {fg[finf][k_[0], a_ + k_[1]]:.6e} group max {np.abs(fr[finf][:, a_:b_]).max():.3e}")
    c.close(); f.close()
    # (a lone particle in the plane and a basis of vertically antisymmetric functions: every coefficient is 0 in the oracle and
    # a rounding of the maps on the device -- the scale is what the particle COULD have contributed)
    cs = max(np.abs(c_ref).max(), 1e-3 * np.abs(m).sum() * np.abs(g.tab[0]).max())
    e_c = max(np.abs(cc - c_ref).max(), np.abs(ss - s_ref).max()) / cs
    fin = np.isfinite(a_ref).all(axis=1) & np.isfinite(p_ref)          # (on-axis particles: NaN kept as the reference has it)
    same_nan = np.array_equal(np.isfinite(out["acc"]).all(axis=1) & np.isfinite(out["pot"]), fin)
    asc = max(np.linalg.norm(a_ref[fin], axis=1).max() if fin.any() else 0.0, 1e-300)
    psc = max(np.abs(p_ref[fin]).max() if fin.any() else 0.0, 1e-300)
    e_a = np.abs(out["acc"][fin] - a_ref[fin]).max() / asc if fin.any() else 0.0
    e_p = np.abs(out["pot"][fin] - p_ref[fin]).max() / psc if fin.any() else 0.0
    lf = lad[fin[lad]]
    e_o = (np.linalg.norm(out["acc"][lf] - a_ref[lf], axis=1) / np.maximum(np.linalg.norm(a_ref[lf], axis=1), 1e-2 * asc)).max() if len(lf) else 0.0
    ok = (used == used_ref and abs(cm - mass_ref) <= 1e-12 * max(abs(mass_ref), 1e-300) and same_nan and e_c <= COEF_TOL
          and e_a <= ACC_TOL and e_p <= ACC_TOL and same_f and e_f <= ACC_TOL and e_o <= ACC_TOL)
    print(f"cyl {t:3d} {key} EVEN_M {even_m} n {n}: coef {e_c:.1e} acc {e_a:.1e} own {e_o:.1e} pot {e_p:.1e} fields[{ctype[:3]}] {e_f:.1e}"
          f"{'' if same_f else ' NAN-PATTERN'} used {used}/{used_ref} "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
    return ok


def main():
    t0 = time.time()
    bad = 0
    for t in range(trials):
        for kind, fn in (("sph", trial_sph), ("cyl", trial_cyl)):
            if which in (kind, "both"):
                rng = np.random.default_rng([seed0, t, 0 if kind == "sph" else 1])
                if not fn(t, rng):
                    bad += 1
                    print(f"  reproduce: python tests/fuzz/fuzz_parity.py {t + 1} {seed0} {kind}   (trial {t})", flush=True)
    print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
