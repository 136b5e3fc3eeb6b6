"""Randomised campaign for the particle store's primitives (GPU): random sequences of upload, upload_acc, upload_levels,
kick / drift by level (Component::incr_velocity / incr_position, src/incvel.cc:90, src/incpos.cc:72), zero_acceleration,
set_center, the re-ordering a coefficient accumulation performs, a fused step, fix_positions, log sums and downloads
against a plain numpy model of the same operations in the CALLER's order: positions, velocities, accelerations, levels must
come back bit for bit (kick and drift are one rounding each: v + a dt, x + v dt), the centre-of-mass sums to 1e-12.
    python tests/fuzz/fuzz_store.py [trials=100] [seed=1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from exp_amd._lib import ExpAmdError
from exp_amd.runtime import Component, Context, SphereSL
from tests.conftest import make_grid

_argv = sys.argv if __name__ == "__main__" else [""]   # imported by tests/test_fuzz_gpu.py: defaults
trials = int(_argv[1]) if len(_argv) > 1 else 100
seed0 = int(_argv[2]) if len(_argv) > 2 else 1
ctx = Context(0)
model, g = make_grid("plummer", 2, 4, 400)


def one(t, rng):
    n = int(rng.choice([1, 2, 63, 64, 65, 1000, 20000]))
    ms = int(rng.choice([0, 0, 2, 4]))
    m = rng.uniform(0.5, 1.5, n) / n
    x, v = rng.normal(0, 0.5, (n, 3)), rng.normal(0, 0.3, (n, 3))
    a, pot = np.zeros((n, 3)), np.zeros(n)
    lev = np.zeros(n, dtype=np.int32)
    c = Component.from_arrays(ctx, m, x, v)
    f = SphereSL(ctx, g, multistep=ms)
    if ms:                                                 # (a kick of level L > 0 needs the levels to exist: ERR_ARG otherwise)
        lev = rng.integers(0, ms + 1, n).astype(np.int32)
        c.upload_levels(lev)
        f.set_multistep_level(0)
        f.determine_coefficients(c)                        # partitions the store by level: level-specific calls are defined on
    hist, status = [], "ok"                                # a partitioned store (the reference's levlist), as every driver has it
    for _ in range(int(rng.integers(3, 14))):
        op = str(rng.choice(["acc", "levels", "kick", "drift", "zero", "center", "sort", "upload", "peek", "download"]))
        hist.append(op)
        if op == "acc":
            a, pot = rng.normal(0, 1.0, (n, 3)), -rng.uniform(0.5, 2, n)
            c.upload_acc(a, pot)
        elif op == "levels" and ms:
            lev = rng.integers(0, ms + 1, n).astype(np.int32)
            c.upload_levels(lev)
            f.set_multistep_level(0)
            f.determine_coefficients(c)
        elif op == "kick":
            L = int(rng.integers(-1, ms + 1)) if ms else -1
            dt = float(rng.choice([0.01, -0.003, 0.5]))
            try:
                c.incr_velocity(dt, L)
            except ExpAmdError as e:                       # a level the store does not have yet (no multistep force has
                assert "beyond" in str(e), e               # partitioned it): refused with ERR_ARG, nothing changes
                continue
            sel = np.ones(n, bool) if L < 0 else lev == L      # exactly that level (src/step.cc:126-160 calls it per level)
            v = np.where(sel[:, None], v + a * dt, v)
        elif op == "drift":
            L = int(rng.integers(-1, ms + 1)) if ms else -1
            dt = float(rng.choice([0.01, -0.003, 0.5]))
            try:
                c.incr_position(dt, L)
            except ExpAmdError as e:
                assert "beyond" in str(e), e
                continue
            sel = np.ones(n, bool) if L < 0 else lev == L
            x = np.where(sel[:, None], x + v * dt, x)
        elif op == "zero":
            L = int(rng.integers(0, ms + 1)) if ms else 0
            try:
                c.zero_acceleration(L)
            except ExpAmdError as e:
                assert "beyond" in str(e), e
                continue
            sel = lev >= L
            a = np.where(sel[:, None], 0.0, a)
            pot = np.where(sel, 0.0, pot)
        elif op == "center":
            c.set_center(rng.normal(0, 0.05, 3))
        elif op == "sort":
            if ms:
                f.set_multistep_level(int(rng.integers(0, ms + 1)))
            f.determine_coefficients(c)                    # re-orders the store by (level, cell); caller order is kept
        elif op == "upload":
            x, v = rng.normal(0, 0.5, (n, 3)), rng.normal(0, 0.3, (n, 3))
            c.upload(m, x, v)
        elif op == "peek":
            fp = c.fix_positions(0)
            com = (m[:, None] * x).sum(0) / m.sum()
            if np.abs(fp["com"] - com).max() > 1e-12 * max(1.0, np.abs(x).max()) or abs(fp["mtot"] - m.sum()) > 1e-13:
                status = f"MISMATCH fix_positions after {hist}"
                break
            ls = c.log_sums()
            ek = 0.5 * (m * (v * v).sum(1)).sum()
            if abs(ls["ektot"] - ek) > 1e-12 * max(ek, 1e-300) or ls["nbodies"] != n:
                status = f"MISMATCH log_sums after {hist}"
                break
        else:
            out = c.download()
            got_lev = c.download_levels() if ms else lev
            for key, want in (("pos", x), ("vel", v), ("acc", a), ("pot", pot), ("mass", m)):
                if not np.array_equal(out[key], want):
                    k = np.unravel_index(np.argmax(np.abs(out[key] - want)), want.shape)
                    status = f"MISMATCH {key} differs by {np.abs(out[key] - want).max():.2e} at {k} after {hist}"
                    break
            if status == "ok" and not np.array_equal(got_lev, lev):
                status = f"MISMATCH levels after {hist}"
            if status != "ok":
                break
    print(f"{t:3d} n {n} multistep {ms} ops {len(hist)}: {status}", flush=True)
    c.close(); f.close()
    return status == "ok"


def main():
    t0 = time.time()
    bad = sum(0 if one(t, np.random.default_rng([seed0, t])) else 1 for t in range(trials))
    print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
