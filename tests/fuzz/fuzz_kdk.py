"""Randomised campaign for the fused single-level KDK step (GPU): `exp_amd_step_kdk` / `exp_amd_step_kdk_n` of both force
methods against the n-body oracle with multistep 0, over random bases, particle numbers, time steps (changed between
steps), step counts, the pre-kicked store on and off, graph replay of step pairs, and read-only calls interleaved between the
steps (download, fix_positions, log sums, coefficient read-back -- none of which may change the trajectory).
    python tests/fuzz/fuzz_kdk.py [trials=60] [seed=1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import tests.config4_util as c4
from exp_amd.runtime import Component, Context, Cylinder, SphereSL
from tests.oracle_lib import NBodyOracle, Oracle

_argv = sys.argv if __name__ == "__main__" else [""]   # imported by tests/test_fuzz_gpu.py: defaults
trials = int(_argv[1]) if len(_argv) > 1 else 60
seed0 = int(_argv[2]) if len(_argv) > 2 else 1
orc = Oracle()
ctx = Context(0)
g, cg = c4.grids()


def one(t, rng):
    sphere = bool(rng.random() < 0.5)
    n = int(rng.choice([1, 63, 64, 65, 1000, 5000, 30000]))
    inp = c4.config4_inputs(n_halo=n if sphere else 8, n_disk=8 if sphere else n)
    sc = float(inp["scale"])
    key = "halo" if sphere else "disk"
    m, pos, vel = inp[key + "_mass"], inp[key + "_pos"], inp[key + "_vel"]
    prekick = bool(rng.random() < 0.5)
    ctx.set_prekick(prekick)
    dt0 = c4.DTIME * float(rng.choice([0.25, 1.0, 4.0]))
    nb = NBodyOracle(orc, 0, dt0, list(c4.DYN))
    if sphere:
        prm = orc.params(**c4.sph_window(g, sc))
        nb.add_sphere(g, prm, m, pos, vel)
        f = SphereSL(ctx, g, **c4.sph_window(g, sc))
    else:
        nb.add_cylinder(cg, m, pos, vel)
        f = Cylinder(ctx, cg)
    nb.init()
    c = Component.from_arrays(ctx, m, pos, vel)
    f.determine_coefficients(c)
    f.get_acceleration_and_potential(c)
    nsteps = int(rng.integers(1, 7))
    status, detail = "ok", ""
    k = 0
    hist = []
    while k < nsteps and status == "ok":
        pair = bool(rng.random() < 0.3) and k + 2 <= nsteps
        try:
            if pair:
                hist.append("pair")
                f.step_kdk_n(c, dt0, 2)
                nb.step(); nb.step()
                k += 2
            else:
                hist.append("step")
                f.step_kdk(c, dt0)
                nb.step()
                k += 1
        except Exception as e:                                  # an error status of the C-ABI is a finding too
            status, detail = "MISMATCH", f"{type(e).__name__}: {e} after {hist}"
            break
        peek = int(rng.integers(0, 5))                          # a read-only call between steps
        hist.append(["-", "fix_positions", "log_sums", "get_coefs", "download"][peek])
        if peek == 1:
            c.fix_positions(0)
        elif peek == 2:
            # the run log's sums see the step-boundary state whatever the store holds (pre-kicked or with its closing
            # half-kick still deferred)
            s = nb.state[0]
            ls = c.log_sums()
            ref = orc.outlog_sums(s["mass"] if "mass" in s else m, np.stack([s[q] for q in "xyz"], 1), np.stack([s["v" + q] for q in "xyz"], 1),
                                  np.stack([s["a" + q] for q in "xyz"], 1), s["pot"])
            for key_ in ("ektot", "eptot", "clausius", "mtot"):
                if abs(ls[key_] - ref[key_]) > 1e-9 * max(abs(ref[key_]), 1e-300):
                    status, detail = "MISMATCH", f"log_sums {key_} {ls[key_]!r} vs {ref[key_]!r} after {hist}"
            if np.abs(ls["angm"] - ref["angm"]).max() > 1e-9 * max(np.abs(m).sum() * np.abs(pos).max() * max(np.abs(vel).max(), 1e-300), 1e-300):
                status, detail = "MISMATCH", f"log_sums angm after {hist}"
        elif peek == 3:
            f.get_coefs()
        if peek == 4 or k >= nsteps:
            s = nb.state[0]
            out = c.download()
            p = np.stack([s[q] for q in "xyz"], 1)
            v = np.stack([s["v" + q] for q in "xyz"], 1)
            a = np.stack([s["a" + q] for q in "xyz"], 1)
            e = [np.abs(out["pos"] - p).max() / 1e-11, np.abs(out["vel"] - v).max() / (1e-9 * max(np.abs(v).max(), 1e-300)),
                 np.abs(out["acc"] - a).max() / (1e-9 * max(np.linalg.norm(a, axis=1).max(), 1e-300)),
                 np.abs(out["pot"] - s["pot"]).max() / (1e-9 * max(np.abs(s["pot"]).max(), 1e-300))]
            gn = f.get_coefs()
            gn = gn.reshape(-1) if sphere else np.concatenate([x.reshape(-1) for x in gn])
            cmax = max(np.abs(s["coefN"][0]).max(), 1e-300)
            e.append(np.abs(gn - s["coefN"][0]).max() / (1e-10 * cmax))
            if max(e) > 1.0 or not np.isfinite(max(e)):
                status, detail = "MISMATCH", f"after step {k}: ratios to tolerance {np.round(e, 2).tolist()}"
    print(f"{t:3d} {'sph' if sphere else 'cyl'} n {n} dt {dt0:.1e} steps {nsteps} prekick {prekick}: {status} {detail}", flush=True)
    c.close(); f.close()
    return status == "ok"


def main():
    t0 = time.time()
    bad = sum(0 if one(t, np.random.default_rng([seed0, t])) else 1 for t in range(trials))
    ctx.set_prekick(True)
    print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
