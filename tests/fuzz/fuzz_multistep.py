"""Randomised block-multistep campaign (GPU): the C++ step driver against the n-body oracle (oracle/nbody_oracle.c) over
random level counts, time steps, time-step criteria, component sizes, interaction lists, sparse-level thresholds and
differencing routes; levels bit for bit, states and per-level coefficient sets to the tolerances of
tests/test_config4_gpu.py.  One line per trial; a level mismatch of a handful of particles is a time-step criterion
within rounding of a power-of-two boundary (reported as `edge`), anything else is a failure.

    python tests/fuzz/fuzz_multistep.py [trials=30] [seed=1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import tests.config4_util as c4
from exp_amd.runtime import Component, Context, Cylinder, Simulation, SphereSL
from tests.oracle_lib import NBodyOracle, Oracle

_argv = sys.argv if __name__ == "__main__" else [""]   # imported by tests/test_fuzz_gpu.py: defaults
trials = int(_argv[1]) if len(_argv) > 1 else 30
seed0 = int(_argv[2]) if len(_argv) > 2 else 1
orc = Oracle()
ctx = Context(0)
g, cg = c4.grids()


def _apply(nb, io, sim, idd, comp, o):
    """the option keys of one component on both sides"""
    if not o:
        return
    nb.set_options(io, rtrunc=o.get("rtrunc"), com0=o.get("com0", (0.0, 0.0, 0.0)), adiabatic=o.get("adiabatic"),
                   self_consistent=o.get("self_consistent", True), fix_l0=o.get("fix_l0", False), mlim=o.get("mlim"),
                   freeze_levels=o.get("freeze_levels", False), noswitch=o.get("noswitch", False), dtreset=o.get("dtreset", True))
    if o.get("freeze_levels") or o.get("noswitch"):
        comp.set_level_policy(noswitch=o.get("noswitch", False), freeze_levels=o.get("freeze_levels", False),
                              dtreset=o.get("dtreset", True))
    if o.get("rtrunc") is not None:
        comp.set_rtrunc(o["rtrunc"], o.get("com0"))
    if o.get("adiabatic") is not None:
        sim.set_adiabatic(idd, *o["adiabatic"])


def one(t, rng):
    ms = int(rng.integers(1, 5))
    dtime = c4.DTIME * float(rng.choice([0.3, 1.0, 2.5]))
    dyn = list(c4.DYN)
    dyn[1] *= float(rng.choice([0.5, 1.0, 2.0]))
    dyn[3] *= float(rng.choice([0.5, 1.0, 3.0]))
    nh, nd = int(rng.choice([1, 130, 600, 2500])), int(rng.choice([1, 130, 600, 2500]))
    inp = c4.config4_inputs(n_halo=nh, n_disk=nd)
    which = rng.choice(["both", "halo", "disk"], p=[0.7, 0.15, 0.15])
    inter = rng.choice(["both", "one", "none"], p=[0.6, 0.2, 0.2]) if which == "both" else "none"
    dense_min, list_min = int(rng.choice([-1, 0, 50, 500])), int(rng.choice([0, 16, 2048]))
    nsteps = int(rng.integers(1, 4))
    thin_max = int(rng.choice([0, 30, 400, 16384]))           # (drawn last: the earlier draws of a (seed, trial) stay what they were)
    # round 5, drawn after everything else: the keys that default to off -- rtrunc / com0 (Component::freeze), ton / toff /
    # twid (Adiabatic), self_consistent: false, FIX_L0 (sphere), mlim (cylinder) -- on half of the trials
    oh, od = {}, {}
    if rng.random() < 0.5:
        for o, kind in ((oh, "halo"), (od, "disk")):
            if rng.random() < 0.5:
                o["rtrunc"] = float(rng.choice([0.02, 0.05, 0.3] if kind == "disk" else [0.05, 0.3, 1.0]))
                o["com0"] = tuple(float(v) for v in rng.normal(0.0, 0.003, 3))
            if rng.random() < 0.4:
                o["adiabatic"] = (float(rng.uniform(-1.0, 2.0)) * dtime, float(rng.choice([1e20, 2.5 * dtime])), float(rng.uniform(0.3, 2.0)) * dtime)
            if rng.random() < 0.25:
                o["self_consistent"] = False
        if rng.random() < 0.3:
            oh["fix_l0"] = True
        if rng.random() < 0.4:
            od["mlim"] = int(rng.integers(0, cg.mmax + 1))
    sc = float(inp["scale"])
    # round 5, drawn after the option keys: on a third of the trials a few particles sit where the reference's arithmetic is
    # ill-conditioned -- a ladder of polar angles and of radii towards the centre in the halo, of distances from the axis in
    # the disk -- and a few start at rest (v.a = 0, v.v = 0 in the time-step criteria)
    if rng.random() < 0.33:
        hp, hv, dp, dv = (inp[k].copy() for k in ("halo_pos", "halo_vel", "disk_pos", "disk_vel"))
        # (not in a halo of a few particles: one that sits 1e-4 from the centre IS the expansion, and the 1e-11 its position is
        # held to is 1e-7 of its radius -- the field every other particle feels then differs by more than the bars allow,
        # seed 307 trial 862)
        nhl = len(hp) if len(hp) >= 100 else 0
        for i in range(min(nhl, 8)):
            th, ph, rr = 10.0 ** rng.uniform(-9, -2), rng.uniform(0, 2 * np.pi), sc * np.exp(rng.uniform(np.log(0.05), np.log(3.0)))
            hp[i] = [rr * np.sin(th) * np.cos(ph), rr * np.sin(th) * np.sin(ph), rng.choice([-1.0, 1.0]) * rr * np.cos(th)]
        for i in range(8, min(nhl, 12)):
            u = rng.normal(0, 1, 3)
            hp[i] = u / np.linalg.norm(u) * sc * 10.0 ** rng.uniform(-12, -3)
        for i in range(12, min(nhl, 15)):
            hv[i] = 0.0
        # (not in a disk of one: the azimuthal force of a particle on ITSELF cancels exactly, Pc sin(m phi) - Ps cos(m phi)
        # with (Pc, Ps) ~ (cos, sin)(m phi), and what rounding leaves of it is divided by R: 1e-8 of the force at R = 1e-13)
        # (... nor under a halo of a few particles: the field of ONE halo particle carries every harmonic at full weight, and a
        # disk particle 1e-2 rad from its polar axis then sits 1.02 x the 1e-9 bar from the oracle -- seed 409 trial 1326, the
        # halo's force alone: 2.1e-9 of |a|; nhl == 0 there)
        for i in range(min(len(dp), 6) if (len(dp) >= 100 and (nhl or which != "both")) else 0):
            Rr, ph = 0.01 * 10.0 ** rng.uniform(-12, -3), rng.uniform(0, 2 * np.pi)
            dp[i] = [Rr * np.cos(ph), Rr * np.sin(ph), 0.001 * rng.normal(0, 1.0)]
        for i in range(6, min(len(dp), 8)):
            dv[i] = 0.0
        inp = dict(inp, halo_pos=hp, halo_vel=hv, disk_pos=dp, disk_vel=dv)
    # round 5 (second session), drawn last of all: the component keys adjust_multistep_level reads -- noswitch / dtreset /
    # freezeL (src/multistep.cc:136-158) -- on a third of the trials
    if rng.random() < 0.33:
        for o in (oh, od):
            if rng.random() < 0.5:
                o["noswitch"] = True
                o["dtreset"] = bool(rng.random() < 0.5)
            if rng.random() < 0.25:
                o["freeze_levels"] = True
    prm = orc.params(**c4.sph_window(g, sc))
    nb = NBodyOracle(orc, ms, dtime, dyn)
    ctx.set_dense_min(dense_min)
    ctx.set_mover_list_min(list_min)
    ctx.set_thin_max(thin_max)
    sim = Simulation(ctx, dtime, multistep=ms, dynfrac=dyn, shiftlevl=0)
    forces, comps, names, ids_o, ids_d = [], [], [], [], []
    if which in ("both", "halo"):
        ids_o.append(nb.add_sphere(g, prm, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"]))
        f = SphereSL(ctx, g, multistep=ms, self_consistent=oh.get("self_consistent", True), FIX_L0=oh.get("fix_l0", False),
                     **c4.sph_window(g, sc))
        c = Component.from_arrays(ctx, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
        ids_d.append(sim.add_component(c, f)); forces.append(f); comps.append(c); names.append("halo")
        _apply(nb, ids_o[-1], sim, ids_d[-1], c, oh)
    if which in ("both", "disk"):
        ids_o.append(nb.add_cylinder(cg, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"]))
        f = Cylinder(ctx, cg, multistep=ms, self_consistent=od.get("self_consistent", True),
                     mlim=od["mlim"] if "mlim" in od else -1)
        c = Component.from_arrays(ctx, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
        ids_d.append(sim.add_component(c, f)); forces.append(f); comps.append(c); names.append("disk")
        _apply(nb, ids_o[-1], sim, ids_d[-1], c, od)
    if inter in ("both", "one"):
        nb.add_interaction(ids_o[0], ids_o[1]); sim.add_interaction(ids_d[0], ids_d[1])
    if inter == "both":
        nb.add_interaction(ids_o[1], ids_o[0]); sim.add_interaction(ids_d[1], ids_d[0])
    nb.init()
    sim.init()
    status, detail = "ok", ""
    switches = 0
    for k in range(nsteps + 1):
        if k:
            # read-only calls between master steps must not change the trajectory
            for c_ in comps:
                peek = int(rng.integers(0, 4))
                if peek == 1:
                    c_.fix_positions(0)
                elif peek == 2:
                    c_.log_sums()
                elif peek == 3:
                    c_.download_levels()
            nsw = nb.step()
            sim.step(1)
            switches += int(sum(nsw))
            if sim.step_switches != sum(nsw):
                status, detail = "STATE", f"step {k}: {sim.step_switches} level changes, oracle {sum(nsw)}"
                break
        for j, (name, f, c) in enumerate(zip(names, forces, comps)):
            s = nb.state[j]
            lev = c.download_levels()
            nbad = int((lev != s["level"]).sum())
            if nbad:
                status, detail = ("edge" if nbad <= 3 else "LEVELS"), f"step {k} {name}: {nbad} levels differ"
                break
            out = c.download()
            p = np.stack([s[q] for q in "xyz"], 1)
            v = np.stack([s["v" + q] for q in "xyz"], 1)
            a = np.stack([s["a" + q] for q in "xyz"], 1)
            # (positions: 1e-11 of the box, or of the distance a particle has been flung to -- one that starts 1e-13 from the
            # centre of an l >= 1 expansion sees a gradient ~ 1/r and leaves at 1e6, in the reference as here)
            e = [(np.abs(out["pos"] - p).max(1) / (1e-11 * np.maximum(1.0, np.abs(p).max(1)))).max(),
                 np.abs(out["vel"] - v).max() / (1e-9 * max(np.abs(v).max(), 1e-300)),
                 np.abs(out["acc"] - a).max() / (1e-9 * max(np.linalg.norm(a, axis=1).max(), 1e-300)),
                 np.abs(out["pot"] - s["pot"]).max() / (1e-9 * max(np.abs(s["pot"]).max(), 1e-300))]
            cmax = max(np.abs(s["coefN"]).max(), 1e-300)
            frozen_set = not (oh if name == "halo" else od).get("self_consistent", True)
            for M in range(ms + 1):
                if frozen_set:      # (the per-level sets are dead once the coefficients are held fixed: the reference's sphere
                    break           #  still differences into them, its cylinder does not, nothing reads them; the COMBINED set is compared)
                gn = f.get_coefs(level=M)
                gn = gn.reshape(-1) if name == "halo" else np.concatenate([x.reshape(-1) for x in gn])
                e.append(np.abs(gn - s["coefN"][M]).max() / (1e-10 * cmax))
            gc = f.get_coefs()
            gc = gc.reshape(-1) if name == "halo" else np.concatenate([x.reshape(-1) for x in gc])
            e.append(np.abs(gc - s["coef"]).max() / (1e-10 * max(np.abs(s["coef"]).max(), 1e-300)))
            if max(e) > 1.0 or not np.isfinite(max(e)):
                status, detail = "STATE", f"step {k} {name}: worst ratio to tolerance {max(e):.2g} ({np.round(e, 2).tolist()})"
                if os.environ.get("FUZZ_VERBOSE"):
                    ip, ia = int(np.abs(out["pos"] - p).max(1).argmax()), int(np.abs(out["acc"] - a).max(1).argmax())
                    o0 = inp["halo_pos" if name == "halo" else "disk_pos"]
                    if name == "disk" and which == "both" and os.environ.get("FUZZ_VERBOSE") == "2":
                        # (which of the two forces on the disk differs: the halo's coefficient set on the disk's positions,
                        # oracle against device, as an external target)
                        hc = forces[0].get_coefs()
                        a_o, p_o = orc.sph_accel(g, prm, p, hc)
                        tmp = Component.from_arrays(ctx, np.ones(len(p)), p)
                        tmp.zero_acceleration(0)
                        forces[0].get_acceleration_and_potential(tmp, external=True)
                        a_d = tmp.download(("acc",))["acc"]
                        tmp.close()
                        dd = np.abs(a_d - a_o).max(1)
                        detail += f"\n    halo-on-disk alone: worst |dev - oracle| {dd.max():.3e} at particle {int(dd.argmax())} (|a| {np.linalg.norm(a_o[int(dd.argmax())]):.3e}); at particle {ia}: {dd[ia]:.3e} of {np.linalg.norm(a_o[ia]):.3e}"
                    detail += (f"\n    pos: particle {ip} started at {o0[ip]} now {out['pos'][ip]} oracle {p[ip]} vel {out['vel'][ip]} / {v[ip]} level {lev[ip]}"
                               f"\n    acc: particle {ia} started at {o0[ia]} acc {out['acc'][ia]} oracle {a[ia]} pos {out['pos'][ia]}")
                break
        if status != "ok":
            break
    print(f"{t:3d} ms {ms} dtime {dtime:.2e} n {nh}/{nd} {which} inter {inter} dense_min {dense_min} list_min {list_min} thin_max {thin_max} steps {nsteps}: "
          f"{'opts ' + str(oh) + ' ' + str(od) + ' ' if (oh or od) else ''}{status} {detail} [{switches} level changes, populated levels "
          f"{[int((np.bincount(st['level'], minlength=ms + 1) > 0).sum()) for st in nb.state]}]", flush=True)
    global total_switches
    total_switches += switches
    sim.close()
    for c in comps:
        c.close()
    for f in forces:
        f.close()
    return status


total_switches = 0


def main():
    global total_switches
    total_switches = 0
    t0 = time.time()
    tally = {"ok": 0, "edge": 0, "LEVELS": 0, "STATE": 0}
    for t in range(trials):
        tally[one(t, np.random.default_rng([seed0, t]))] += 1
    ctx.set_dense_min(-1)
    ctx.set_mover_list_min(8192)
    ctx.set_thin_max(8192)
    print(f"{trials} trials: {tally}, {total_switches} level changes in all, {time.time() - t0:.0f} s")
    sys.exit(1 if tally["LEVELS"] or tally["STATE"] else 0)


if __name__ == "__main__":
    main()
