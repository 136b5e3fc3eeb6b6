"""Long run of config 4 at reduced size: N master steps, then the self-consistency identities of
tests/test_config4_gpu.py::test_config4_full_size_level_sets_add_up (per-level coefficient sets add up to a
from-scratch accumulation of the final state), finite state, total level changes.   python tools/dbg/soak_cfg4.py [n] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from exp_amd.empcyl import build_empcyl
from exp_amd.models import NFWModel, sample_disk, sample_sphere
from exp_amd.runtime import Component, Context, Cylinder, Simulation, SphereSL
from exp_amd.slgrid import build_slgrid

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ms, a, h, scale = 4, 0.01, 0.001, 0.1
ctx = Context(0)
model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
g = build_slgrid(model, 6, 18, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
cg = build_empcyl(mmax=6, norder=12, numx=256, numy=128, acyl=a, hcyl=h, lmaxfid=16, nmaxfid=12, numr=800, rnum=100, tnum=40)
hm, hpos, hvel = sample_sphere(model, n, seed=23)
hpos, hvel = hpos * scale, hvel * np.sqrt(1.0 / scale)
dm, dpos, dvel = sample_disk(n, 29, a=a, h=h, mass=0.1)
dvel = dvel + 0.3 * np.random.default_rng(3).standard_normal(dvel.shape)
kw = dict(scale=scale, rmin=g.rmin * scale, rmax=g.rmax * scale)
f1, f2 = SphereSL(ctx, g, multistep=ms, **kw), Cylinder(ctx, cg, multistep=ms)
c1, c2 = Component.from_arrays(ctx, hm, hpos, hvel), Component.from_arrays(ctx, dm, dpos, dvel)
sim = Simulation(ctx, 4e-4, multistep=ms)
i1, i2 = sim.add_component(c1, f1), sim.add_component(c2, f2)
sim.add_interaction(i1, i2); sim.add_interaction(i2, i1)
sim.init()
tot = 0
for k in range(steps):
    sim.step(1)
    tot += sim.step_switches
    if k % 50 == 49:
        print("step", k + 1, "switches so far", tot, "levels", np.bincount(c1.download_levels(), minlength=ms + 1).tolist(),
              np.bincount(c2.download_levels(), minlength=ms + 1).tolist(), flush=True)
flat = lambda x: np.ravel(x) if isinstance(x, np.ndarray) else np.concatenate([np.ravel(y) for y in x])
ok = True
for c, f, fresh in ((c1, f1, lambda: SphereSL(ctx, g, **kw)), (c2, f2, lambda: Cylinder(ctx, cg))):
    out = c.download(("mass", "pos", "vel", "acc"))
    assert all(np.isfinite(out[k]).all() for k in ("pos", "vel", "acc")), "non-finite state"
    total = np.sum([flat(f.get_coefs(level=M)) for M in range(ms + 1)], axis=0)
    ff = fresh(); cc = Component.from_arrays(ctx, out["mass"], out["pos"])
    ff.determine_coefficients(cc)
    ref = flat(ff.get_coefs())
    err = np.abs(total - ref).max() / np.abs(total).max()
    print(type(f).__name__, "sum of level sets vs from-scratch accumulation:", err)
    ok = ok and err <= 1e-8
print("OK" if ok else "FAILED", "after", steps, "master steps,", tot, "level changes")
sys.exit(0 if ok else 1)
