import sys, numpy as np
sys.path.insert(0, '.')
from exp_amd.runtime import Component, Context, Cylinder, Simulation, SphereSL
from tests.test_cyl_gpu import _disk, cyl_grid
from tests.test_multistep_gpu import _halo
ctx = Context(0)
g, m, pos, vel = _halo(3000, 5)
cg = cyl_grid(4, 6)
pos = pos * (3.0 * cg.ascale); vel = vel * 0.05
dm, dpos, dvel = _disk(3000, 6, cg)
for ms in (0, 2):
    f1 = SphereSL(ctx, g, scale=3.0 * cg.ascale, rmin=g.rmin * 3.0 * cg.ascale, rmax=g.rmax * 3.0 * cg.ascale, multistep=ms)
    f2 = Cylinder(ctx, cg, multistep=ms)
    c1 = Component.from_arrays(ctx, m, pos, vel); c2 = Component.from_arrays(ctx, dm, dpos, dvel)
    sim = Simulation(ctx, 1e-4, multistep=ms, dynfrac=[1e9] * 5)
    i1 = sim.add_component(c1, f1); i2 = sim.add_component(c2, f2)
    sim.add_interaction(i1, i2); sim.add_interaction(i2, i1)
    sim.init()
    for k in range(3):
        o1, o2 = c1.download(), c2.download()
        print(ms, k, "halo |pos|max %.4g |vel|max %.4g |acc|max %.4g  disk |pos|max %.4g |vel| %.4g |acc|max %.4g" % (
            np.abs(o1["pos"]).max(), np.abs(o1["vel"]).max(), np.abs(o1["acc"]).max(),
            np.abs(o2["pos"]).max(), np.abs(o2["vel"]).max(), np.abs(o2["acc"]).max()),
            "lev", np.bincount(c1.download_levels(), minlength=3), np.bincount(c2.download_levels(), minlength=3))
        j = np.argmax(np.abs(o1["acc"]).max(axis=1)); print("   halo worst", j, o1["pos"][j], o1["acc"][j])
        j = np.argmax(np.abs(o2["acc"]).max(axis=1)); print("   disk worst", j, o2["pos"][j], o2["acc"][j])
        sim.step(1)
