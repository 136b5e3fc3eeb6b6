import sys, numpy as np
sys.path.insert(0, '.')
from exp_amd.runtime import Component, Context, Cylinder, Simulation, SphereSL
from tests.test_cyl_gpu import _disk, cyl_grid
from tests.test_multistep_gpu import _halo
ctx = Context(0)
g, m, pos, vel = _halo(3000, 5)
cg = cyl_grid(4, 6)
sc = 3.0 * cg.ascale
pos, vel = pos * sc, vel * 0.05
dm, dpos, dvel = _disk(3000, 6, cg)
dvel = dvel + 0.3 * np.random.default_rng(2).standard_normal(dvel.shape)
dt, ms = 1e-4, 2
kw = dict(scale=sc, rmin=g.rmin * sc, rmax=g.rmax * sc)
f1, f2 = SphereSL(ctx, g, multistep=ms, **kw), Cylinder(ctx, cg, multistep=ms)
c1, c2 = Component.from_arrays(ctx, m, pos, vel), Component.from_arrays(ctx, dm, dpos, dvel)
sim = Simulation(ctx, dt, multistep=ms, dynfrac=[1e9] * 5)
i1, i2 = sim.add_component(c1, f1), sim.add_component(c2, f2)
sim.add_interaction(i1, i2); sim.add_interaction(i2, i1)
sim.init()
g1, g2 = c1.download(), c2.download()
h1, h2 = SphereSL(ctx, g, **kw), Cylinder(ctx, cg)
d1, d2 = Component.from_arrays(ctx, m, pos, vel), Component.from_arrays(ctx, dm, dpos, dvel)
h1.determine_coefficients(d1); h2.determine_coefficients(d2)
d1.zero_acceleration(); d2.zero_acceleration()
h1.get_acceleration_and_potential(d1); h2.get_acceleration_and_potential(d2)
h1.get_acceleration_and_potential(d2, external=True); h2.get_acceleration_and_potential(d1, external=True)
w1, w2 = d1.download(), d2.download()
for nm, a, b in (("halo", g1, w1), ("disk", g2, w2)):
    for k in ("pos", "vel", "acc", "pot"):
        print(nm, k, np.abs(a[k] - b[k]).max(), np.abs(b[k]).max())
print("coef N0 vs", np.abs(f1.get_coefs(level=0) - h1.get_coefs()).max(), np.abs(f1.get_coefs() - h1.get_coefs()).max(), np.abs(h1.get_coefs()).max())
print("coef L0 vs", np.abs(f1.get_coefs(level=0, last=True) - h1.get_coefs()).max())
cc, ss = f2.get_coefs(); hc, hs = h2.get_coefs()
print("cyl", np.abs(cc - hc).max(), np.abs(hc).max(), f2.cylmass, h2.cylmass, f2.Used(), h2.Used())
print("levels", np.bincount(c1.download_levels()), np.bincount(c2.download_levels()))
