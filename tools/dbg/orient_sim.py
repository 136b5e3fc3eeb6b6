import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from exp_amd.models import sample_sphere
from exp_amd.runtime import Component, Context, Orient, Simulation, SphereSL
from tests.conftest import make_grid
ctx = Context(0)
model, g = make_grid("plummer", 4, 8, 400)
n, dt, nstep = 20000, 0.02, 3
m, pos, vel = sample_sphere(model, n, seed=43)
pos = pos + np.array([0.3, 0.0, -0.1]); vel = vel + np.array([0.5, -0.25, 0.125])
def mk():
    return SphereSL(ctx, g), Component.from_arrays(ctx, m, pos, vel), Orient(ctx, 2, 1500, Orient.CENTER, Orient.KE, dT=0.0, damping=1.0)
for attach in (False, True):
    f, c, o = mk()
    sim = Simulation(ctx, dt); sim.add_component(c, f)
    if attach: sim.set_orient(0, o)
    sim.init()
    f2, c2, o2 = mk()
    def potential(tnow, gp):
        if attach:
            ctr = o2.currentCenter(); c2.set_center(ctr)
            if gp: o2.accumulate(tnow, c2, dt)
        c2.zero_acceleration(0); f2.get_acceleration_and_potential(c2)
    f2.set_multistep_level(0); f2.determine_coefficients(c2); potential(0.0, False)
    t = 0.0
    for k in range(nstep):
        sim.step(1)
        t += dt
        c2.incr_velocity(0.5 * dt); c2.incr_position(dt); f2.determine_coefficients(c2); potential(t, True); c2.incr_velocity(0.5 * dt)
        a, b = c.download(("pos", "vel", "acc")), c2.download(("pos", "vel", "acc"))
        print(attach, k, [float(np.abs(a[q] - b[q]).max()) for q in a], o.state()["center"], o2.state()["center"], o.state()["Ecurr"], o2.state()["Ecurr"])
