import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.conftest import make_grid
from exp_amd.models import sample_sphere
from exp_amd.runtime import Component, SphereSL, Context
model, g = make_grid("plummer", 6, 18, 800)
m, pos, vel = sample_sphere(model, 200000, seed=9)
m = m * np.random.default_rng(1).uniform(0.5, 1.5, len(m))
perm = np.random.default_rng(2).permutation(len(m))
ctx = Context(0); ctx.set_deterministic(True)
f = SphereSL(ctx, g)
res = []
for order in (np.arange(len(m)), np.arange(len(m)), perm):
    c = Component.from_arrays(ctx, m[order], pos[order], vel[order])
    f.determine_coefficients(c)
    c0 = f.get_coefs().copy()
    c.zero_acceleration(0); f.get_acceleration_and_potential(c)
    a0 = c.download()["acc"][np.argsort(order)]
    f.step_kdk(c, 0.01)
    c1 = f.get_coefs().copy()
    res.append((c0, a0, c1)); c.close()
for k in (1, 2):
    print([float(np.abs(a - b).max()) for a, b in zip(res[0], res[k])], np.abs(res[0][0]).max())
