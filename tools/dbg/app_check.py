import sys, numpy as np
sys.path.insert(0, '/root/repo')
from tests.conftest import make_grid
from exp_amd.models import sample_sphere
from exp_amd.runtime import Component, Context, SphereSL
model, g = make_grid("plummer", 6, 18, 800)
n = 300007
m, pos, vel = sample_sphere(model, n, seed=41)
ctx = Context(0)
for violent in (False, True):
    v = vel + (8.0 * pos / np.linalg.norm(pos, axis=1)[:, None] if violent else 0.0)
    ctx.set_append_min(1000)
    f = SphereSL(ctx, g); c = Component.from_arrays(ctx, m, pos, v)
    f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
    for k in range(8):
        ctx.profile(True); ctx.profile_reset()
        f.step_kdk(c, 0.01)
        rep = ctx.profile_report()
        print(violent, k, {a: (b["launches"], round(b["ms_total"], 3)) for a, b in rep.items() if b["launches"]})
        ctx.profile(False)
    c.close(); f.close()
