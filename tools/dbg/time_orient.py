# time of one Orient::accumulate (device selection of the most bound particles) and of one
# Component::fix_positions reduction at bench scale
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from exp_amd.runtime import Component, Context, Orient
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ctx = Context(0)
rng = np.random.default_rng(1)
c = Component(ctx, n)
chunk = 10_000_000
pos = rng.standard_normal((n, 3)).astype(np.float64)
vel = rng.standard_normal((n, 3)) * 0.3
c.upload(np.full(n, 1.0 / n), pos, vel)
pot = -1.0 / np.sqrt(0.05 + (pos ** 2).sum(axis=1))
c.upload_acc(np.zeros((n, 3)), pot)
del pos, vel, pot
o = Orient(ctx, 2, 100000, Orient.AXIS | Orient.CENTER, Orient.KE)
ctx.profile(True)
for k in range(4):
    t0 = time.perf_counter(); o.accumulate(float(k), c); ctx.synchronize(); t1 = time.perf_counter()
    c.fix_positions(0); ctx.synchronize(); t2 = time.perf_counter()
    print(f"n={n:.0e} orient accumulate {1e3*(t1-t0):.2f} ms, fix_positions {1e3*(t2-t1):.2f} ms, used {o.currentUsed()}")
print({k: round(v["ms_total"] / max(1, v["launches"]), 3) for k, v in ctx.profile_report().items()})
