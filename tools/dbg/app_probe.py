"""per-step kernel times of the fused step at bench scale (1e8 NFW, S10), append form on: step 0 ordinary, step 1 the entry
(ordinary scatter into the regions + the first placing force pass), steps 2.. the append form"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from exp_amd.models import NFWModel
from exp_amd.runtime import Component, Context, SphereSL
from exp_amd.slgrid import build_slgrid
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
grid = build_slgrid(model, 10, 24, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
x, y, z, vx, vy, vz = bench.make_halo(model, n, 23456, dev)
mass = torch.full((n,), 1.0 / n, device=dev, dtype=torch.float64)
ts = torch.cuda.Stream(dev); torch.cuda.set_stream(ts)
ctx = Context(0, stream=ts.cuda_stream)
ctx.set_append_min(1 << 20)
c = Component(ctx, n); c.upload_device(mass, x, y, z, vx, vy, vz)
f = SphereSL(ctx, grid)
f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
for k in range(nst):
    ctx.profile(True); ctx.profile_reset()
    f.step_kdk(c, 0.002)
    rep = ctx.profile_report()
    print(k, {a: round(b["ms_total"], 3) for a, b in rep.items() if b["launches"]}, flush=True)
    ctx.profile(False)
