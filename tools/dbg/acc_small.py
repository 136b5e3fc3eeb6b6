"""Time of k_sph_accumulate on SMALL components (the dense kernel on a thinly populated multistep level): n particles of
an NFW halo, S6, single level.   python tools/dbg/acc_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_halo
from exp_amd.models import NFWModel
from exp_amd.runtime import Component, Context, SphereSL
from exp_amd.slgrid import build_slgrid
device = torch.device("cuda", 0)
ts = torch.cuda.Stream(device); torch.cuda.set_stream(ts)
ctx = Context(0, stream=ts.cuda_stream)
model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
for lmax, nmax in ((6, 18), (10, 24)):
    g = build_slgrid(model, lmax, nmax, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
    f = SphereSL(ctx, g)
    for n in (20_000, 120_000, 250_000, 1_000_000, 10_000_000):
        x, y, z, vx, vy, vz = make_halo(model, n, 23456, device)
        mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
        c = Component(ctx, n); c.upload_device(mass, x, y, z, vx, vy, vz)
        for _ in range(3): f.determine_coefficients(c)
        ctx.synchronize()
        ctx.profile(True); ctx.profile_reset()
        for _ in range(10): f.determine_coefficients(c)
        ctx.synchronize()
        r = ctx.profile_report(); ctx.profile(False)
        print(f"S{lmax} n={n:9d}", {k: round(1e3 * v["ms_total"] / v["launches"], 1) for k, v in r.items() if v["launches"] and ("accum" in k or "contract" in k)}, "us", flush=True)
        c.close()
    f.close()
