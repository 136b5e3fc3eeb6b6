"""Accumulation of ONE un-sorted (sparse) multistep level through the staged per-particle path, against the dense kernel
on the same particles cell-sorted (tools/dbg/acc_small.py).   python tools/dbg/acc_staged.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_halo
from exp_amd.models import NFWModel
from exp_amd.runtime import Component, Context, Simulation, SphereSL
from exp_amd.slgrid import build_slgrid
device = torch.device("cuda", 0)
ts = torch.cuda.Stream(device); torch.cuda.set_stream(ts)
ctx = Context(0, stream=ts.cuda_stream)
model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
for lmax, nmax in ((6, 18), (10, 24)):
    g = build_slgrid(model, lmax, nmax, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
    for n in (20_000, 120_000, 250_000, 1_000_000):
        for dense_min in (0, 10**9):
            ctx.set_dense_min(dense_min)
            x, y, z, vx, vy, vz = make_halo(model, n, 23456, device)
            mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
            c = Component(ctx, n); c.upload_device(mass, x, y, z, vx, vy, vz)
            f = SphereSL(ctx, g, multistep=1)
            sim = Simulation(ctx, 4e-4, multistep=1, dynfrac=(1e3, 1e3, 1e3, 1e3, 1e3))
            sim.add_component(c, f); sim.init(); sim.step(2)
            ctx.profile(True); ctx.profile_reset(); sim.step(5); ctx.synchronize()
            r = ctx.profile_report(); ctx.profile(False)
            print(f"S{lmax} n={n:8d} dense_min={dense_min:10d}", {k: (round(1e3 * v["ms_total"] / v["launches"], 1), v["launches"]) for k, v in r.items() if v["launches"] and ("accum" in k or "force" in k or "scatter" in k or "key_hist" in k)}, flush=True)
            sim.close(); c.close(); f.close()
