"""k_sph_accumulate / k_cyl_accumulate of the same component: fused single-level step vs the multistep engine (all particles
on level 0 at first).   python tools/dbg/acc_ms_vs_fused.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import make_halo, make_disk
from exp_amd.empcyl import build_empcyl
from exp_amd.models import NFWModel
from exp_amd.runtime import Component, Context, Cylinder, Simulation, SphereSL
from exp_amd.slgrid import build_slgrid
n = 10_000_000
device = torch.device("cuda", 0)
ts = torch.cuda.Stream(device); torch.cuda.set_stream(ts)
ctx = Context(0, stream=ts.cuda_stream)
model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
g = build_slgrid(model, 6, 18, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
cg = build_empcyl(mmax=6, norder=12, numx=256, numy=128, acyl=0.01, hcyl=0.001, lmaxfid=16, nmaxfid=12, numr=800, rnum=100, tnum=40)
def prof(fn, tag):
    ctx.profile(True); ctx.profile_reset(); fn(); ctx.synchronize()
    r = ctx.profile_report(); ctx.profile(False)
    print(tag, {k: (round(v["ms_total"], 3), v["launches"]) for k, v in r.items() if v["launches"] and ("accum" in k or "scatter" in k or k.endswith("force"))}, flush=True)
for kind in ("sphere", "cylinder"):
    for ms in (0, 4):
        if kind == "sphere":
            x, y, z, vx, vy, vz = make_halo(model, n, 23456, device)
            mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
            f = SphereSL(ctx, g, multistep=ms)
        else:
            x, y, z, vx, vy, vz = make_disk(n, 0.01, 0.001, 34567, device, vscale=7.0)
            mass = torch.full((n,), 0.1 / n, device=device, dtype=torch.float64)
            f = Cylinder(ctx, cg, multistep=ms)
        c = Component(ctx, n); c.upload_device(mass, x, y, z, vx, vy, vz)
        if ms == 0:
            f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
            for _ in range(3): f.step_kdk(c, 4e-4)
            prof(lambda: f.step_kdk(c, 4e-4), f"{kind} fused step")
        else:
            sim = Simulation(ctx, 4e-4, multistep=ms, dynfrac=(1e3, 1e3, 1e3, 1e3, 1e3))      # nobody leaves level 0
            sim.add_component(c, f); sim.init(); sim.step(2)
            prof(lambda: sim.step(1), f"{kind} multistep {ms}, one master step")
            print("   levels", np.bincount(c.download_levels(), minlength=ms + 1).tolist())
            sim.close()
        c.close(); f.close()
