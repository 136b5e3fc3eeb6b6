"""k_sph_accumulate on thinly populated levels: time against the minimum chunk (EXP_AMD_ACC_CHUNK_MIN), S6 numr 2000.
    for c in 64 128 256 512; do EXP_AMD_ACC_CHUNK_MIN=$c python tools/dbg/acc_thin.py; done"""
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
g = build_slgrid(model, 6, 18, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
f = SphereSL(ctx, g)
out = []
for n in (30_000, 120_000, 250_000, 400_000, 1_000_000):
    x, y, z, vx, vy, vz = make_halo(model, n, 23456, device)
    mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
    c = Component(ctx, n); c.upload_device(mass, x, y, z, vx, vy, vz)
    for _ in range(3): f.determine_coefficients(c)
    ctx.synchronize()
    ctx.profile(True); ctx.profile_reset()
    for _ in range(10): f.determine_coefficients(c)
    ctx.synchronize()
    r = ctx.profile_report(); ctx.profile(False)
    out.append((n, round(1e3 * r["k_sph_accumulate"]["ms_total"] / r["k_sph_accumulate"]["launches"], 1)))
    c.close()
print("chunk_min", os.environ.get("EXP_AMD_ACC_CHUNK_MIN", "64"), out, "us", flush=True)
