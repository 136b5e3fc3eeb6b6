#!/usr/bin/env python3
"""Secondary configurations of BASELINE.json (configs 2-4) on one GPU; one JSON line each.

    python tools/bench_configs.py [--n 1e7] [--steps 10]

config 2: 1e7 NFW halo, SphericalSL lmax 6 nmax 18            (fused KDK step)
config 3: 1e7 exponential disk, EmpCylSL mmax 6 nmax 12        (fused KDK step)
config 4: disk + halo (1e7 each), both self and both cross forces, multistep 4 (C++ step driver);
          reported as master-step-equivalents (N particles x 1 per dtime) and raw sub-steps.
These are parity-test configurations, not the headline bench line (bench.py)."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def disk_device(n, a, h, seed, device, vscale):
    import torch
    gen = torch.Generator(device=device).manual_seed(seed)
    f64 = torch.float64
    x = torch.linspace(0.0, 12.0, 16384, device=device, dtype=f64)
    cdf = 1.0 - (1.0 + x) * torch.exp(-x)
    cdf = cdf / cdf[-1]
    u = torch.rand(n, device=device, dtype=f64, generator=gen)
    idx = torch.searchsorted(cdf, u).clamp_(1, 16383)
    w = (u - cdf[idx - 1]) / (cdf[idx] - cdf[idx - 1])
    R = a * (x[idx - 1] + w * (x[idx] - x[idx - 1]))
    ph = torch.rand(n, device=device, dtype=f64, generator=gen) * (2 * math.pi)
    uz = torch.rand(n, device=device, dtype=f64, generator=gen).clamp_(1e-12, 1 - 1e-12)
    z = h * torch.atanh(2 * uz - 1)
    X, Y = (R * torch.cos(ph)).contiguous(), (R * torch.sin(ph)).contiguous()
    # roughly circular orbits + dispersion so that the cell order really changes every step
    vc = vscale * torch.sqrt(R / (R + a))
    vx = (-vc * torch.sin(ph) + 0.1 * vscale * torch.randn(n, device=device, dtype=f64, generator=gen)).contiguous()
    vy = (vc * torch.cos(ph) + 0.1 * vscale * torch.randn(n, device=device, dtype=f64, generator=gen)).contiguous()
    vz = (0.05 * vscale * torch.randn(n, device=device, dtype=f64, generator=gen)).contiguous()
    return X, Y, z.contiguous(), vx, vy, vz


def timed(fn, steps, warmup):
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--only", type=int, default=0)
    args = ap.parse_args()
    import torch
    from bench import make_halo
    from exp_amd.empcyl import build_empcyl
    from exp_amd.models import NFWModel
    from exp_amd.runtime import Component, Context, Cylinder, Simulation, SphereSL
    from exp_amd.slgrid import build_slgrid
    n = int(args.n)
    device = torch.device("cuda", 0)
    ts = torch.cuda.Stream(device)
    torch.cuda.set_stream(ts)
    ctx = Context(0, stream=ts.cuda_stream)
    model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
    grid = build_slgrid(model, 6, 18, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)

    def halo(scale=1.0, vfac=1.0, mult=0):
        x, y, z, vx, vy, vz = make_halo(model, n, 23456, device)
        mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
        c = Component(ctx, n)
        c.upload_device(mass, (x * scale).contiguous(), (y * scale).contiguous(), (z * scale).contiguous(),
                        (vx * vfac).contiguous(), (vy * vfac).contiguous(), (vz * vfac).contiguous())
        f = SphereSL(ctx, grid, scale=scale, rmin=grid.rmin * scale, rmax=grid.rmax * scale, multistep=mult)
        return c, f

    out = []
    if args.only in (0, 2):
        c, f = halo()
        f.determine_coefficients(c); c.zero_acceleration(); f.get_acceleration_and_potential(c)
        dt = timed(lambda: f.step_kdk(c, 0.002), args.steps, args.warmup)
        ctx.profile(True); ctx.profile_reset()
        for _ in range(4):
            f.step_kdk(c, 0.002)
        prof = {k: round(v["ms_total"] / 4, 3) for k, v in ctx.profile_report().items()}
        ctx.profile(False)
        out.append({"config": "2: 1e7 NFW halo, SphericalSL lmax 6 nmax 18", "n": n,
                    "ms_per_step": 1e3 * dt, "particle_steps_per_s": n / dt,
                    "hbm_frac_232B": 232.0 * n / dt / 8e12, "kernels_ms_per_step": prof})
        c.close(); f.close()
    a, h = 0.01, 0.001
    if args.only in (0, 3, 4):
        cg = build_empcyl(mmax=6, norder=12, numx=256, numy=128, acyl=a, hcyl=h, lmaxfid=32,
                          nmaxfid=24, numr=2000, rnum=200, tnum=80)
    if args.only in (0, 3):
        X, Y, Z, vx, vy, vz = disk_device(n, a, h, 34567, device, vscale=7.0)
        mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
        c = Component(ctx, n)
        c.upload_device(mass, X, Y, Z, vx, vy, vz)
        f = Cylinder(ctx, cg)
        f.determine_coefficients(c); c.zero_acceleration(); f.get_acceleration_and_potential(c)
        dt = timed(lambda: f.step_kdk(c, 2e-5), args.steps, args.warmup)
        ctx.profile(True); ctx.profile_reset()
        for _ in range(4):
            f.step_kdk(c, 2e-5)
        prof = {k: round(v["ms_total"] / 4, 3) for k, v in ctx.profile_report().items()}
        ctx.profile(False)
        out.append({"config": "3: 1e7 exponential disk, EmpCylSL mmax 6 nmax 12 (256x128 grid)",
                    "n": n, "ms_per_step": 1e3 * dt, "particle_steps_per_s": n / dt,
                    "hbm_frac_232B": 232.0 * n / dt / 8e12, "kernels_ms_per_step": prof})
        c.close(); f.close()
    if args.only in (0, 4):
        ms = 4
        ch, fh = halo(scale=0.1, vfac=math.sqrt(10.0), mult=ms)      # a / rs = 0.1
        X, Y, Z, vx, vy, vz = disk_device(n, a, h, 34567, device, vscale=7.0)
        mass = torch.full((n,), 0.1 / n, device=device, dtype=torch.float64)
        cd = Component(ctx, n)
        cd.upload_device(mass, X, Y, Z, vx, vy, vz)
        fd = Cylinder(ctx, cg, multistep=ms)
        sim = Simulation(ctx, 4e-4, multistep=ms)
        i1, i2 = sim.add_component(ch, fh), sim.add_component(cd, fd)
        sim.add_interaction(i1, i2)
        sim.add_interaction(i2, i1)
        sim.init()
        lev_h = np.bincount(ch.download_levels(), minlength=ms + 1)
        lev_d = np.bincount(cd.download_levels(), minlength=ms + 1)
        dt = timed(lambda: sim.step(1), max(2, args.steps // 3), 1)
        ctx.profile(True); ctx.profile_reset()
        sim.step(1)
        prof = {k: round(v["ms_total"], 3) for k, v in ctx.profile_report().items()}
        ctx.profile(False)
        sub = sum(int(lev_h[M] + lev_d[M]) * (1 << M) for M in range(ms + 1))
        out.append({"config": "4: disk+halo (1e7 each), SphericalSL+EmpCylSL, multistep 4, cross forces",
                    "n": 2 * n, "ms_per_master_step": 1e3 * dt,
                    "master_step_particle_steps_per_s": 2 * n / dt,
                    "raw_particle_substeps_per_s": sub / dt,
                    "levels_halo": lev_h.tolist(), "levels_disk": lev_d.tolist(),
                    "level_switches_per_master_step": sim.step_switches,
                    "kernels_ms_per_master_step": prof})
    for o in out:
        print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
