#!/usr/bin/env python3
"""Experiment (GPU): does an HBM-streaming kernel with a small, persistent footprint run beside the
VALU-bound force / accumulate passes without slowing them?  tools/dbg/corun.hip is the copy kernel.

    python tools/dbg/corun.py [n=1e8]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from bench import make_halo
    from exp_amd.models import NFWModel
    from exp_amd.runtime import Component, Context, SphereSL
    from exp_amd.slgrid import build_slgrid
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    lib = ctypes.CDLL(os.path.join(ROOT, "build", "libcorun.so"))
    lib.corun_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    device = torch.device("cuda", 0)
    model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
    grid = build_slgrid(model, 10, 24, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
    x, y, z, vx, vy, vz = make_halo(model, n, seed=23456, device=device)
    mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
    sA, sB = torch.cuda.Stream(device), torch.cuda.Stream(device)
    torch.cuda.set_stream(sA)
    ctx = Context(0, stream=sA.cuda_stream)
    comp = Component(ctx, n)
    comp.upload_device(mass, x, y, z, vx, vy, vz)
    del x, y, z, vx, vy, vz, mass
    force = SphereSL(ctx, grid)
    for _ in range(3):
        force.step_kdk(comp, 0.002)          # sorted state
    torch.cuda.synchronize()
    nb = 4 << 30                               # 4 GiB copied per call = 8 GiB of traffic
    src = torch.empty(nb, dtype=torch.uint8, device=device)
    dst = torch.empty(nb, dtype=torch.uint8, device=device)
    src.zero_(); dst.zero_()
    torch.cuda.synchronize()

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def run_force(reps):
        for _ in range(reps):
            comp.zero_acceleration(0)
            force.get_acceleration_and_potential(comp)

    def run_acc(reps):
        for _ in range(reps):
            force.determine_coefficients(comp)

    def run_copy(reps, blocks, unroll):
        for _ in range(reps):
            lib.corun_copy(src.data_ptr(), dst.data_ptr(), nb, blocks, unroll, sB.cuda_stream)

    def timed(fa, fb):
        a0, a1, b0, b1 = ev(), ev(), ev(), ev()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if fa:
            a0.record(sA); fa(); a1.record(sA)
        if fb:
            b0.record(sB); fb(); b1.record(sB)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) * 1e3
        return el, (a0.elapsed_time(a1) if fa else 0.0), (b0.elapsed_time(b1) if fb else 0.0)

    R = 4
    for name, fn in (("force", run_force), ("accumulate", run_acc)):
        timed(lambda: fn(1), None)
        el, ta, _ = timed(lambda: fn(R), None)
        print(f"{name} alone: {ta / R:.3f} ms per pass")
        base = ta / R
        for blocks, unroll in ((256, 8), (512, 8), (1024, 8), (2048, 8), (512, 16), (1024, 4), (8192, 4)):
            timed(None, lambda: run_copy(1, blocks, unroll))
            _, _, tb = timed(None, lambda: run_copy(2, blocks, unroll))
            bw = 2 * 2 * nb / (tb * 1e-3) / 1e12
            # co-run: copies sized to cover the compute passes
            ncopy = max(1, int(round(base * R / (tb / 2))))
            el, ta, tb2 = timed(lambda: fn(R), lambda: run_copy(ncopy, blocks, unroll))
            print(f"  copy blocks={blocks:5d} unroll={unroll:2d}: alone {bw:.2f} TB/s ({tb / 2:.3f} ms per 8 GiB) | co-run "
                  f"{name} {ta / R:.3f} ms (x{ta / R / base:.2f}), {ncopy} copies {tb2 / ncopy:.3f} ms each "
                  f"(x{tb2 / ncopy / (tb / 2):.2f}), wall {el:.2f} ms vs serial {base * R + ncopy * tb / 2:.2f}")
    ctx.close()


if __name__ == "__main__":
    main()
