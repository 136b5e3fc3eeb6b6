"""config 3 (1e7 exponential disk, C6 on the 256 x 128 grid) stepped with the fused step: 12 steps, for a kernel trace
(rocprofv3 --kernel-trace --stats -- python3 tools/dbg/cyl_app_probe.py [append_min])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import make_disk
from exp_amd.empcyl import build_empcyl
from exp_amd.runtime import Component, Context, Cylinder
device = torch.device("cuda", 0)
ts = torch.cuda.Stream(device); torch.cuda.set_stream(ts)
ctx = Context(0, stream=ts.cuda_stream)
ctx.set_append_min(int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 20)
n, a, h = 10_000_000, 0.01, 0.001
cg = build_empcyl(mmax=6, norder=12, numx=256, numy=128, acyl=a, hcyl=h, lmaxfid=32, nmaxfid=24, numr=2000, rnum=200, tnum=80)
X, Y, Z, vx, vy, vz = make_disk(n, a, h, 34567, device, vscale=7.0)
mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
c = Component(ctx, n); c.upload_device(mass, X, Y, Z, vx, vy, vz)
f = Cylinder(ctx, cg)
f.determine_coefficients(c); c.zero_acceleration(); f.get_acceleration_and_potential(c)
for _ in range(4): f.step_kdk(c, 2e-5)
ctx.synchronize()
t0 = time.time()
for _ in range(20): f.step_kdk(c, 2e-5)
ctx.synchronize()
print("ms per step", (time.time() - t0) / 20 * 1e3, flush=True)
