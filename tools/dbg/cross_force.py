# The halo basis' force on the 1e7 disk particles of config 4 (the largest single launch of a master step): general
# pass with global row gathers (EXP_AMD_STAGE_ROWS=0) against the LDS-staged rows (k_sph_force_staged).
#   python tools/dbg/cross_force.py [rows ...]
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import make_disk
from exp_amd.models import NFWModel
from exp_amd.runtime import Component, Context, Cylinder, SphereSL
from exp_amd.empcyl import build_empcyl
from exp_amd.slgrid import build_slgrid
device = torch.device("cuda", 0)
ts = torch.cuda.Stream(device); torch.cuda.set_stream(ts)
ctx = Context(0, stream=ts.cuda_stream)
n, a, h, scale = 10_000_000, 0.01, 0.001, 0.1
model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
lmax, nmax = (int(os.environ.get("LMAX", 6)), int(os.environ.get("NMAX", 18)))
g = build_slgrid(model, lmax, nmax, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
cg = build_empcyl(mmax=6, norder=12, numx=256, numy=128, acyl=a, hcyl=h, lmaxfid=16, nmaxfid=12, numr=800, rnum=100, tnum=40)
X, Y, Z, vx, vy, vz = make_disk(n, a, h, 34567, device, vscale=7.0)
mass = torch.full((n,), 0.1 / n, device=device, dtype=torch.float64)
disk = Component(ctx, n); disk.upload_device(mass, X, Y, Z, vx, vy, vz)
fc = Cylinder(ctx, cg)
fc.determine_coefficients(disk)                       # the disk in ITS basis' cell order
fh = SphereSL(ctx, g, scale=scale, rmin=g.rmin * scale, rmax=g.rmax * scale)
import numpy as np
fh.set_coefs(np.random.default_rng(1).standard_normal((fh.nrows, fh.nmax)) * 1e-3)
for rows in [int(v) for v in sys.argv[1:]] or [0, 32]:
    os.environ["EXP_AMD_STAGE_ROWS"] = str(rows)
    for _ in range(3):
        fh.get_acceleration_and_potential(disk, external=True)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fh.get_acceleration_and_potential(disk, external=True)
    ctx.synchronize()
    print(f"stage_rows {rows:3d}: {(time.perf_counter() - t0) * 100:.3f} ms per evaluation of {n:.0e} disk particles (lmax {lmax})")
ctx.close()
