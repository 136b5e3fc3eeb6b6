# per-phase s_memtime counters of k_sph_accumulate<10> (tools/build_variant_l10.sh stiming -DEXPT_TIMING)
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["EXP_AMD_LIB"] = os.path.join(ROOT, "exp_amd", "libexp_amd_stiming.so")
import torch
from bench import make_halo
from exp_amd.models import NFWModel
from exp_amd.runtime import Component, Context, SphereSL
from exp_amd.slgrid import build_slgrid
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
device = torch.device("cuda", 0)
ts = torch.cuda.Stream(device); torch.cuda.set_stream(ts)
ctx = Context(0, stream=ts.cuda_stream)
model = NFWModel(rs=1.0, rtrunc=20.0, wtrunc=6.0, rmin=1e-3, rmax=50.0)
grid = build_slgrid(model, 10, 24, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0)
x, y, z, vx, vy, vz = make_halo(model, n, 23456, device)
mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
c = Component(ctx, n); c.upload_device(mass, x, y, z, vx, vy, vz)
del x, y, z, vx, vy, vz, mass
f = SphereSL(ctx, grid)
f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
for _ in range(3): f.step_kdk(c, 0.002)
ctx.synchronize()
raw = ctypes.CDLL(os.environ["EXP_AMD_LIB"])
raw.exp_amd_debug_sph_zero()
ctx.profile(True); ctx.profile_reset()
for _ in range(4): f.step_kdk(c, 0.002)
ctx.synchronize()
prof = ctx.profile_report(); ctx.profile(False)
print({k: round(v['ms_total'] / max(1, v['launches']), 4) for k, v in prof.items() if v['launches']})
out = (ctypes.c_ulonglong * 16)()
raw.exp_amd_debug_sph_read(out)
for name, o in (("wave of m-range 0", 0), ("waves of the other m-ranges", 8)):
    tl, ti, tb, tr, ta, nw, nt = [out[o + k] for k in range(7)]
    print(f"{name}: {nw} waves, {nt} tiles; per tile ticks: wait for loads {tl/nt:.0f}, inputs {ti/nt:.0f}, barrier {tb/nt:.0f}, "
          f"four groups {tr/nt:.0f}; whole wave {ta/nw:.0f} ({100*tl/ta:.0f}% / {100*ti/ta:.0f}% / {100*tb/ta:.0f}% / {100*tr/ta:.0f}%)")
