# reads the per-phase cycle counters of the EXPT_TIMING build of k_cyl_accumulate (tools/build_variant_cyl.sh timing -DEXPT_TIMING)
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["EXP_AMD_LIB"] = os.path.join(ROOT, "exp_amd", "libexp_amd_timing.so")
import torch
from bench import make_disk
from exp_amd.empcyl import build_empcyl
from exp_amd.runtime import Component, Context, Cylinder
from exp_amd import _lib
device = torch.device("cuda", 0)
ts = torch.cuda.Stream(device); torch.cuda.set_stream(ts)
ctx = Context(0, stream=ts.cuda_stream)
lib = ctx.lib
n, a, h = 10_000_000, 0.01, 0.001
cg = build_empcyl(mmax=6, norder=12, numx=256, numy=128, acyl=a, hcyl=h, lmaxfid=32, nmaxfid=24, numr=2000, rnum=200, tnum=80)
X, Y, Z, vx, vy, vz = make_disk(n, a, h, 34567, device, vscale=7.0)
mass = torch.full((n,), 1.0 / n, device=device, dtype=torch.float64)
c = Component(ctx, n); c.upload_device(mass, X, Y, Z, vx, vy, vz)
f = Cylinder(ctx, cg)
f.determine_coefficients(c); c.zero_acceleration(); f.get_acceleration_and_potential(c)
for _ in range(3): f.step_kdk(c, 2e-5)
ctx.synchronize()
raw = ctypes.CDLL(os.environ["EXP_AMD_LIB"])
raw.exp_amd_debug_zero()
ctx.profile(True); ctx.profile_reset()
for _ in range(4): f.step_kdk(c, 2e-5)
ctx.synchronize()
prof = ctx.profile_report(); ctx.profile(False)
print({k: round(v['ms_total'] / max(1, v['launches']), 4) for k, v in prof.items() if v['launches']})
out = (ctypes.c_ulonglong * 8)()
raw.exp_amd_debug_read(out)
t_load, t_prep, t_red, t_all, nw, ng = [out[k] for k in range(6)]
nfl, nw = nw >> 32, nw & 0xffffffff
print(f"flushes {nfl} ({nfl / ng:.2f} per group): {out[6] / max(1, nfl):.0f} ticks each + {out[7] / max(1, nfl):.0f} until every outstanding memory operation is back")
print(f"waves {nw}, groups {ng}: per group cycles (s_memtime): wait-for-loads {t_load/ng:.0f}, prepare {t_prep/ng:.0f}, reduce {t_red/ng:.0f}; per wave total {t_all/nw:.0f}")
