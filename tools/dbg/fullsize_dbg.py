import os, sys, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tests.conftest import make_grid
from tests.oracle_lib import Oracle
from exp_amd.runtime import Context, Component, SphereSL
from exp_amd.models import sphere_sampling_tables
orc = Oracle(); ctx = Context(0)
model, g = make_grid("nfw", 6, 18, 2000)
n = 2_000_000
gen = torch.Generator(device="cuda").manual_seed(5)
u_tab, r_tab, _ = sphere_sampling_tables(model, 49.0)
u = torch.rand(n, device="cuda", dtype=torch.float64, generator=gen)
ut = torch.tensor(u_tab, device="cuda"); rt = torch.tensor(r_tab, device="cuda")
idx = torch.searchsorted(ut, u).clamp(1, len(u_tab) - 1)
w = (u - ut[idx - 1]) / (ut[idx] - ut[idx - 1])
r = rt[idx - 1] + w * (rt[idx] - rt[idx - 1])
ct = torch.rand(n, device="cuda", dtype=torch.float64, generator=gen) * 2 - 1
ph = torch.rand(n, device="cuda", dtype=torch.float64, generator=gen) * 2 * math.pi
st = torch.sqrt(1 - ct * ct)
print("u_tab", u_tab[:3], u_tab[-3:], "r_tab", r_tab[:3], r_tab[-3:], "r stats", float(r.min()), float(r.median()), float(r.max()))
for sq in (1.0, 0.8):
    x, y, z = (r * st * torch.cos(ph)).contiguous(), (r * st * torch.sin(ph)).contiguous(), (sq * r * ct).contiguous()
    mass = torch.full((n,), 1.0 / n, device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
    f = SphereSL(ctx, g); c = Component(ctx, n)
    c.upload_device(mass, x, y, z)
    f.determine_coefficients(c)
    coef = f.get_coefs()
    c.zero_acceleration(); f.get_acceleration_and_potential(c)
    out = c.download(("acc", "pos"))
    rr = np.linalg.norm(out["pos"], axis=1); sel = (rr > 0.5) & (rr < 5.0)
    arad = -(out["acc"][sel] * out["pos"][sel]).sum(1) / rr[sel]
    ratio = arad * rr[sel] ** 2 / model.mass(rr[sel])
    prm = orc.params(rmin=g.rmin, rmax=g.rmax)
    pos = out["pos"]; m = np.full(n, 1.0 / n)
    cref, used = orc.sph_accumulate(g, prm, pos[:200000], m[:200000] * 10)
    print("squash", sq, "median ratio", np.median(ratio), "c00 gpu", coef[0, :3], "c00 oracle(subset*10)", cref[0, :3], "used", f.Used())
    c.close(); f.close()
