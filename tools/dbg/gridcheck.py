import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import threadpoolctl
print(threadpoolctl.threadpool_info())
from exp_amd.models import NFWModel
import exp_amd.slgrid as sg
m = NFWModel(1.0, 20.0, 6.0, 1e-3, 50.0)
res = {}
for lim in (None, 1, 2, 8):
    t0 = time.time()
    if lim is None:
        g = sg.build_slgrid(m, 6, 18, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0, nel=32, P=8)
    else:
        sg.blas_limit = lambda cap=2, _l=lim: threadpoolctl.threadpool_limits(limits=_l)
        g = sg.build_slgrid(m, 6, 18, numr=2000, rmin=1e-3, rmax=49.5, cmap=1, rmap=1.0, nel=32, P=8)
    res[lim] = g
    print(lim, round(time.time() - t0, 2), "s ev0", g.ev[0, :3], "ef", float(np.abs(g.ef).sum()))
for lim in (1, 2, 8):
    print(lim, "max diff vs default", float(np.abs(res[lim].ef - res[None].ef).max()), float(np.abs(res[lim].ev - res[None].ev).max()))
