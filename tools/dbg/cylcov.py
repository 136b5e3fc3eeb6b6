import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from exp_amd.runtime import Component, Context, Cylinder
from tests.oracle_lib import Oracle
from tests.test_cyl_gpu import cyl_grid, _disk
ctx = Context(0); oracle = Oracle()
g = cyl_grid(4, 6)
n, sampT = 12000, 7
m, pos, _ = _disk(n, 97, g)
g2 = copy.copy(g); g2.tab = g.tab.copy()
rng = np.random.default_rng(5)
g2.tab[3] = g.tab[0] * (1.0 + 0.5 * rng.standard_normal(g.tab[0].shape))
f2 = Cylinder(ctx, g2); f2.cov_enable(sampT)
c = Component.from_arrays(ctx, m, pos)
f2.cov_accumulate(c)
ref = oracle.cyl_covariance(g2, pos, m, sampT); got = f2.cov_get()
d = got["covr"] - ref["covr"]
print("re err", np.abs(d.real).max(), "im err", np.abs(d.imag).max(), "scale", np.abs(ref["covr"]).max(), np.abs(ref["covr"].imag).max())
for mm in range(5):
    print(mm, np.abs(d[:, mm].real).max(), np.abs(d[:, mm].imag).max(), np.abs(ref["covr"][:, mm].imag).max())
print(got["covr"][0, 1, :2, :2], ref["covr"][0, 1, :2, :2])
