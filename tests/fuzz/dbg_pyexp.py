import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from exp_amd.basis import Basis
from tests.oracle_lib import Oracle
GOLD = os.path.join(ROOT, "tests", "golden")
cfg = f"""
id : sphereSL
parameters :
  numr: 1000
  rmin: 0.0001
  rmax: 1.95
  Lmax: 2
  nmax: 10
  rmapping : 0.0667
  modelname: {GOLD}/SLGridSph.model
  cachename: /tmp/dbg.cache
"""
basis = Basis.factory(cfg)
oracle = Oracle()
rng = np.random.default_rng(11)
pos = rng.normal(0, 0.35, (3000, 3)); pos[:40] *= 12.0; pos[40:50] *= 1e-5
m = rng.uniform(0.5, 1.5, 3000) / 3000
coefs = basis.createFromArray(m, pos)
prm = oracle.params(scale=1.0, rmin=basis.rmin, rmax=basis.rmax)
ref, used = oracle.pyexp_sph_accumulate(basis.grid, prm, pos, m)
basis.set_coefs(coefs)
test = np.concatenate([rng.normal(0, 0.5, (300, 3)), rng.normal(0, 3e-5, (20, 3)), rng.normal(0, 4.0, (40, 3)),
                       np.array([[1e-7, 0.0, 0.4], [0.0, -3e-8, -0.9], [1e-9, 1e-9, 1.2]])])
a_ref = oracle.pyexp_sph_accel(basis.grid, prm, ref, test)
acc = basis.getAccel(test)
err = np.linalg.norm(acc - a_ref, axis=1) / np.linalg.norm(a_ref, axis=1)
rt = np.linalg.norm(test, axis=1)
for name, sel in (("bulk", slice(0, 300)), ("inside rmin", slice(300, 320)), ("beyond", slice(320, 360)), ("axis", slice(360, 363))):
    print(name, err[sel].max(), rt[sel].min(), rt[sel].max())
bad = np.argsort(err)[-5:]
for i in bad: print(i, rt[i], test[i], acc[i], a_ref[i])
axis = np.array([[0.0, 0.0, 0.7], [0.0, 0.0, -0.3], [1e-5,0,0.7], [1e-3,0,0.7]])
lit = oracle.pyexp_sph_accel(basis.grid, prm, ref, axis)
got = basis.getAccel(axis)
np.set_printoptions(precision=17)
print(lit); print(got)
prm2 = oracle.params(scale=1.0, rmin=basis.rmin, rmax=basis.rmax)
a2, _ = oracle.sph_accel(basis.grid, prm2, axis, ref)
print("nbody oracle", a2)
