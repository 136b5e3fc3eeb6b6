"""Where createFromArray spends its time (1e7 particles): host transform, column copies, upload, accumulation, read-back."""
import cProfile
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from exp_amd.basis import Basis

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
tmp = tempfile.mkdtemp()
basis = Basis.factory(f"""
id : sphereSL
parameters :
  numr: 2000
  rmin: 0.0001
  rmax: 1.95
  Lmax: 6
  nmax: 18
  rmapping : 0.0667
  modelname: {os.path.join(gold, 'SLGridSph.model')}
  cachename: {os.path.join(tmp, 'sl.cache')}
""")
rng = np.random.default_rng(1)
m, pos = np.full(n, 1.0 / n), rng.normal(0, 0.3, (n, 3))
basis.createFromArray(m[:1000], pos[:1000])
for label, p in (("[N,3]", pos), ("[3,N]", np.ascontiguousarray(pos.T))):
    t = time.time(); basis.createFromArray(m, p); dt = time.time() - t
    print(f"{label}: {dt:.3f} s = {n / dt:.2e} particles/s")
cProfile.run("basis.createFromArray(m, pos)", "/tmp/create.prof")
pstats.Stats("/tmp/create.prof").sort_stats("cumtime").print_stats(14)

# ---- the evaluation side: getAccel and getFields on many points
basis.set_coefs(basis.createFromArray(m, pos))
for k in (1_000_000, 10_000_000):
    q = pos[:k]
    basis.getAccel(q[:1000])
    t = time.time(); acc = basis.getAccel(q); dt = time.time() - t
    print(f"getAccel {k:.0e} points: {dt:.3f} s = {k / dt:.2e} points/s")
x, y, z = np.ascontiguousarray(pos[:1_000_000].T)
basis.getFields(x[:10], y[:10], z[:10])
t = time.time(); f = basis.getFields(x, y, z); dt = time.time() - t
print(f"getFields 1e6 points: {dt:.3f} s = {1e6 / dt:.2e} points/s")
cProfile.run("basis.getAccel(pos)", "/tmp/acc.prof")
pstats.Stats("/tmp/acc.prof").sort_stats("cumtime").print_stats(12)
