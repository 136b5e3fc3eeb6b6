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
