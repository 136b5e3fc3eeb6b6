"""createFromReader end to end on a large phase-space file: write a PSP file of N particles (doubles, indexed), then time
the read (PSPout.arrays), the transform + upload + accumulation (createFromReader) and, for comparison, createFromArray on
arrays already in memory.    python tools/dbg/reader_rate.py [N=10000000]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from exp_amd import reader as R
from exp_amd.basis import Basis

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
tmp = tempfile.mkdtemp(prefix="reader_rate_")
cfg = f"""
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
"""
basis = Basis.factory(cfg)
rng = np.random.default_rng(1)
comp = dict(info=R.component_info("dark", "sphereSL", {"Lmax": 6}, {"indexing": True}), mass=np.full(n, 1.0 / n),
            pos=rng.normal(0, 0.3, (n, 3)), vel=rng.normal(0, 0.3, (n, 3)), indx=np.arange(1, n + 1, dtype=np.uint64))
path = os.path.join(tmp, "OUT.big")
t = time.time(); R.write_psp(path, 0.0, [comp]); tw = time.time() - t
size = os.path.getsize(path)
basis.createFromArray(comp["mass"][:1000], comp["pos"][:1000])                     # warm the device path
t = time.time(); rd = R.PSPout([path]); a = rd.arrays(); tr = time.time() - t
t = time.time(); c1 = basis.createFromReader(rd); tc = time.time() - t
t = time.time(); c2 = basis.createFromArray(comp["mass"], comp["pos"]); ta = time.time() - t
err = np.abs(c1.coefs - c2.coefs).max() / np.abs(c2.coefs).max()
print(f"N {n:.1e}, file {size / 1e9:.2f} GB: write {tw:.2f} s ({size / tw / 1e9:.2f} GB/s), read into arrays {tr:.2f} s "
      f"({size / tr / 1e9:.2f} GB/s), createFromReader (arrays cached) {tc:.2f} s = {n / tc:.2e} particles/s, "
      f"createFromArray {ta:.2f} s = {n / ta:.2e} particles/s, coefficient difference {err:.1e}")
os.remove(path)
