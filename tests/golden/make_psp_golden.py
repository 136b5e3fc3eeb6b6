"""Golden phase-space files written by the REFERENCE's own code: `Particle::writeBinary` and `ComponentHeader::write`
(exputil/Particle.cc, exputil/header.cc compiled in place into oracle/_ref/libref_particle.so) driven in the order of
OutPSN::Run / Component::write_binary.  Run where /root/reference exists:

    python tests/golden/make_psp_golden.py

Writes tests/golden/psp_ref_f8.bin, psp_ref_f4.bin (the files) and psp_ref_inputs.npz (what went in).  The bytes are the
reference's; tests/test_golden_cpu.py reads them with exp_amd.reader and rewrites them with write_psp."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes

from exp_amd import reader as R
from tests.test_ref_particle import LIB, _ref_write

ref = ctypes.CDLL(LIB)
rng = np.random.default_rng(20261003)
here = os.path.dirname(os.path.abspath(__file__))
comps = []
for name, n, idx, ni, nd in (("dark halo", 24, True, 1, 2), ("star disk", 9, False, 0, 1)):
    c = dict(info=R.component_info(name, "sphereSL" if idx else "cylinder", {"Lmax": 2, "nmax": 10} if idx else {"mmax": 4},
                                   {"nlevel": 1, "indexing": idx}, extra={"bodyfile": name.split()[0] + ".bods"}),
             indexing=idx, name=name, mass=rng.uniform(1, 2, n) / n, pos=rng.normal(size=(n, 3)), vel=rng.normal(size=(n, 3)),
             pot=-rng.uniform(0.5, 2, n), potext=rng.normal(size=n) * 0.01,
             iattrib=rng.integers(-1000, 1000, (n, ni)).astype(np.int32), dattrib=rng.normal(size=(n, nd)),
             indx=(rng.permutation(n) + 1).astype(np.uint64) * 11 if idx else np.arange(1, n + 1, dtype=np.uint64))
    comps.append(c)
for real4, tag in ((False, "f8"), (True, "f4")):
    _ref_write(ref, os.path.join(here, f"psp_ref_{tag}.bin"), 0.625, comps, real4)
flat = {}
for k, c in enumerate(comps):
    for key, v in c.items():
        flat[f"c{k}_{key}"] = np.asarray(v)
np.savez(os.path.join(here, "psp_ref_inputs.npz"), **flat)
print("written:", [f for f in os.listdir(here) if f.startswith("psp_ref")])
