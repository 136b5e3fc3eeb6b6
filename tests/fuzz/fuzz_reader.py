"""Randomised campaign for the phase-space files (CPU): random component sets (1-4 components, 0-600 particles, 0-3 integer
and real attributes, indexed or not, names with spaces, info stanzas around the 1024-byte limit), both precisions, 1-5
ranks.  Per trial: `write_psp` must be byte-identical to the oracle's writer; `PSPout` on the oracle's file must give the
oracle reader's arrays for every rank; `write_spl` -> `PSPspl` and `write_psp_hdf5` -> `PSPhdf5` (both layouts) must give
what `PSPout` gives.    python tests/fuzz/fuzz_reader.py [trials=200] [seed=1]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from exp_amd import reader as R
from tests.oracle_lib import Oracle

_argv = sys.argv if __name__ == "__main__" else [""]   # imported by tests/test_fuzz_cpu.py: defaults
trials = int(_argv[1]) if len(_argv) > 1 else 200
seed0 = int(_argv[2]) if len(_argv) > 2 else 1
orc = Oracle()
try:
    from exp_amd import reader_h5 as H
    H._h5()
except (RuntimeError, OSError):
    H = None
tmp = tempfile.mkdtemp(prefix="fuzz_reader_")


def one(t, rng):
    ncomp = int(rng.integers(1, 5))
    real4 = bool(rng.random() < 0.5)
    comps = []
    for k in range(ncomp):
        n = int(rng.choice([0, 1, 2, 63, 64, 65, 600]))
        ni, nd = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        idx = bool(rng.random() < 0.5)
        name = ["dark", "star disk", "gas", "bulge 2"][k]
        pad = int(rng.choice([0, 0, 900, 960, 1100]))
        c = dict(info=R.component_info(name, "sphereSL", {"note": "x" * pad} if pad else {"Lmax": 4},
                                       {"nlevel": 1, "indexing": idx}),
                 indexing=idx, mass=rng.uniform(0.1, 2, n), pos=rng.normal(size=(n, 3)) * 10.0 ** rng.integers(-3, 4),
                 vel=rng.normal(size=(n, 3)), pot=rng.normal(size=n), potext=rng.normal(size=n))
        if ni:
            c["iattrib"] = rng.integers(-2 ** 31, 2 ** 31 - 1, (n, ni)).astype(np.int32)
        if nd:
            c["dattrib"] = rng.normal(size=(n, nd)) * 1e3
        if idx:
            c["indx"] = rng.integers(0, 2 ** 62, n).astype(np.uint64)
        c["name"], c["force"], c["fconf"] = name, "sphereSL", "{Lmax: 4}"
        comps.append(c)
    a, b = os.path.join(tmp, f"a{t}"), os.path.join(tmp, f"b{t}")
    R.write_psp(a, 0.25 * t, comps, real4)
    orc.psp_write(b, 0.25 * t, comps, real4)
    bad = []
    if open(a, "rb").read() != open(b, "rb").read():
        bad.append("writer bytes")
    rd = R.PSPout([b])
    numprocs = int(rng.integers(1, 6))
    for myid in range(numprocs):
        _, _, ref = orc.psp_read(b, [c["indexing"] for c in comps], numprocs, myid)
        rd.numprocs, rd.myid = numprocs, myid
        for c, s in zip(comps, ref):
            rd.SelectType(c["name"])
            got = rd.arrays()
            for key in ("indx", "mass", "pos", "vel", "pot"):
                if not np.array_equal(got[key], s[key]):
                    bad.append(f"reader rank {myid}/{numprocs} {c['name']} {key}")
            if s["niatr"] and len(s["mass"]) and not np.array_equal(got["iattrib"], s["iattrib"]):
                bad.append(f"reader {c['name']} iattrib")
            if s["ndatr"] and len(s["mass"]) and not np.array_equal(got["dattrib"], s["dattrib"]):
                bad.append(f"reader {c['name']} dattrib")
    rd.numprocs, rd.myid = 1, 0
    m = os.path.join(tmp, f"SPL.{t}")
    R.write_spl(m, 0.25 * t, comps, nparts=int(rng.integers(1, 5)), real4=real4)
    rs = R.PSPspl([m])
    for c in comps:
        rd.SelectType(c["name"]); rs.SelectType(c["name"])
        for key, v in rd.arrays().items():
            if not np.array_equal(v, rs.arrays()[key]):
                bad.append(f"spl {c['name']} {key}")
    if H is not None:
        for g4 in (False, True):
            hp = os.path.join(tmp, f"h{t}_{int(g4)}.1")
            # (the HDF5 layouts always hold an index; the compound's id is a 32-bit integer)
            hc = [dict(c, indx=(c["indx"] % 2 ** 31) if c.get("indx") is not None else np.arange(len(c["mass"]), dtype=np.uint64))
                  for c in comps]
            H.write_psp_hdf5(hp, 0.25 * t, hc, real4=real4, gadget4=g4, chunk=int(rng.choice([0, 32])), compress=int(rng.choice([0, 4])))
            rh = R.ParticleReader.createReader("PSPhdf5", [hp])
            for c, c0 in zip(hc, comps):
                rd.SelectType(c["name"]); rh.SelectType(c["name"])
                got, ref = rh.arrays(), rd.arrays()
                for key in ("pos", "vel") + (("iattrib",) if "iattrib" in ref else ()) + (("dattrib",) if "dattrib" in ref else ()):
                    if len(ref["mass"]) and not np.array_equal(got[key], ref[key]):
                        bad.append(f"hdf5 g4={g4} {c['name']} {key}")
                f = (lambda x: np.asarray(x).astype(np.float32).astype(np.float64)) if real4 else np.asarray
                if not np.array_equal(got["pot"], f(c0["pot"])) or not np.array_equal(got["potext"], f(c0["potext"])):
                    bad.append(f"hdf5 g4={g4} {c['name']} pot")
                if not np.array_equal(got["indx"], c["indx"]):
                    bad.append(f"hdf5 g4={g4} {c['name']} indx")
    for fn in os.listdir(tmp):
        os.remove(os.path.join(tmp, fn))
    print(f"{t:3d} ncomp {ncomp} real4 {real4} n {[len(c['mass']) for c in comps]} ranks {numprocs}: {'ok' if not bad else 'MISMATCH ' + '; '.join(bad[:4])}",
          flush=True)
    return not bad


def main():
    t0 = time.time()
    bad = sum(0 if one(t, np.random.default_rng([seed0, t])) else 1 for t in range(trials))
    print(f"{trials} trials, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
