"""The HDF5 layouts of this package against the reference's WRITER SOURCE: the object names and attribute types of
every file kind on the path -- SLGridSph cache, EmpCylSL cache, spherical / cylindrical coefficient files, the
subsample-covariance store -- are extracted from the `createAttribute<T>("name")`, `createDataSet("name")` and
`createGroup("name")` calls of the reference functions that write them, and compared with what `h5dump -H` finds in
the files exp_amd/csrc_host/h5cache.c writes.  No reference-written file exists anywhere (the reference ships none
and cannot be built here), so this is the reference-side pin of the layout that IS available: a name that EXP's reader
would look for and not find, an attribute of another type, or an object EXP never writes fails here.

Runs only where /root/reference exists (this container); nothing of it is copied -- the test reads the sources in place.
CPU only."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from tests.conftest import make_grid

REF = "/root/reference"
H5DUMP = shutil.which("h5dump") or "/opt/conda/bin/h5dump"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or not os.path.exists(H5DUMP),
                                reason="needs the reference sources and h5dump")

CTYPE = {"int": "H5T_STD_I32LE", "double": "H5T_IEEE_F64LE", "std::string": "H5T_STRING", "unsigned": "H5T_STD_U32LE",
         "float": "H5T_IEEE_F32LE"}


@pytest.fixture(scope="module")
def h5():
    from exp_amd import h5cache
    if not h5cache.available():
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.run(["make", "-s", "h5"], cwd=root, check=False)
    if not h5cache.available():
        pytest.skip("HDF5 C headers/library not available")
    return h5cache


def ref_calls(relpath, *functions):
    """{("attr" | "dset" | "group", name): C++ type or None} over the bodies of the named functions of one source file
    (a body runs from the line that defines `Class::function` to the brace that closes it)."""
    lines = open(os.path.join(REF, relpath), errors="replace").read().split("\n")
    out = {}
    for fn in functions:
        # (a definition: return type, the qualified name, then the parameter list on this line or the next)
        starts = [i for i, l in enumerate(lines)
                  if re.match(r"^\s*[\w:<>&\*]+(\s+[\w:<>&\*]+)*\s+" + re.escape(fn) + r"\s*(\(.*)?$", l) and '"' not in l
                  and not l.rstrip().endswith(";")]
        assert starts, (relpath, fn)
        for s in starts:
            # the body: from the first opening brace to its match (braces inside string literals are not counted)
            depth, e, seen = 0, s, False
            while e < len(lines):
                code = re.sub(r'"(\\.|[^"\\])*"', '""', lines[e]).split("//")[0]
                depth += code.count("{") - code.count("}")
                seen = seen or "{" in code
                if seen and depth <= 0:
                    break
                e += 1
            body = "\n".join(lines[s:e + 1])
            for t, name in re.findall(r'createAttribute<\s*([^>]+?)\s*>\s*\(\s*"([^"]+)"', body):
                out[("attr", name)] = t
            for name in re.findall(r'createDataSet(?:<[^>]*>)?\s*\(\s*"([^"]+)"', body):
                out[("dset", name)] = None
            for name in re.findall(r'createGroup\s*\(\s*"([^"]+)"', body):
                out[("group", name)] = None
    return out


def dumped(path):
    """{("attr" | "dset" | "group", name): HDF5 type class} of a file; groups with all-digit names (harmonic orders,
    snapshot numbers -- the reference forms them with an ostringstream) are left out."""
    txt = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True, check=True).stdout.split("\n")
    out = {}
    for i, l in enumerate(txt):
        m = re.match(r'\s*(ATTRIBUTE|DATASET|GROUP) "([^"]+)"', l)
        if not m or m.group(2) == "/":
            continue
        kind = {"ATTRIBUTE": "attr", "DATASET": "dset", "GROUP": "group"}[m.group(1)]
        if kind == "group" and m.group(2).isdigit():
            continue
        typ = None
        if kind != "group":
            t = re.match(r"\s*DATATYPE\s+(\S+)", txt[i + 1])
            assert t, (path, l, txt[i + 1])
            typ = t.group(1)
        prev = out.setdefault((kind, m.group(2)), typ)
        assert prev == typ, (m.group(2), prev, typ)
    return out


def compare(ref, mine, optional=(), dset_types=None):
    for (kind, name), ctype in ref.items():
        if (kind, name) in optional and (kind, name) not in mine:
            continue
        assert (kind, name) in mine, f"the reference writes {kind} '{name}', the file does not have it"
        if kind == "attr":
            assert mine[(kind, name)] == CTYPE[ctype], (name, ctype, mine[(kind, name)])
    for key in mine:
        assert key in ref, f"{key[0]} '{key[1]}' is in the file, the reference's writer has no such object"
    for name, typ in (dset_types or {}).items():
        assert mine[("dset", name)] == typ, (name, mine[("dset", name)], typ)


def test_slgrid_cache_names_and_types(h5, tmp_path):
    ref = ref_calls("exputil/SLGridMP2.cc", "SLGridSph::WriteH5Cache")
    assert ("attr", "rmapping") in ref and ("dset", "ef") in ref and len(ref) >= 15      # (the extraction found the function)
    model, g = make_grid("plummer", 3, 6, 200)
    path = str(tmp_path / "SLGridSph.cache")
    h5.write_slgrid_cache(path, g, "SLGridSph.model")
    compare(ref, dumped(path), dset_types={"ev": "H5T_IEEE_F64LE", "ef": "H5T_IEEE_F64LE"})


def test_empcyl_cache_names_and_types(h5, tmp_path):
    from exp_amd.empcyl import build_empcyl
    ref = ref_calls("exputil/EmpCylSL.cc", "EmpCylSL::WriteH5Cache")
    assert ("attr", "lmaxfid") in ref and ("dset", "zforceS") in ref and ("group", "Sine") in ref
    g = build_empcyl(mmax=1, norder=2, numx=8, numy=4, lmaxfid=6, nmaxfid=4, numr=200, rnum=24, tnum=12)
    path = str(tmp_path / ".eof.cache")
    h5.write_empcyl_cache(path, g, lmaxfid=6, nmaxfid=4, cmass=0.5)
    compare(ref, dumped(path), dset_types={k: "H5T_IEEE_F64LE" for k in ("potC", "rforceC", "zforceC", "densC", "potS")})


def test_coefficient_file_names_and_types(h5, tmp_path):
    from exp_amd.basis import CylStruct, SphStruct
    from exp_amd.coefs import CylCoefs, SphCoefs
    rng = np.random.default_rng(1)
    common = ("Coefs::WriteH5Coefs", "Coefs::WriteH5Units")
    for geom in ("sphere", "cylinder"):
        cls = "SphCoefs" if geom == "sphere" else "CylCoefs"
        ref = ref_calls("expui/Coefficients.cc", *common, cls + "::WriteH5Params", cls + "::WriteH5Times")
        assert ("attr", "CoefficientOutputVersion") in ref and ("dset", "coefficients") in ref and ("attr", "Rotation") in ref
        assert (("attr", "scale") in ref) == (geom == "sphere") and (("attr", "mmax") in ref) == (geom == "cylinder")
        if geom == "sphere":
            cs = SphCoefs("halo")
            cs.add(SphStruct(2, 4, 1.0, 0.0, rng.standard_normal((6, 4)) + 0j, np.zeros(3), np.eye(3)))
        else:
            cs = CylCoefs("disk")
            cs.add(CylStruct(2, 4, 0.0, rng.standard_normal((3, 4)) + 0j, np.zeros(3), np.eye(3)))
        path = str(tmp_path / f"outcoef.{geom}.h5")
        cs.setUnits([("length", "kpc", 1.0), ("mass", "Msun", 1.0e10), ("time", "Gyr", 1.0), ("G", "mixed", 43007.1)])
        cs.WriteH5Coefs(path, config="id: x")
        # (Coefs::WriteH5Units: a compound {char[16] name, char[16] unit, float value} per unit, expui/Coefficients.cc:20-27)
        compare(ref, dumped(path), dset_types={"count": "H5T_STD_U32LE", "coefficients": "H5T_COMPOUND", "Units": "H5T_COMPOUND"})


def test_covariance_store_names_and_types(h5, tmp_path):
    ref = ref_calls("expui/Covariance.cc", "SubsampleCovariance::writeCovarH5", "SubsampleCovariance::writeCoefCovariance")
    ref.update(ref_calls("expui/BiorthBasis.cc", "Spherical::writeCovarH5Params"))
    assert ("attr", "FloatSize") in ref and ("dset", "covariance_real_total") in ref and ("attr", "rmin") in ref
    rng = np.random.default_rng(2)
    T, ltot, nmax = 3, 3, 4
    counts = rng.integers(5, 50, T).astype(np.int32)
    masses = rng.uniform(0.1, 1.0, T)
    mean = rng.standard_normal((T, ltot, nmax)) + 1j * rng.standard_normal((T, ltot, nmax))
    a = rng.standard_normal((T, ltot, nmax, nmax)) + 1j * rng.standard_normal((T, ltot, nmax, nmax))
    covr = a + np.conj(np.swapaxes(a, 2, 3))
    seen = {}
    for summed in (True, False):
        path = str(tmp_path / f"covar{int(summed)}.h5")
        assert h5.covar_append(path, "SphereSL", 0, (1, nmax), (1.0, 1e-4, 1.95), 0.25, counts, masses, mean, covr,
                               summed=summed, covar=True)
        seen.update(dumped(path))
    # the summed and the per-sample forms together use every dataset name of the writer
    compare(ref, seen, dset_types={"count": "H5T_STD_U32LE", "sampleCounts": "H5T_STD_I32LE", "sampleMasses": "H5T_IEEE_F64LE",
                                   "coefficients_real": "H5T_IEEE_F64LE", "covariance_real": "H5T_IEEE_F64LE"})
