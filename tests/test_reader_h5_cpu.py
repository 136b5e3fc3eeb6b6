"""The HDF5 phase-space formats (exp_amd/reader_h5.py over exp_amd/csrc_host/h5part.c): EXP's OutHDF5 snapshots in both
layouts, written here and read back through ``PSPhdf5``; Gadget HDF5 snapshots through ``GadgetHDF5``; and the object
names / types of the written files against the reference's WRITER SOURCE (src/OutHDF5.cc, src/Component.cc) the way
tests/test_ref_h5layout.py does it for the basis caches.  CPU only."""
import ctypes
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from exp_amd import reader as R

REF = "/root/reference"
H5DUMP = shutil.which("h5dump") or "/opt/conda/bin/h5dump"


@pytest.fixture(scope="module")
def H():
    from exp_amd import reader_h5
    if not os.path.exists(os.path.join(os.path.dirname(reader_h5.__file__), "libexp_amd_h5.so")):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.run(["make", "-s", "h5"], cwd=root, check=False)
    try:
        reader_h5._h5()
    except (RuntimeError, OSError):
        pytest.skip("HDF5 C headers/library not available")
    if not hasattr(reader_h5._h5(), "exp_h5p_particles_write"):
        pytest.skip("exp_amd/libexp_amd_h5.so predates h5part.c")
    return reader_h5


def _comp(rng, name, n, ni, nd, equal=False):
    c = dict(name=name, force="sphereSL", fconf="{Lmax: 2, nmax: 10}",
             mass=np.full(n, 0.5 / max(n, 1)) if equal else rng.uniform(1, 2, n) / max(n, 1),
             pos=rng.normal(size=(n, 3)), vel=rng.normal(size=(n, 3)), pot=rng.normal(size=n),
             potext=rng.normal(size=n), indx=(rng.permutation(n) + 1).astype(np.uint64))
    if ni:
        c["iattrib"] = rng.integers(-5, 5, (n, ni)).astype(np.int32)
    if nd:
        c["dattrib"] = rng.normal(size=(n, nd))
    return c


@pytest.mark.parametrize("gadget4", [False, True])
@pytest.mark.parametrize("real4", [False, True])
def test_psphdf5_round_trip(H, tmp_path, gadget4, real4):
    """write_psp_hdf5 -> PSPhdf5 for both layouts and both precisions, with chunking + deflate + shuffle: every field
    back bit for bit (floats: the narrowing of the doubles), a component of equal masses through MassTable, an empty one."""
    rng = np.random.default_rng(3)
    comps = [_comp(rng, "dark halo", 300, 2, 1), _comp(rng, "star", 40, 0, 0, equal=True), _comp(rng, "gas", 0, 0, 0)]
    path = str(tmp_path / "snap_00001.1")
    H.write_psp_hdf5(path, 0.4, comps, real4=real4, gadget4=gadget4, chunk=64, compress=3, version=("abc", "main", "today"))
    rd = R.ParticleReader.createReader("PSPhdf5", [path])
    assert rd.GetTypes() == ["dark halo", "star", "gas"] and rd.CurrentTime() == 0.4 and rd.NumFiles() == 1
    assert rd.gadget4 == gadget4 and rd.real4 == real4 and rd.CurrentNumber() == 300
    f = (lambda x: np.asarray(x).astype(np.float32).astype(np.float64)) if real4 else np.asarray
    for c in comps:
        rd.SelectType(c["name"])
        a = rd.arrays()
        assert rd.CurrentNumber() == len(c["mass"]) == len(a["mass"])
        for k in ("pos", "vel", "pot", "potext"):
            assert np.array_equal(a[k], f(c[k])), (c["name"], k)
        table = gadget4 and c["name"] == "star"             # the table holds the double
        assert np.array_equal(a["mass"], np.asarray(c["mass"]) if table else f(c["mass"]))
        assert np.array_equal(a["indx"], c["indx"])
        if "iattrib" in c:
            assert np.array_equal(a["iattrib"], c["iattrib"]) and np.array_equal(a["dattrib"], f(c["dattrib"]))
    rd.SelectType("star")
    p = rd.firstParticle()
    assert p.indx == int(comps[1]["indx"][0]) and p.potext == f(comps[1]["potext"])[0]
    with pytest.raises(RuntimeError):
        rd.SelectType("bulge")


def test_psphdf5_directory_of_parts_and_ranks(H, tmp_path):
    """A snapshot in three files inside a directory (names ending in a digit; anything else is ignored), read as one;
    NumFilesPerSnapshot must match; with two ranks particle n of each file goes to rank n mod 2."""
    rng = np.random.default_rng(5)
    whole = _comp(rng, "dark", 90, 1, 0)
    d = tmp_path / "snap_00002"
    d.mkdir()
    (d / "README").write_text("not a part")
    cuts = [0, 20, 55, 90]
    for k in range(3):
        s = slice(cuts[k], cuts[k + 1])
        part = {key: (v[s] if isinstance(v, np.ndarray) else v) for key, v in whole.items()}
        H.write_psp_hdf5(str(d / f"snap_00002.{k + 1}"), 1.5, [part], nfiles=3, totals=[90])
    rd = R.ParticleReader.createReader("PSPhdf5", [str(d)])
    assert rd.NumFiles() == 3 and rd.CurrentNumber() == 90 and os.path.basename(rd._files[0]) == "snap_00002.1"
    rd._files[1:] = sorted(rd._files[1:])
    a = rd.arrays()
    assert np.array_equal(a["pos"], whole["pos"]) and np.array_equal(a["indx"], whole["indx"])
    got = []
    for myid in range(2):
        rd.numprocs, rd.myid, rd._sel = 2, myid, None
        got.append(rd.arrays()["indx"])
    want = [np.concatenate([whole["indx"][cuts[k]:cuts[k + 1]][myid::2] for k in range(3)]) for myid in range(2)]
    assert all(np.array_equal(g, w) for g, w in zip(got, want))
    os.remove(d / "snap_00002.3")
    with pytest.raises(RuntimeError, match="number of files"):
        R.ParticleReader.createReader("PSPhdf5", [str(d)])


def _write_gadget_h5(H, path, npart, table, time, rng, masses_for=(), ids=True, double=False):
    lib = H._h5()
    p = path.encode()
    assert lib.exp_h5p_create(p) == 0 and lib.exp_h5p_group(p, b"/Header") == 0
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)

    def attr(name, kind, val, scalar=False):
        a = np.ascontiguousarray(val, dtype=H._KIND[kind]).reshape(-1)
        assert lib.exp_h5p_attr_write(p, b"/Header", name.encode(), ctypes.c_char(kind.encode()), -1 if scalar else len(a), vp(a)) == 0
    attr("Time", "d", [time], scalar=True)
    attr("MassTable", "d", table)
    attr("NumPart_ThisFile", "i", npart)
    out = {}
    real = "d" if double else "f"
    for k, n in enumerate(npart):
        if n == 0:
            continue
        g = f"/PartType{k}"
        assert lib.exp_h5p_group(p, g.encode()) == 0
        data = {"Coordinates": rng.normal(size=(n, 3)), "Velocities": rng.normal(size=(n, 3))}
        if k in masses_for:
            data["Masses"] = rng.uniform(1, 2, n)
        for name, arr in data.items():
            a = np.ascontiguousarray(arr, dtype=H._KIND[real])
            dims = (ctypes.c_longlong * 4)(*(list(a.shape) + [1] * (4 - a.ndim)))
            assert lib.exp_h5p_dset_write(p, (g + "/" + name).encode(), ctypes.c_char(real.encode()), a.ndim, dims, vp(a), 0, 0, 0) == 0
            data[name] = a
        idv = (np.arange(n) * 7 + 1000 * k).astype(np.uint32) if ids else np.zeros(0, np.uint32)
        dims = (ctypes.c_longlong * 4)(len(idv), 1, 1, 1)
        assert lib.exp_h5p_dset_write(p, (g + "/ParticleIDs").encode(), ctypes.c_char(b"u"), 1, dims, vp(idv), 0, 0, 0) == 0
        data["ParticleIDs"] = idv
        out[k] = data
    return out


def test_gadget_hdf5(H, tmp_path):
    """exputil/ParticleReader.cc:333-690: "Halo" by default, masses from the table unless a Masses dataset with storage
    exists, everything converted to float on reading (a double file loses its low bits, as there), ids as unsigned 32-bit;
    an id dataset without storage numbers the particles from 1."""
    rng = np.random.default_rng(7)
    a_path, b_path = str(tmp_path / "snap_010.0.hdf5.0"), str(tmp_path / "snap_010.1.hdf5.1")
    fa = _write_gadget_h5(H, a_path, [4, 25, 10, 0, 0, 0], [0.0, 0.125, 0.0, 0, 0, 0], 2.5, rng, masses_for=(0, 2))
    fb = _write_gadget_h5(H, b_path, [0, 15, 5, 0, 0, 0], [0.0, 0.125, 0.0, 0, 0, 0], 2.5, rng, masses_for=(2,), double=True)
    rd = R.ParticleReader.createReader("GadgetHDF5", [a_path, b_path])
    assert rd.GetTypes() == ["Disk", "Gas", "Halo"] and rd.CurrentTime() == 2.5 and rd.CurrentNumber() == 25
    a = rd.arrays()
    assert len(a["mass"]) == 40 and np.all(a["mass"] == 0.125)
    want = np.concatenate([fa[1]["Coordinates"].astype(np.float64), fb[1]["Coordinates"].astype(np.float32).astype(np.float64)])
    assert np.array_equal(a["pos"], want)
    assert np.array_equal(a["indx"], np.concatenate([fa[1]["ParticleIDs"], fb[1]["ParticleIDs"]]).astype(np.uint64))
    rd.SelectType("Disk")
    a = rd.arrays()
    assert np.array_equal(a["mass"], np.concatenate([fa[2]["Masses"].astype(np.float64),
                                                     fb[2]["Masses"].astype(np.float32).astype(np.float64)]))
    c_path = str(tmp_path / "noid.0")
    fc = _write_gadget_h5(H, c_path, [0, 6, 0, 0, 0, 0], [0, 0.5, 0, 0, 0, 0], 0.0, rng, ids=False)
    r2 = R.ParticleReader.createReader("GadgetHDF5", [c_path])
    assert np.array_equal(r2.arrays()["indx"], np.arange(1, 7)) and np.array_equal(r2.arrays()["vel"], fc[1]["Velocities"].astype(np.float64))
    with pytest.raises(RuntimeError):
        rd.SelectType("Dark")


# ---- the names and types against the reference's writer source ---------------------------------------------
CTYPE = {"int": "H5T_STD_I32LE", "double": "H5T_IEEE_F64LE", "string": "H5T_STRING", "ulong": "H5T_STD_U64LE",
         "long": "H5T_STD_I64LE", "float": "H5T_IEEE_F32LE"}


def _dumped(path):
    txt = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True, check=True).stdout.split("\n")
    out, stack = {}, []
    for i, l in enumerate(txt):
        m = re.match(r'\s*(ATTRIBUTE|DATASET|GROUP) "([^"]+)"', l)
        if m:
            kind, name = m.groups()
            if kind == "GROUP":
                if name != "/":
                    out[("group", name)] = None
                continue
            t = next((re.search(r"DATATYPE\s+(\S+)", x).group(1) for x in txt[i + 1:i + 4] if "DATATYPE" in x), None)
            out[("attr" if kind == "ATTRIBUTE" else "dset", name)] = t
    return out


@pytest.mark.skipif(not os.path.isdir(REF) or not os.path.exists(H5DUMP), reason="needs the reference sources and h5dump")
def test_layout_against_the_reference_writers_source(H, tmp_path):
    """Every attribute OutHDF5::RunGadget4 / RunPSP create and every dataset Component::write_HDF5 / write_H5 create, by
    name, must be in the file written here with the type the C++ variable has; and nothing else may be in it."""
    src = open(os.path.join(REF, "src/OutHDF5.cc")).read()
    g4 = src[src.index("void OutHDF5::RunGadget4"):src.index("// Helper for writing scalar")] if "// Helper for writing scalar" in src \
        else src[src.index("void OutHDF5::RunGadget4"):src.index("void OutHDF5::RunPSP")]
    psp = src[src.index("void OutHDF5::RunPSP"):src.index("void OutHDF5::checkParticleMasses")]
    a_g4 = set(re.findall(r'createAttribute(?:<[^>]+>)?\("(\w+)"', g4))
    a_psp = set(re.findall(r'write(?:Scalar|Vector)\(\w+,\s*"(\w+)"', psp))
    assert a_g4 == a_psp and {"MassTable", "NumPart_ThisFile", "Time", "NumFilesPerSnapshot", "NumPart_Total", "PSPstyle",
                              "NTYPES", "DOUBLEPRECISION", "Niattrib", "Ndattrib", "ComponentNames"} <= a_g4
    assert set(re.findall(r'createGroup\("(\w+)"\)', g4)) == {"Header", "Config", "Parameters"}
    comp = open(os.path.join(REF, "src/Component.cc")).read()
    wh = comp[comp.index("void Component::write_HDF5(HighFive::Group& group"):comp.index("void Component::write_H5(H5::Group& group)")]
    d_g4 = set(re.findall(r'createDataSet\("(\w+)"', wh))
    assert d_g4 == {"Masses", "ParticleIDs", "Coordinates", "Velocities", "Potential", "PotentialExt", "IntAttributes",
                    "RealAttributes"}
    w5 = comp[comp.index("void Component::write_H5(H5::Group& group)"):comp.index("void Component::write_binary_header")]
    members = re.findall(r'insertMember\("(\w+)"', w5)
    assert members == ["id", "mass", "pos", "vel", "pot", "potext", "iattrib", "dattrib"]
    assert re.search(r'createDataSet\("particles"', w5)
    # C++ types of the variables written
    types = {"MassTable": "double", "NumPart_ThisFile": "ulong", "NumPart_Total": "ulong", "Time": "double",
             "Flag_DoublePrecision": "int", "HubbleParam": "double", "Omega0": "double", "OmegaBaryon": "double",
             "OmegaLambda": "double", "Redshift": "double", "NumFilesPerSnapshot": "int", "PSPstyle": "int", "NTYPES": "int",
             "DOUBLEPRECISION": "int", "Niattrib": "int", "Ndattrib": "int", "Git_commit": "string", "Git_branch": "string",
             "Compile_date": "string", "ComponentNames": "string", "ForceMethods": "string", "ForceConfigurations": "string",
             "EXPConfiguration": "string"}
    assert set(types) == a_g4
    assert "std::vector<unsigned long> nums(masses.size());" in g4 and "std::vector<long int> ids;" in wh
    rng = np.random.default_rng(1)
    comps = [_comp(rng, "dark", 50, 1, 2)]
    for gadget4 in (True, False):
        path = str(tmp_path / f"layout_{int(gadget4)}.1")
        H.write_psp_hdf5(path, 0.0, comps, gadget4=gadget4, expconfig="Global: {}")
        got = _dumped(path)
        for name, t in types.items():
            assert got.get(("attr", name)) == CTYPE[t], (name, got.get(("attr", name)))
        assert {k[1] for k in got if k[0] == "attr"} == a_g4
        assert {k[1] for k in got if k[0] == "group"} == {"Header", "Config", "Parameters", "PartType0"}
        dsets = {k[1]: v for k, v in got.items() if k[0] == "dset"}
        if gadget4:
            assert set(dsets) == d_g4
            assert dsets["ParticleIDs"] == CTYPE["long"] and dsets["Coordinates"] == CTYPE["double"]
            assert dsets["IntAttributes"] == CTYPE["int"]
        else:
            assert set(dsets) == {"particles"} and dsets["particles"] == "H5T_COMPOUND"
            txt = subprocess.run([H5DUMP, "-H", "-d", "/PartType0/particles", path], capture_output=True, text=True).stdout
            assert re.findall(r'"(\w+)";', txt) == members
            assert re.search(r'H5T_STD_I32LE "id";', txt) and re.search(r'H5T_VLEN \{ H5T_STD_I32LE\} "iattrib";', txt)
            assert re.search(r'H5T_ARRAY \{ \[3\] H5T_IEEE_F64LE \} "pos";', txt)
    # and what the reference's readers ask for is what was written
    rsrc = open(os.path.join(REF, "exputil/ParticleReader.cc")).read()
    rd = rsrc[rsrc.index("void PSPhdf5::getInfo()"):rsrc.index("template <typename T>\n  struct H5Particle")]
    assert set(re.findall(r'getAttribute\("(\w+)"\)', rd)) <= a_g4
    assert set(re.findall(r'getDataSet\("(\w+)"\s*\)', rd)) == d_g4
