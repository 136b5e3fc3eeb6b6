"""The particle readers and writers of exp_amd/reader.py (pyEXP.read; include/ParticleReader.H, exputil/ParticleReader.cc)
against the oracle's statement-by-statement restatement of the reference's writer and reader (oracle/psp_oracle.c), and
-- for the two headers of the reference that compile on their own -- against the reference itself
(oracle/_ref/libref_headers.so: include/P2Quantile.H, include/gadget.H).  CPU only."""
import ctypes
import io
import os
import struct

import numpy as np
import pytest

from exp_amd import reader as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _comp(rng, name, n, indexing, ni, nd, force="sphereSL"):
    c = dict(info=R.component_info(name, force, {"Lmax": 2, "nmax": 10, "modelname": "SLGridSph.model"},
                                   {"nlevel": 1, "indexing": indexing}, extra={"bodyfile": f"{name}.bods"}),
             indexing=indexing, mass=rng.uniform(1, 2, n) / n, pos=rng.normal(size=(n, 3)), vel=rng.normal(size=(n, 3)),
             pot=rng.normal(size=n), potext=rng.normal(size=n) * 0.1)
    if ni:
        c["iattrib"] = rng.integers(-2 ** 31, 2 ** 31 - 1, (n, ni)).astype(np.int32)
    if nd:
        c["dattrib"] = rng.normal(size=(n, nd))
    if indexing:
        c["indx"] = (rng.permutation(n) + 1).astype(np.uint64) * 3
    return c


@pytest.mark.parametrize("real4", [False, True])
def test_psp_writer_is_byte_identical_to_the_oracles(oracle, tmp_path, real4):
    """write_psp against Component::write_binary / Particle::writeBinary restated one stream write per field; three
    components: indexed with both attribute kinds, plain, and one whose YAML stanza exceeds the default 1024-byte info
    field (the header grows, src/Component.cc:2399-2408); an empty component too."""
    rng = np.random.default_rng(5)
    comps = [_comp(rng, "dark", 501, True, 2, 1), _comp(rng, "star", 64, False, 0, 3), _comp(rng, "gas", 0, False, 0, 0),
             _comp(rng, "bulge", 7, True, 0, 0)]
    comps[3]["info"] = R.component_info("bulge", "sphereSL", {"note": "x" * 1500}, {"nlevel": 1, "indexing": True})
    a, b = tmp_path / "OUT.a", tmp_path / "OUT.b"
    R.write_psp(str(a), 1.25, comps, real4)
    oracle.psp_write(b, 1.25, comps, real4)
    assert a.read_bytes() == b.read_bytes()
    assert struct.unpack("<dii", a.read_bytes()[:16]) == (1.25, 501 + 64 + 7, 4)


@pytest.mark.parametrize("real4", [False, True])
def test_pspout_reader_against_the_oracles_reader(oracle, tmp_path, real4):
    """PSPout on a file written by the ORACLE's writer against the oracle's PSPout restatement, for one rank and dealt
    over three (stagger by myid, stride numprocs: exputil/ParticleReader.cc:1689-1735); a file without indices numbers
    its particles from 0 (``indx = pcount``, include/ParticleReader.H:283)."""
    rng = np.random.default_rng(6)
    comps = [_comp(rng, "dark", 1000, True, 1, 2), _comp(rng, "star", 333, False, 2, 0)]
    path = tmp_path / "OUT.run0.00003"
    oracle.psp_write(path, 0.5, comps, real4)
    rd = R.ParticleReader.createReader("PSPout", [str(path)], 0, False)
    assert rd.GetTypes() == ["dark", "star"] and rd.CurrentTime() == 0.5 and rd.CurrentNumber() == 1000
    t, ntot, ref = oracle.psp_read(path, [True, False])
    assert (t, ntot) == (0.5, 1333) and [s["nbod"] for s in ref] == [1000, 333]
    assert ref[0]["r_size"] == (4 if real4 else 8)
    for name, s, c in zip(("dark", "star"), ref, comps):
        rd.SelectType(name)
        a = rd.arrays()
        for k in ("indx", "mass", "pos", "vel", "pot"):
            assert np.array_equal(a[k], s[k]), (name, k)
        if s["niatr"]:
            assert np.array_equal(a["iattrib"], s["iattrib"]) and np.array_equal(a["iattrib"], c["iattrib"])
        if s["ndatr"]:
            assert np.array_equal(a["dattrib"], s["dattrib"])
        # the values: the doubles themselves, or their float narrowing; pot is pot + potext (exputil/Particle.cc:369)
        want = c["pot"] + c["potext"]
        if real4:
            assert np.array_equal(a["mass"], c["mass"].astype(np.float32).astype(np.float64))
            assert np.array_equal(a["pot"], want.astype(np.float32).astype(np.float64))
        else:
            assert np.array_equal(a["pos"], c["pos"]) and np.array_equal(a["pot"], want)
        assert np.array_equal(a["indx"], c["indx"] if c["indexing"] else np.arange(len(c["mass"])))
    # the one-at-a-time interface
    rd.SelectType("star")
    p = rd.firstParticle()
    assert p.indx == 0 and p.mass == ref[1]["mass"][0] and list(p.iattrib) == list(ref[1]["iattrib"][0])
    count = 1
    while rd.nextParticle() is not None:
        count += 1
    assert count == 333
    # three ranks
    for myid in range(3):
        _, _, part = oracle.psp_read(path, [True, False], 3, myid)
        rd.numprocs, rd.myid = 3, myid
        for name, s in zip(("dark", "star"), part):
            rd.SelectType(name)
            a = rd.arrays()
            assert len(a["mass"]) == len(s["mass"]) == len(range(myid, s["nbod"], 3))
            for k in ("indx", "mass", "pos", "vel", "pot"):
                assert np.array_equal(a[k], s[k]), (myid, name, k)
    with pytest.raises(RuntimeError):
        rd.SelectType("gas")


def test_psp_old_style_info_and_missing_magic(tmp_path):
    """A pre-YAML info string ``name : id : cparam : fparam`` with ``indexing=1`` in cparam (exputil/ParticleReader.cc:
    1405-1437), and a component whose first word is not the magic: the reader assumes doubles and does NOT step back
    (:1327-1333)."""
    rng = np.random.default_rng(2)
    n = 5
    info = b"halo : sphereSL : nlevel=1, indexing=1, foo=2 : Lmax=2, nmax=10"
    rec = np.zeros(n, dtype=R.psp_record_dtype(8, True, 0, 0))
    rec["indx"], rec["mass"], rec["pos"] = np.arange(10, 10 + n), rng.uniform(size=n), rng.normal(size=(n, 3))
    path = tmp_path / "OUT.old"
    with open(path, "wb") as f:
        f.write(struct.pack("<dii", 2.0, n, 1))
        f.write(struct.pack("<Q", 12345))                       # no magic: 8-byte reals
        f.write(struct.pack("<4i", n, 0, 0, 128) + info.ljust(128, b"\0"))
        f.write(rec.tobytes())
    rd = R.PSPout([str(path)])
    st = rd.stanzas[0]
    assert (st.name, st.id, st.cparam, st.fparam) == ("halo", "sphereSL", "nlevel=1, indexing=1, foo=2", "Lmax=2, nmax=10")
    assert st.index_size == 8 and st.r_size == 8
    a = rd.arrays()
    assert np.array_equal(a["indx"], rec["indx"]) and np.array_equal(a["pos"], rec["pos"])
    out = io.StringIO()
    rd.PrintSummary(stats=False, out=out)
    assert "Time=2\n" in out.getvalue() and "name :: halo" in out.getvalue() and "rsize :: 8" in out.getvalue()


def test_split_psp_round_trip(tmp_path, oracle):
    """write_spl -> PSPspl gives what PSPout gives for the monolithic file of the same components; the master's layout
    (magic, int number of parts, header, 1024-byte names: src/Component.cc:2737-2759) checked byte by byte."""
    rng = np.random.default_rng(9)
    comps = [_comp(rng, "dark", 100, True, 1, 1), _comp(rng, "star", 11, False, 0, 0)]
    mono, master = tmp_path / "OUT.x", tmp_path / "SPL.x.00001"
    R.write_psp(str(mono), 3.0, comps, True)
    parts = R.write_spl(str(master), 3.0, comps, nparts=4, real4=True)
    assert [os.path.basename(p) for p in parts[:5]] == ["SPL.x.00001_0-0", "SPL.x.00001_0-1", "SPL.x.00001_0-2",
                                                        "SPL.x.00001_0-3", "SPL.x.00001_1-0"]
    raw = master.read_bytes()
    assert struct.unpack("<dii", raw[:16]) == (3.0, 111, 2)
    magic, number = struct.unpack("<Qi", raw[16:28])
    assert magic == 0xadbfabc0 + 4 and number == 4
    nbod, niatr, ndatr, ninfo = struct.unpack("<4i", raw[28:44])
    assert (nbod, niatr, ndatr, ninfo) == (100, 1, 1, 1024)
    names = raw[44 + ninfo: 44 + ninfo + 4 * 1024]
    assert names[:1024].rstrip(b"\0") == b"SPL.x.00001_0-0" and names[3072:].rstrip(b"\0") == b"SPL.x.00001_0-3"
    assert struct.unpack("<I", open(parts[0], "rb").read(4)) == (25,)
    a, b = R.ParticleReader.createReader("PSPout", [str(mono)]), R.ParticleReader.createReader("PSPspl", [str(master)])
    assert a.GetTypes() == b.GetTypes() and b.CurrentTime() == 3.0
    for name in ("dark", "star"):
        a.SelectType(name); b.SelectType(name)
        for numprocs, myid in ((1, 0), (3, 1)):
            a.numprocs = b.numprocs = numprocs
            a.myid = b.myid = myid
            a._sel = b._sel = None
            for k, v in a.arrays().items():
                assert np.array_equal(v, b.arrays()[k]), (name, k)


def test_empty_component_with_attributes(oracle, tmp_path):
    """nbod = 0 with niatr / ndatr > 0: the header still states the attribute counts (found by tests/fuzz/fuzz_reader.py)."""
    rng = np.random.default_rng(4)
    comps = [_comp(rng, "dark", 0, True, 2, 3), _comp(rng, "star", 5, False, 1, 0)]
    a, b, m = tmp_path / "OUT.a", tmp_path / "OUT.b", tmp_path / "SPL.e"
    R.write_psp(str(a), 0.0, comps)
    oracle.psp_write(b, 0.0, comps)
    assert a.read_bytes() == b.read_bytes()
    R.write_spl(str(m), 0.0, comps, nparts=3)
    rd = R.PSPspl([str(m)])
    assert (rd.stanzas[0].nbod, rd.stanzas[0].niatr, rd.stanzas[0].ndatr) == (0, 2, 3) and len(rd.arrays()["mass"]) == 0
    rd.SelectType("star")
    assert np.array_equal(rd.arrays()["iattrib"], comps[1]["iattrib"])


def test_psp_copy_as_the_reference_writes_it(tmp_path):
    """PSP::writePSP (exputil/ParticleReader.cc:1883-1930) writes float records whatever ``real4`` says: the real4 copy
    of a double file reads back as the float narrowing of the original."""
    rng = np.random.default_rng(3)
    comps = [_comp(rng, "dark", 50, True, 1, 1)]
    src, dst = tmp_path / "OUT.d", tmp_path / "OUT.f"
    R.write_psp(str(src), 0.1, comps, False)
    with open(dst, "wb") as f:
        R.PSPout([str(src)]).writePSP(f, True)
    a = R.PSPout([str(dst)]).arrays()
    assert np.array_equal(a["pos"], comps[0]["pos"].astype(np.float32).astype(np.float64))
    assert np.array_equal(a["indx"], comps[0]["indx"]) and np.array_equal(a["iattrib"], comps[0]["iattrib"])


def _write_gadget(path, npart, masstab, time, rng, ids_from=1):
    h = np.zeros(1, dtype=R.GADGET_HEADER)
    h["npart"][0], h["mass"][0], h["time"], h["num_files"] = npart, masstab, time, 1
    h["npartTotal"][0] = npart
    tot = sum(npart)
    pos, vel = rng.normal(size=(tot, 3)).astype(np.float32), rng.normal(size=(tot, 3)).astype(np.float32)
    ids = np.arange(ids_from, ids_from + tot, dtype=np.int32)
    nm = sum(n for n, m in zip(npart, masstab) if m == 0)
    mass = rng.uniform(1, 2, nm).astype(np.float32)
    with open(path, "wb") as f:
        for blob in (h.tobytes(), pos.tobytes(), vel.tobytes(), ids.tobytes()) + ((mass.tobytes(),) if nm else ()):
            f.write(struct.pack("<i", len(blob)) + blob + struct.pack("<i", len(blob)))
    return pos, vel, ids, mass


def test_gadget_native(tmp_path):
    """A Gadget-2 snapshot in two files (exputil/ParticleReader.cc:37-318): "Halo" is selected at construction, the types
    found are those with particles, masses come from the table where it is non-zero and from the mass block -- which holds
    only the types whose table entry is zero -- otherwise; the iteration runs through both files."""
    rng = np.random.default_rng(4)
    d = tmp_path / "snap"
    d.mkdir()
    npart, tab = [5, 40, 30, 0, 7, 0], [0.0, 0.25, 0.0, 0.0, 0.0, 0.0]
    f0 = _write_gadget(d / "snap.0", npart, tab, 0.7, rng)
    f1 = _write_gadget(d / "snap.1", [3, 10, 20, 0, 0, 0], tab, 0.7, rng, ids_from=1000)
    (d / "snap.info").write_text("not a snapshot")              # no trailing digit: ignored by scanDirectory
    rd = R.ParticleReader.createReader("GadgetNative", [str(d)])
    assert sorted(os.path.basename(f) for f in rd._files) == ["snap.0", "snap.1"]
    rd._files.sort()
    assert rd.GetTypes() == ["Disk", "Gas", "Halo", "Stars"] and rd.CurrentTime() == 0.7
    assert rd.CurrentNumber() == 40                              # the file being read, not the snapshot
    a = rd.arrays()
    assert len(a["mass"]) == 50 and np.all(a["mass"] == 0.25)
    assert np.array_equal(a["pos"], np.concatenate([f0[0][5:45], f1[0][3:13]]).astype(np.float64))
    assert np.array_equal(a["indx"], np.concatenate([f0[2][5:45], f1[2][3:13]]).astype(np.uint64))
    rd.SelectType("Disk")
    a = rd.arrays()
    assert np.array_equal(a["vel"], np.concatenate([f0[1][45:75], f1[1][13:33]]).astype(np.float64))
    # the mass block holds Gas (5), Disk (30), Stars (7) of file 0: Disk starts after the 5 gas masses
    assert np.array_equal(a["mass"], np.concatenate([f0[3][5:35], f1[3][3:23]]).astype(np.float64))
    rd.SelectType("Stars")
    assert np.array_equal(rd.arrays()["mass"][:7], f0[3][35:42].astype(np.float64))
    with pytest.raises(RuntimeError):
        rd.SelectType("Dark")


def test_gadget_header_is_the_references(tmp_path):
    lib = os.path.join(ROOT, "oracle", "_ref", "libref_headers.so")
    if not os.path.exists(lib):
        pytest.skip("no oracle/_ref/libref_headers.so (built where /root/reference exists)")
    ref = ctypes.CDLL(lib)
    if not hasattr(ref, "ref_gadget_layout"):
        pytest.skip("oracle/_ref/libref_headers.so predates ref_gadget_layout")
    out = (ctypes.c_long * 12)()
    ref.ref_gadget_layout(out)
    dt = R.GADGET_HEADER
    mine = [dt.itemsize] + [dt.fields[k][1] for k in ("npart", "mass", "time", "redshift", "flag_sfr", "npartTotal",
                                                       "num_files", "BoxSize", "flag_metals", "npartTotalHighWord", "fill")]
    assert list(out) == mine


def test_p2quantile_is_the_references():
    """exp_amd.reader.P2Quantile against include/P2Quantile.H compiled in place, bit for bit, for sequences shorter and
    longer than the five markers, sorted, constant and heavy-tailed input, three probabilities."""
    lib = os.path.join(ROOT, "oracle", "_ref", "libref_headers.so")
    if not os.path.exists(lib):
        pytest.skip("no oracle/_ref/libref_headers.so (built where /root/reference exists)")
    ref = ctypes.CDLL(lib)
    if not hasattr(ref, "ref_p2quantile"):
        pytest.skip("oracle/_ref/libref_headers.so predates ref_p2quantile")
    ref.ref_p2quantile.restype = ctypes.c_double
    rng = np.random.default_rng(8)
    for n in (1, 2, 4, 5, 6, 7, 50, 1000):
        for make in (lambda: rng.normal(size=n), lambda: np.sort(rng.normal(size=n)), lambda: np.full(n, 2.5),
                     lambda: rng.standard_cauchy(size=n), lambda: -np.sort(rng.uniform(size=n))):
            x = np.ascontiguousarray(make(), dtype=np.float64)
            for p in (0.5, 0.1, 0.9):
                q = R.P2Quantile(p)
                for v in x:
                    q.addValue(float(v))
                want = ref.ref_p2quantile(ctypes.c_long(n), x.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(p))
                assert q.getQuantile() == want, (n, p)


def _write_tipsy(path, time, ngas, ndark, nstar, rng):
    h = np.zeros(1, dtype=R.TIPSY_HEADER)
    h["time"], h["nbodies"], h["ndim"], h["nsph"], h["ndark"], h["nstar"] = time, ngas + ndark + nstar, 3, ngas, ndark, nstar
    recs = []
    for n, dt in ((ngas, R.TIPSY_GAS), (ndark, R.TIPSY_DARK), (nstar, R.TIPSY_STAR)):
        r = np.zeros(n, dtype=dt)
        for name in dt.names:
            r[name] = rng.normal(size=r[name].shape).astype(np.float32)
        recs.append(r)
    with open(path, "wb") as f:
        f.write(h.tobytes())
        for r in recs:
            f.write(r.tobytes())
    return recs


def test_tipsy_native_and_bonsai(tmp_path):
    """include/tipsy.H + exputil/ParticleReader.cc:2091-2296: no type until SelectType; native indices are position + 1;
    Bonsai ids are the (eps, phi) words as one 64-bit integer, Bonsai1 the phi word as a 32-bit one; ranks take contiguous
    blocks, the last one the remainder (ios_psize)."""
    rng = np.random.default_rng(10)
    path = tmp_path / "snap.tipsy"
    gas, dark, star = _write_tipsy(path, 1.5, 7, 23, 5, rng)
    rd = R.ParticleReader.createReader("TipsyNative", [str(path)])
    assert rd.GetTypes() == ["Dark", "Gas", "Star"] and rd.CurrentTime() == 1.5 and rd.CurrentNumber() == 0
    with pytest.raises(RuntimeError):
        rd.firstParticle()
    rd.SelectType("Dark")
    a = rd.arrays()
    assert rd.CurrentNumber() == 23 and np.array_equal(a["pos"], dark["pos"].astype(np.float64))
    assert np.array_equal(a["indx"], np.arange(1, 24))
    rd.SelectType("Star")
    assert np.array_equal(rd.arrays()["mass"], star["mass"].astype(np.float64))
    rd.SelectType("Gas")
    assert np.array_equal(rd.arrays()["vel"], gas["vel"].astype(np.float64))
    b = R.ParticleReader.createReader("Bonsai", [str(path)])
    b.SelectType("Dark")
    want = [struct.unpack("<Q", struct.pack("<ff", e, p))[0] for e, p in zip(dark["eps"], dark["phi"])]
    assert [int(v) for v in b.arrays()["indx"]] == want
    b1 = R.ParticleReader.createReader("Bonsai1", [str(path)])
    b1.SelectType("Star")
    want = [struct.unpack("<i", struct.pack("<f", p))[0] & 0xffffffffffffffff for p in star["phi"]]
    assert [int(v) for v in b1.arrays()["indx"]] == want
    # three ranks: 23 // 3 = 7 each, the last takes 9; indices continue across the blocks
    rd.SelectType("Dark")
    got = []
    for myid in range(3):
        rd.numprocs, rd.myid, rd._sel = 3, myid, None
        a = rd.arrays()
        assert len(a["mass"]) == (9 if myid == 2 else 7)
        got.append(a["indx"])
    assert np.array_equal(np.concatenate(got), np.arange(1, 24))
    with pytest.raises(RuntimeError):
        R.ParticleReader.createReader("TipsyXDR", [str(path)])
    with pytest.raises(RuntimeError):
        R.ParticleReader.createReader("Ramses", [str(path)])


def test_file_batches():
    """parseStringList (exputil/ParticleReader.cc:1979-2034): sorted, grouped by what precedes the last delimiter; a name
    without delimiter closes the batch it lands in (the reference appends it to the open one)."""
    P = R.ParticleReader.parseStringList
    assert P(["snap_002.1", "snap_001.0", "snap_001.1", "snap_002.0"], ".") == [["snap_001.0", "snap_001.1"],
                                                                                  ["snap_002.0", "snap_002.1"]]
    assert P(["b-1", "a.7", "a.3"], ".-") == [["a.3", "a.7"], ["b-1"]]
    assert P(["x", "y"], ".") == [["x"], ["y"]]
    assert P(["a.0", "a.1", "c"], ".") == [["a.0", "a.1", "c"]]
    assert R.ParticleReader.getReaders()[:2] == ["PSPout", "PSPspl"]


def test_body_file_round_trip(tmp_path):
    """Component::read_bodies_and_distribute_ascii / Particle::readAscii: header `n niattrib ndattrib`, sequence numbers
    from 1 without `aindex`, missing attributes read as zero."""
    rng = np.random.default_rng(1)
    n = 20
    m, pos, vel = rng.uniform(size=n), rng.normal(size=(n, 3)), rng.normal(size=(n, 3))
    ia, da = rng.integers(-9, 9, (n, 2)), rng.normal(size=(n, 1))
    path = tmp_path / "new.bods"
    R.write_bodies_ascii(str(path), m, pos, vel, ia, da)
    b = R.read_bodies_ascii(str(path))
    assert np.array_equal(b["mass"], m) and np.array_equal(b["pos"], pos) and np.array_equal(b["vel"], vel)
    assert np.array_equal(b["iattrib"], ia) and np.array_equal(b["dattrib"], da) and np.array_equal(b["indx"], np.arange(1, n + 1))
    R.write_bodies_ascii(str(path), m, pos, vel, indx=np.arange(n) * 5 + 2)
    assert np.array_equal(R.read_bodies_ascii(str(path), aindex=True)["indx"], np.arange(n) * 5 + 2)
    (tmp_path / "short.bods").write_text("2 2 1\n1.0 0 0 0 0 0 0 7\n2.0 1 1 1 0 0 0 8 9 0.5\n")
    s = R.read_bodies_ascii(str(tmp_path / "short.bods"))
    assert s["iattrib"].tolist() == [[7, 0], [8, 9]] and s["dattrib"].tolist() == [[0.0], [0.5]]


def test_histograms_cpu_parts_match_the_oracle(oracle, tmp_path):
    """The bin indices and normalisations of FieldGenerator's histograms (expui/FieldGenerator.cc:776-1009); the float
    accumulation itself is the C-ABI's (tests/test_reader_gpu.py), here a float64 accumulation is compared to 1e-5."""
    from exp_amd.field import FieldGenerator

    class Seq(FieldGenerator):
        @staticmethod
        def _binsum(bins, vals, nbins):
            out = np.zeros(nbins)
            ok = bins >= 0
            np.add.at(out, bins[ok], vals[ok])
            return out.astype(np.float32)
    rng = np.random.default_rng(12)
    comps = [_comp(rng, "dark", 4000, False, 0, 0)]
    path = tmp_path / "OUT.h"
    R.write_psp(str(path), 0.0, comps)
    rd = R.PSPout([str(path)])
    ctr = [0.1, -0.2, 0.05]
    fg = Seq([0.0], [-2.0, -1.5, -1.0], [2.0, 1.5, 1.0], [16, 12, 0])
    got = fg.histo2d(rd, ctr)
    want = oracle.histo2d(comps[0]["mass"], comps[0]["pos"], ctr, fg.pmin, fg.pmax, fg.grid)
    assert set(got) == set(want) == {"xy"} and got["xy"].shape == (16, 12)
    assert np.allclose(got["xy"], want["xy"], rtol=1e-5, atol=0) and want["xy"].sum() > 0
    for proj in ("xy", "xz", "yz", "r"):
        assert np.allclose(fg.histo1d(rd, 2.5, 20, proj, ctr), oracle.histo1d(comps[0]["mass"], comps[0]["pos"], ctr, 2.5, 20, proj),
                           rtol=1e-5, atol=0)
    for a, b in zip(fg.histo1dlog(rd, 0.05, 3.0, 15, ctr),
                    oracle.histo1dlog(comps[0]["mass"], comps[0]["pos"], comps[0]["vel"], ctr, 0.05, 3.0, 15)):
        assert np.allclose(a, b, rtol=2e-4, atol=0)
    with pytest.raises(RuntimeError):
        fg.histo1d(rd, 1.0, 4, "zz")
    with pytest.raises(RuntimeError):
        fg.histo1dlog(rd, 0.0, 1.0, 4)


def test_truncated_and_malformed_files_fail_loudly(tmp_path):
    rng = np.random.default_rng(8)
    comps = [_comp(rng, "dark", 50, False, 0, 0), _comp(rng, "star", 20, False, 0, 0)]
    path = tmp_path / "OUT.t"
    R.write_psp(str(path), 0.0, comps)
    raw = path.read_bytes()
    (tmp_path / "cut1").write_bytes(raw[: len(raw) - 100])          # inside the last component's particles
    rd = R.PSPout([str(tmp_path / "cut1")])
    assert rd.GetTypes() == ["dark", "star"]
    rd.SelectType("dark")
    assert len(rd.arrays()["mass"]) == 50
    rd.SelectType("star")
    with pytest.raises(RuntimeError, match="ends inside"):
        rd.arrays()
    (tmp_path / "cut2").write_bytes(raw[: 16 + 8 + 16 + 1024 + 50 * 64 + 4])   # inside the second magic
    with pytest.raises(RuntimeError, match="magic"):
        R.PSPout([str(tmp_path / "cut2")])
    (tmp_path / "cut3").write_bytes(raw[:10])
    with pytest.raises(RuntimeError, match="master header"):
        R.PSPout([str(tmp_path / "cut3")])
    with pytest.raises(RuntimeError):
        R.PSPout([str(tmp_path / "nonexistent")])
    bad = dict(comps[0], info="parameters: {indexing: false}\n")    # YAML without a name
    R.write_psp(str(tmp_path / "noname"), 0.0, [bad])
    with pytest.raises(RuntimeError, match="name"):
        R.PSPout([str(tmp_path / "noname")])
    # an empty stanza prints its summary without statistics
    R.write_psp(str(tmp_path / "empty"), 0.0, [_comp(rng, "gas", 0, False, 0, 0)])
    out = io.StringIO()
    R.PSPout([str(tmp_path / "empty")]).PrintSummary(stats=True, out=out)
    assert "nbod :: 0" in out.getvalue() and "Position" not in out.getvalue()
