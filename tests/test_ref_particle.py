"""The phase-space record, the component header, the body-file line and the Tipsy native reader of the REFERENCE ITSELF
(exputil/Particle.cc, exputil/header.cc, include/tipsy.H compiled where they lie into oracle/_ref/libref_particle.so by
oracle/ref/Makefile -- they need only <mpi.h> / libmpi, which the image has under /opt/conda) against exp_amd/reader.py
and the oracle's restatement (oracle/psp_oracle.c): the same BYTES out of `Particle::writeBinary` /
`ComponentHeader::write`, the same values out of `Particle::readBinary`, `Particle::readAscii` and
`TipsyReader::TipsyNative`.  This is the reference-side pin of the PSP format: with it the restatement in
oracle/psp_oracle.c is itself checked against reference code.

CPU only; skipped where neither the reference tree nor a prebuilt library exists."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

from exp_amd import reader as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libref_particle.so")


@pytest.fixture(scope="module")
def ref():
    if not os.path.exists(LIB) and os.path.exists("/root/reference/exputil/Particle.cc"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref")], check=False)
    if not os.path.exists(LIB):
        pytest.skip("no oracle/_ref/libref_particle.so (needs the reference tree and <mpi.h> to build)")
    try:
        return ctypes.CDLL(LIB)
    except OSError as e:
        pytest.skip(f"oracle/_ref/libref_particle.so does not load here: {e}")


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _comp(rng, name, n, indexing, ni, nd, pad=0):
    c = dict(info=R.component_info(name, "sphereSL", {"note": "x" * pad} if pad else {"Lmax": 2}, {"nlevel": 1, "indexing": indexing}),
             indexing=indexing, mass=rng.uniform(1, 2, n) / max(n, 1), pos=rng.normal(size=(n, 3)), vel=rng.normal(size=(n, 3)),
             pot=rng.normal(size=n), potext=rng.normal(size=n) * 0.1, name=name)
    c["iattrib"] = rng.integers(-2 ** 31, 2 ** 31 - 1, (n, ni)).astype(np.int32)
    c["dattrib"] = rng.normal(size=(n, nd))
    c["indx"] = (rng.permutation(n) + 1).astype(np.uint64) * 5 if indexing else np.arange(1, n + 1, dtype=np.uint64)
    return c


def _ref_write(ref, path, time, comps, real4):
    n = [len(c["mass"]) for c in comps]
    arr_i = lambda v: (ctypes.c_int * len(v))(*[int(x) for x in v])
    cat = lambda key: np.ascontiguousarray(np.concatenate([np.asarray(c[key], dtype=np.float64).reshape(-1) for c in comps]))
    infos = (ctypes.c_char_p * len(comps))(*[(c["info"] if c["info"].endswith("\n") else c["info"] + "\n").encode() for c in comps])
    indx = np.ascontiguousarray(np.concatenate([c["indx"] for c in comps]).astype(np.uint64))
    ia = np.ascontiguousarray(np.concatenate([c["iattrib"].reshape(-1) for c in comps]).astype(np.int32))
    da = np.ascontiguousarray(np.concatenate([c["dattrib"].reshape(-1) for c in comps]).astype(np.float64))
    m, p, v, ph, px = cat("mass"), cat("pos"), cat("vel"), cat("pot"), cat("potext")
    rc = ref.ref_psp_write(str(path).encode(), ctypes.c_double(time), len(comps), arr_i(n), arr_i([c["iattrib"].shape[1] for c in comps]),
                           arr_i([c["dattrib"].shape[1] for c in comps]), infos, arr_i([int(c["indexing"]) for c in comps]),
                           int(real4), _vp(indx), _vp(m), _vp(p), _vp(v), _vp(ph), _vp(px), _vp(ia), _vp(da))
    assert rc == 0, rc


@pytest.mark.parametrize("real4", [False, True])
def test_the_references_writer_gives_the_same_bytes(ref, oracle, tmp_path, real4):
    """Particle::writeBinary + ComponentHeader::write of the reference, driven in the order of OutPSN::Run /
    Component::write_binary, against `write_psp` and against the oracle's restatement: identical files.  Components:
    indexed with attributes, plain, empty, and one whose stanza outgrows the default info field."""
    rng = np.random.default_rng(41)
    comps = [_comp(rng, "dark", 300, True, 2, 1), _comp(rng, "star", 77, False, 0, 3), _comp(rng, "gas", 0, False, 1, 1),
             _comp(rng, "bulge", 9, True, 0, 0, pad=1400)]
    a, b, c = tmp_path / "ref", tmp_path / "mine", tmp_path / "orc"
    _ref_write(ref, a, 2.5, comps, real4)
    R.write_psp(str(b), 2.5, comps, real4)
    oracle.psp_write(c, 2.5, comps, real4)
    raw = a.read_bytes()
    assert raw == b.read_bytes() and raw == c.read_bytes()
    assert struct.unpack("<dii", raw[:16]) == (2.5, 386, 4) and struct.unpack("<Q", raw[16:24])[0] == 0xadbfabc0 + (4 if real4 else 8)
    # and the reader on the reference's file
    rd = R.PSPout([str(a)])
    assert rd.GetTypes() == ["dark", "star", "gas", "bulge"]
    for comp in comps:
        rd.SelectType(comp["name"])
        got = rd.arrays()
        f = (lambda x: np.asarray(x).astype(np.float32).astype(np.float64)) if real4 else np.asarray
        assert np.array_equal(got["pos"], f(comp["pos"])) and np.array_equal(got["pot"], f(comp["pot"] + comp["potext"]))
        if comp["indexing"]:
            assert np.array_equal(got["indx"], comp["indx"])


@pytest.mark.parametrize("real4", [False, True])
def test_the_references_reader_on_files_written_here(ref, tmp_path, real4):
    """ComponentHeader::read + Particle::readBinary (the restart path of Component) on a file `write_psp` wrote: every
    field back, the info string intact, sequence numbers from 1 where the file holds no index (Component's convention;
    ParticleReader's PParticle numbers from 0, include/ParticleReader.H:283)."""
    rng = np.random.default_rng(43)
    comps = [_comp(rng, "dark", 120, True, 1, 2), _comp(rng, "star", 33, False, 2, 0)]
    path = tmp_path / "OUT.mine"
    R.write_psp(str(path), 0.5, comps, real4)
    idx = (ctypes.c_int * 2)(1, 0)
    f = (lambda x: np.asarray(x).astype(np.float32).astype(np.float64)) if real4 else np.asarray
    for which, c in enumerate(comps):
        n = len(c["mass"])
        ni, nd, ninfo, rsize = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_ulong()
        info = ctypes.create_string_buffer(4096)
        indx, mass, pos, vel, pot = np.zeros(n, np.uint64), np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
        ia, da = np.zeros((n, 4), np.int32), np.zeros((n, 4))
        ref.ref_psp_read.restype = ctypes.c_long
        got = ref.ref_psp_read(str(path).encode(), which, idx, ctypes.byref(ni), ctypes.byref(nd), ctypes.byref(ninfo), info, 4096,
                               ctypes.byref(rsize), _vp(indx), _vp(mass), _vp(pos), _vp(vel), _vp(pot), _vp(ia), _vp(da), n)
        assert got == n and (ni.value, nd.value, ninfo.value, rsize.value) == (c["iattrib"].shape[1], c["dattrib"].shape[1], 1024, 4 if real4 else 8)
        assert info.value.decode() == c["info"]
        assert np.array_equal(mass, f(c["mass"])) and np.array_equal(pos, f(c["pos"])) and np.array_equal(vel, f(c["vel"]))
        assert np.array_equal(pot, f(c["pot"] + c["potext"]))
        assert np.array_equal(indx, c["indx"] if c["indexing"] else np.arange(1, n + 1))
        assert np.array_equal(ia.reshape(-1)[: n * ni.value].reshape(n, ni.value), c["iattrib"])
        assert np.array_equal(da.reshape(-1)[: n * nd.value].reshape(n, nd.value), f(c["dattrib"]))


def test_body_file_lines_as_the_reference_reads_them(ref, tmp_path):
    """Particle::readAscii line by line against read_bodies_ascii: a file written here; a hand-written one with short
    lines (attributes the line does not hold are zero) and an index column."""
    rng = np.random.default_rng(47)
    n = 40
    m, pos, vel = rng.uniform(size=n), rng.normal(size=(n, 3)) * 1e3, rng.normal(size=(n, 3)) * 1e-3
    ia, da = rng.integers(-99, 99, (n, 2)), rng.normal(size=(n, 3))
    paths = [(tmp_path / "a.bods", False), (tmp_path / "b.bods", True), (tmp_path / "c.bods", False)]
    R.write_bodies_ascii(str(paths[0][0]), m, pos, vel, ia, da)
    R.write_bodies_ascii(str(paths[1][0]), m, pos, vel, ia, da, indx=np.arange(n) * 3 + 7)
    paths[2][0].write_text("3 2 2\n1.0 0 0 0 0 0 0 7\n2.5e-1 1 1 1 -1 -1 -1 8 9 0.5\n3 1e-3 2 3 4 5 6 1 2 3.5 4.5 extra\n")
    ref.ref_bodies_read.restype = ctypes.c_long
    for path, aindex in paths:
        mine = R.read_bodies_ascii(str(path), aindex=aindex)
        k = len(mine["mass"])
        ni, nd = ctypes.c_int(), ctypes.c_int()
        indx, mass, p, v = np.zeros(k, np.uint64), np.zeros(k), np.zeros((k, 3)), np.zeros((k, 3))
        iat, dat = np.zeros((k, 8), np.int32), np.zeros((k, 8))
        got = ref.ref_bodies_read(str(path).encode(), int(aindex), k, ctypes.byref(ni), ctypes.byref(nd), _vp(indx), _vp(mass), _vp(p),
                                  _vp(v), _vp(iat), _vp(dat), 8)
        assert got == k
        assert np.array_equal(indx, mine["indx"]) and np.array_equal(mass, mine["mass"])
        assert np.array_equal(p, mine["pos"]) and np.array_equal(v, mine["vel"])
        if ni.value:
            assert np.array_equal(iat.reshape(-1)[: k * ni.value].reshape(k, ni.value), mine["iattrib"])
        if nd.value:
            assert np.array_equal(dat.reshape(-1)[: k * nd.value].reshape(k, nd.value), mine["dattrib"])
    assert np.array_equal(R.read_bodies_ascii(str(paths[0][0]))["pos"], pos)        # %.17g round-trips through the reference too


def test_tipsy_native_as_the_reference_reads_it(ref, tmp_path):
    """TipsyReader::TipsyNative for 1, 2 and 3 ranks against the Tipsy reader here: the same block of every group per rank
    (nsize / numprocs each, the remainder to the last), the same index offsets; Bonsai ids as dark_particle::ID / ID2."""
    from tests.test_reader_cpu import _write_tipsy
    rng = np.random.default_rng(53)
    path = tmp_path / "snap.tipsy"
    _write_tipsy(path, 3.25, 11, 29, 7, rng)
    ref.ref_tipsy_read.restype = ctypes.c_long
    for nranks in (1, 2, 3):
        for rank in range(nranks):
            rd = R.Tipsy(str(path))
            rd.numprocs, rd.myid = nranks, rank
            for ptype, name in enumerate(("Gas", "Dark", "Star")):
                rd.SelectType(name)
                mine = rd.arrays()
                cap = 64
                time, off = ctypes.c_double(), ctypes.c_ulong()
                mass, pos, vel, phi = np.zeros(cap, np.float32), np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32), np.zeros(cap, np.float32)
                n = ref.ref_tipsy_read(str(path).encode(), nranks, rank, ptype, ctypes.byref(time), ctypes.byref(off), _vp(mass), _vp(pos),
                                       _vp(vel), _vp(phi), cap)
                assert n == len(mine["mass"]) and time.value == 3.25 == rd.CurrentTime()
                assert np.array_equal(mine["mass"], mass[:n].astype(np.float64)) and np.array_equal(mine["pos"], pos[:n].astype(np.float64))
                assert np.array_equal(mine["vel"], vel[:n].astype(np.float64))
                assert np.array_equal(mine["indx"], off.value + 1 + np.arange(n))       # getIndexOffset + pcount + 1
    b = R.Tipsy(str(path), "bonsai")
    b.SelectType("Dark")
    b1 = R.Tipsy(str(path), "bonsai1")
    b1.SelectType("Dark")
    raw = np.fromfile(path, dtype=R.TIPSY_DARK, count=29, offset=32 + 11 * 48)
    for k in range(29):
        i1, i2 = ctypes.c_int(), ctypes.c_ulong()
        ref.ref_tipsy_ids(ctypes.c_float(raw["eps"][k]), ctypes.c_float(raw["phi"][k]), ctypes.byref(i1), ctypes.byref(i2))
        assert int(b.arrays()["indx"][k]) == i2.value
        assert int(b1.arrays()["indx"][k]) == (i1.value & 0xffffffffffffffff)
