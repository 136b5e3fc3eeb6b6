"""``createFromReader`` and the particle histograms on files read by exp_amd.reader (GPU: the coefficient accumulation is
the device's; the histograms' float sums go through the C-ABI's exp_amd_host_binsum_f32)."""
import os

import numpy as np
import pytest

from exp_amd import reader as R

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SPH = """
id : sphereSL
parameters :
  numr: 1000
  rmin: 0.0001
  rmax: 1.95
  Lmax: 4
  nmax: 10
  rmapping : 0.0667
  modelname: {model}
  cachename: {cache}
"""
CYL = """
id: cylinder
parameters:
  acyl: 0.01
  hcyl: 0.001
  mmax: 2
  nmax: 4
  ncylodd: 1
  ncylnx: 32
  ncylny: 16
  ncylr: 400
  lmaxfid: 10
  nmaxfid: 8
  rnum: 30
  tnum: 20
  cachename: {cache}
"""


def _rot(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return q * np.sign(np.linalg.det(q))


@pytest.fixture(scope="module")
def snapshot(tmp_path_factory):
    """one PSP file in doubles with a halo (indexed, shuffled indices) and a disk component, and its float twin"""
    from exp_amd.models import sample_disk
    d = tmp_path_factory.mktemp("psp")
    rng = np.random.default_rng(21)
    nh = 6000
    halo = dict(info=R.component_info("dark halo", "sphereSL", {"Lmax": 4, "nmax": 10}, {"nlevel": 1, "indexing": True}),
                mass=rng.uniform(0.5, 1.5, nh) / nh, pos=rng.normal(0, 0.3, (nh, 3)) * np.array([1.0, 0.8, 0.5]),
                vel=rng.normal(0, 0.4, (nh, 3)), pot=np.zeros(nh), indx=(rng.permutation(nh) + 1).astype(np.uint64))
    m, pos, vel = sample_disk(5000, 3, a=0.01, h=0.001)
    disk = dict(info=R.component_info("star disk", "cylinder", {"mmax": 2}, {"nlevel": 1}), mass=m, pos=pos, vel=vel)
    paths = {}
    for real4 in (False, True):
        paths[real4] = str(d / f"OUT.run0.{int(real4):05d}")
        R.write_psp(paths[real4], 0.125, [halo, disk], real4)
    return paths, halo, disk


def test_create_from_reader_is_create_from_array(snapshot, tmp_path):
    """BiorthBasis::createFromReader (expui/BiorthBasis.cc:4517-4581) on a PSP file against createFromArray on the
    arrays the file was written from: same coefficients to 1e-13 (double file; the transformed positions come out of two differently strided matrix products), the
    reader's time, centre and rotation in the structure; the float file gives the coefficients of the narrowed
    particles.  Both bases."""
    from exp_amd.basis import Basis
    paths, halo, disk = snapshot
    rng = np.random.default_rng(2)
    ctr, rot = rng.normal(0, 0.02, 3), _rot(rng)
    sph = Basis.factory(SPH.format(model=os.path.join(GOLD, "SLGridSph.model"), cache=str(tmp_path / "sl.cache")))
    cyl = Basis.factory(CYL.format(cache=str(tmp_path / "eof.cache")))
    for basis, comp, name in ((sph, halo, "dark halo"), (cyl, disk, "star disk")):
        c0 = ctr if basis is sph else ctr * 0.05
        for real4 in (False, True):
            rd = R.ParticleReader.createReader("PSPout", [paths[real4]])
            rd.SelectType(name)
            got = basis.createFromReader(rd, c0, rot)
            pos, m = comp["pos"], comp["mass"]
            if real4:
                pos, m = pos.astype(np.float32).astype(np.float64), m.astype(np.float32).astype(np.float64)
            want = basis.createFromArray(m, pos, time=0.125, center=c0, rot=rot, posvelrows=False)
            assert got.time == 0.125 and np.array_equal(got.ctr, c0) and np.array_equal(got.rot, rot)
            assert np.abs(got.coefs - want.coefs).max() <= 1e-13 * np.abs(want.coefs).max(), (name, real4)
            assert np.abs(got.coefs).max() > 0
        # default centre and rotation
        rd = R.PSPout([paths[False]])
        rd.SelectType(name)
        plain = basis.createFromReader(rd)
        ref = basis.createFromArray(comp["mass"], comp["pos"], time=0.125).coefs
        assert np.abs(plain.coefs - ref).max() <= 1e-13 * np.abs(ref).max()
        assert np.array_equal(plain.rot, np.eye(3)) and not np.any(plain.ctr)


def test_selector_sees_the_files_index_and_rotated_velocity(snapshot, tmp_path):
    """The functor of setSelector is called with (mass, transformed position, rotated velocity, p->indx) and the particle's
    own index is what accumulate is handed (expui/BiorthBasis.cc:4561-4578)."""
    from exp_amd.basis import Basis
    paths, halo, _ = snapshot
    rng = np.random.default_rng(4)
    ctr, rot = rng.normal(0, 0.02, 3), _rot(rng)
    sph = Basis.factory(SPH.format(model=os.path.join(GOLD, "SLGridSph.model"), cache=str(tmp_path / "sl.cache")))
    rd = R.PSPout([paths[False]])
    rd.SelectType("dark halo")
    seen = []

    def ftor(m, p, v, i):
        seen.append((m, p.copy(), v.copy(), i))
        return i % 3 == 0
    sph.setSelector(ftor)
    got = sph.createFromReader(rd, ctr, rot)
    sph.clrSelector()
    assert len(seen) == len(halo["mass"])
    k = 17
    assert seen[k][3] == int(halo["indx"][k]) and seen[k][0] == halo["mass"][k]
    assert np.allclose(seen[k][1], rot @ (halo["pos"][k] - ctr), rtol=0, atol=1e-15)
    assert np.allclose(seen[k][2], rot @ halo["vel"][k], rtol=0, atol=1e-15)
    keep = halo["indx"] % 3 == 0
    want = sph.createFromArray(halo["mass"][keep], halo["pos"][keep], time=0.125, center=ctr, rot=rot)
    assert np.abs(got.coefs - want.coefs).max() <= 1e-13 * np.abs(want.coefs).max()
    assert np.abs(got.coefs - sph.createFromArray(halo["mass"], halo["pos"], center=ctr, rot=rot).coefs).max() > \
        1e-3 * np.abs(want.coefs).max()                       # the selection matters


def test_particle_histograms_bit_for_bit(snapshot, oracle):
    """FieldGenerator::histogram2d / histogram1d / histo1dlog (expui/FieldGenerator.cc:776-1009) on a reader against the
    oracle's particle-at-a-time restatement: the same floats."""
    from exp_amd.field import FieldGenerator
    paths, halo, disk = snapshot
    rd = R.PSPout([paths[False]])
    rd.SelectType("dark halo")
    ctr = [0.01, -0.02, 0.005]
    for grid in ([24, 20, 0], [16, 0, 12], [0, 8, 8], [10, 12, 14]):
        fg = FieldGenerator([0.0], [-0.8, -0.6, -0.5], [0.8, 0.6, 0.5], grid)
        got = fg.histo2d(rd, ctr)
        want = oracle.histo2d(halo["mass"], halo["pos"], ctr, fg.pmin, fg.pmax, grid)
        assert set(got) == set(want)
        for k in got:
            assert got[k].dtype == np.float32 and np.array_equal(got[k], want[k]) and got[k].sum() > 0, (grid, k)
    fg = FieldGenerator([0.0], [-1, -1, -1], [1, 1, 1], [4, 4, 0])
    for proj in ("xy", "xz", "yz", "r"):
        assert np.array_equal(fg.histo1d(rd, 1.2, 30, proj, ctr), oracle.histo1d(halo["mass"], halo["pos"], ctr, 1.2, 30, proj))
    got = fg.histo1dlog(rd, 0.02, 2.0, 25, ctr)
    want = oracle.histo1dlog(halo["mass"], halo["pos"], halo["vel"], ctr, 0.02, 2.0, 25)
    for a, b in zip(got, want):
        assert a.dtype == np.float32 and np.array_equal(a, b)
    assert got[1].max() > 0 and got[2].max() > 0
    # the disk from the same file, default centre
    rd.SelectType("star disk")
    fg = FieldGenerator([0.0], [-0.05, -0.05, -0.01], [0.05, 0.05, 0.01], [32, 32, 0])
    assert np.array_equal(fg.histo2d(rd)["xy"], oracle.histo2d(disk["mass"], disk["pos"], [0, 0, 0], fg.pmin, fg.pmax, fg.grid)["xy"])


def test_component_restart_from_a_phase_space_file(snapshot, tmp_path):
    """A component uploaded from a PSP file, advanced one step, written back and read again: the file carries the device
    state bit for bit (doubles), with the potential in the record's eighth real."""
    from exp_amd.runtime import Component, Context
    paths, halo, _ = snapshot
    ctx = Context(0)
    rd = R.PSPout([paths[False]])
    rd.SelectType("dark halo")
    a = rd.arrays()
    c = Component.from_arrays(ctx, a["mass"], a["pos"], a["vel"])
    c.upload_acc(np.zeros((len(a["mass"]), 3)), np.linspace(-1.0, 0.0, len(a["mass"])))
    c.incr_position(0.01)
    out = c.download(("mass", "pos", "vel", "pot"))
    path = str(tmp_path / "OUT.restart")
    R.write_psp(path, 0.135, [dict(info=rd.stanzas[0].info.split(b"\0")[0].decode(), indx=a["indx"], **out)])
    back = R.PSPout([path])
    assert back.CurrentTime() == 0.135 and back.GetTypes() == ["dark halo"]
    b = back.arrays()
    for k in ("mass", "pos", "vel", "pot"):
        assert np.array_equal(b[k], out[k])
    assert np.array_equal(b["indx"], halo["indx"]) and np.allclose(b["pos"], halo["pos"] + 0.01 * halo["vel"], rtol=0, atol=1e-15)
    c.close()


def test_upload_frame_is_the_host_transform():
    """exp_amd_comp_upload_frame: positions as the caller holds them ([n, 3] C-ordered = stride 3, or the [n, 3] view of a
    [3, n] array = three columns) taken into rot (x - ctr) on the device, velocities rotated only -- against the numpy
    expression addFromArray used to evaluate on the host, to a few ulps; no frame = the bits of the input."""
    from exp_amd.runtime import Component, Context
    ctx = Context(0)
    rng = np.random.default_rng(6)
    for n in (1, 2, 3, 1000, 4097):
        pos, vel, m = rng.normal(size=(n, 3)), rng.normal(size=(n, 3)), rng.uniform(1, 2, n)
        ctr, rot = rng.normal(0, 0.1, 3), _rot(rng)
        for p, v in ((pos, vel), (np.ascontiguousarray(pos.T).T, np.ascontiguousarray(vel.T).T), (pos, np.ascontiguousarray(vel.T).T),
                     (pos[::-1][::-1], None)):
            c = Component.from_frame(ctx, m, p, v, ctr, rot)
            out = c.download(("mass", "pos", "vel"))
            want = (pos - ctr) @ rot.T
            assert np.abs(out["pos"] - want).max() <= 8e-16 * max(1.0, np.abs(want).max())
            if v is not None:
                assert np.abs(out["vel"] - vel @ rot.T).max() <= 8e-16 * max(1.0, np.abs(vel).max() * 2)
            else:
                assert not np.any(out["vel"])
            assert np.array_equal(out["mass"], m)
            c.close()
            c = Component.from_frame(ctx, m, p, v)                    # no frame: nothing is computed
            out = c.download(("pos", "vel"))
            assert np.array_equal(out["pos"], pos) and (v is None or np.array_equal(out["vel"], vel))
            c.close()
        c = Component.from_frame(ctx, m, pos, None, ctr, None)        # shift only
        assert np.array_equal(c.download(("pos",))["pos"], pos - ctr)
        c.close()
    ctx.close()
