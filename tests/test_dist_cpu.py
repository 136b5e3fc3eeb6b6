"""N>1 path on CPU: world_size-2 gloo.  Particles are block-sharded, every rank forms its partial
coefficient set (here with the CPU oracle as the stand-in for the device accumulation), and ONE
all-reduce of the contiguous buffer must reproduce the single-rank coefficients."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from exp_amd.dist import allreduce_coefs_, shard_range
    from exp_amd.models import sample_sphere
    from tests.conftest import make_grid
    from tests.oracle_lib import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, g = make_grid("plummer", 4, 8, 400)
    m, pos, _ = sample_sphere(model, 3001, seed=42, velocities=False)
    pos[:, 2] *= 0.7
    orc = Oracle()
    prm = orc.params(rmin=g.rmin, rmax=g.rmax)
    n0, n1 = shard_range(len(m), rank, world)
    part, used = orc.sph_accumulate(g, prm, pos[n0:n1], m[n0:n1])
    allreduce_coefs_(part)
    full, _ = orc.sph_accumulate(g, prm, pos, m)
    err = np.abs(part - full).max() / np.abs(full).max()
    q.put((rank, n0, n1, float(err)))
    dist.destroy_process_group()


def test_shard_ranges_cover():
    from exp_amd.dist import shard_range
    for n in (0, 1, 7, 100, 10 ** 8 + 3):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            for a, b in zip(edges[:-1], edges[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


def test_level_balanced_shards():
    """shard_indices / shard_by_level: disjoint, covering, and every rank within one of N_level / world in every level."""
    from exp_amd.dist import shard_by_level, shard_indices
    rng = np.random.default_rng(3)
    for n in (0, 1, 10, 1001):
        for w in (1, 2, 3, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            allidx = np.sort(np.concatenate(parts)) if n else np.zeros(0, dtype=np.int64)
            assert np.array_equal(allidx, np.arange(n)) and max(map(len, parts)) - min(map(len, parts)) <= 1
    # radius-ordered levels (deep levels first): the block partition is badly skewed, the two others are not
    n, w = 20000, 8
    lev = np.sort(rng.choice(5, size=n, p=[0.02, 0.05, 0.13, 0.3, 0.5]))[::-1].copy()      # level 4 ... 0 along the input
    full = np.bincount(lev, minlength=5)
    parts = [shard_by_level(lev, r, w) for r in range(w)]
    assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(n))
    for r in range(w):
        mine = np.bincount(lev[parts[r]], minlength=5)
        assert np.all(np.abs(mine - full / w) < 1.0 + 1e-9), (r, mine, full)
        strided = np.bincount(lev[shard_indices(n, r, w)], minlength=5)
        assert np.all(np.abs(strided - full / w) <= 1.0 + 1e-9)
    from exp_amd.dist import shard_range
    n0, n1 = shard_range(n, 0, w)
    assert np.bincount(lev[n0:n1], minlength=5)[4] == n1 - n0 > 2 * full[4] / w          # (the control)
    assert len(shard_by_level(np.zeros(0, dtype=np.int32), 0, 2)) == 0


def test_two_rank_gloo_allreduce_matches_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n0, n1, err in res:
        assert err < 1e-13, (rank, err)


def _reader_worker(rank, world, port, path, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from exp_amd import reader as R
    from exp_amd.field import FieldGenerator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rd = R.ParticleReader.createReader("PSPout", [path])
    assert (rd.numprocs, rd.myid) == (world, rank)            # ParticleReader() asks the process group as the reference asks MPI
    rd.SelectType("dark")
    mine = rd.arrays()["indx"].astype(np.int64)
    # every particle on exactly one rank, dealt round-robin
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([len(mine)]))
    buf = [torch.zeros(int(s), dtype=torch.int64) for s in sizes]
    pad = torch.from_numpy(mine)
    if len({int(s) for s in sizes}) == 1:
        dist.all_gather(buf, pad)
    else:                                                      # ragged shares: gather padded
        big = max(int(s) for s in sizes)
        bufp = [torch.zeros(big, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(bufp, torch.nn.functional.pad(pad, (0, big - len(mine))))
        buf = [b[: int(s)] for b, s in zip(bufp, sizes)]
    # the histogram: each rank bins its share in float, the float sums are added (MPI_Reduce(MPI_FLOAT, MPI_SUM))
    fg = FieldGenerator([0.0], [-1, -1, -1], [1, 1, 1], [8, 8, 0])
    got = fg.histo1d(rd, 2.0, 12, "r")
    parts = []
    for r in range(world):
        solo = R.PSPout([path])
        solo.numprocs, solo.myid = world, r
        solo.SelectType("dark")
        one = FieldGenerator([0.0], [-1, -1, -1], [1, 1, 1], [8, 8, 0])
        one._reduce_f32 = staticmethod(lambda a: a)
        a = solo.arrays()
        rad = np.sqrt((a["pos"] ** 2).sum(axis=1))
        bins = np.where(np.floor(rad / (2.0 / 12)) < 12, np.floor(rad / (2.0 / 12)), -1).astype(np.int64)
        parts.append(one._binsum(bins, a["mass"], 12))
    q.put((rank, [b.numpy().tolist() for b in buf], got.tolist(), [p.tolist() for p in parts]))
    dist.destroy_process_group()


def test_two_ranks_share_a_phase_space_file(tmp_path):
    """The readers deal a file over the ranks of the process group as the reference deals it over MPI ranks, and the
    particle histograms add the ranks' float sums (world_size 2, gloo; the float accumulation is the C-ABI's host loop)."""
    import torch.multiprocessing as mp
    from exp_amd import reader as R
    if not os.path.exists(os.path.join(ROOT, "exp_amd", "libexp_amd.so")):
        pytest.skip("exp_amd/libexp_amd.so not built")
    rng = np.random.default_rng(3)
    n = 1001
    comp = dict(info=R.component_info("dark", "sphereSL", {}, {"indexing": True}), mass=rng.uniform(1, 2, n) / n,
                pos=rng.normal(0, 0.6, (n, 3)), vel=rng.normal(size=(n, 3)), indx=(rng.permutation(n) + 1).astype(np.uint64))
    path = str(tmp_path / "OUT.shared")
    R.write_psp(path, 0.0, [comp])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_reader_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, shares, got, parts in res:
        assert shares[0] == comp["indx"][0::2].tolist() and shares[1] == comp["indx"][1::2].tolist()
        want = (np.array(parts[0], dtype=np.float32) + np.array(parts[1], dtype=np.float32))
        i = np.arange(12)
        want = (want.astype(np.float64) / (4.0 * 3.14159265358979323846 / 3.0 * (2.0 / 12) ** 3 * (3 * i * (i + 1) + 1))).astype(np.float32)
        assert np.array_equal(np.array(got, dtype=np.float32), want) and want.sum() > 0
