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
