"""Particle sharding and the one coefficient all-reduce per accumulation.

EXP reduces the coefficient rows with (L+1)^2 host MPI_Allreduce calls
(``src/SphericalBasis.cc:864-903``).  Here particles are block-sharded over ranks (one process
per GPU) and the whole contiguous coefficient buffer is reduced ONCE, in place, on the compute
stream: RCCL through ``torch.distributed`` (backend "nccl") on GPUs, gloo on CPU for tests.
No other data-path collective exists: particles never migrate between ranks.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def shard_range(ntot: int, rank: int, world: int) -> Tuple[int, int]:
    """Static BLOCK partition [n0, n1) of ntot particles (every rank within one of equal).

    Balanced in particle count only: on an input ordered by radius or binding energy (what the reference's `gensph`
    writes) the blocks are radial shells, and under block multistep the deep time-step levels all land on the rank
    that holds the centre.  Use it for single-level runs and for inputs in random order; `shard_indices` /
    `shard_by_level` below are what a multistep run wants (each GPU owns ~N/world of EVERY level).
    """
    return ntot * rank // world, ntot * (rank + 1) // world


def shard_indices(ntot: int, rank: int, world: int) -> np.ndarray:
    """STRIDED partition: rank r owns the particles r, r + world, r + 2 world, ... (every rank within one of equal).

    Whatever varies smoothly along the input order -- radius, energy, hence the time-step level a particle will be
    given -- is dealt evenly to all ranks, so every rank holds ~1/world of every level and the sub-steps of a block-
    multistep run cost every GPU the same; the reference reaches the same end with its rate-weighted `load_balance`
    (src/Component.cc:3780, :3868).  The caller index of a rank's k-th particle is rank + k * world (downloads come back
    in that order)."""
    return np.arange(rank, ntot, world, dtype=np.int64)


def shard_by_level(levels, rank: int, world: int) -> np.ndarray:
    """EXACT level balance for a known level assignment (e.g. the one `begin_run` produced, or a restart file's): within
    each level, in input order, the k-th particle goes to rank k mod world (rotated per level so that the remainders
    do not pile up on rank 0).  Every rank's population of every level is within one of N_level / world.  Returns the
    sorted caller indices owned by `rank`."""
    lev = np.asarray(levels)
    order = np.argsort(lev, kind="stable")                     # level-major, input order within a level
    counts = np.bincount(lev.astype(np.int64)) if len(lev) else np.zeros(0, dtype=np.int64)
    start = np.concatenate([[0], np.cumsum(counts)[:-1]]) if len(counts) else np.zeros(0, dtype=np.int64)
    k = np.arange(len(lev)) - np.repeat(start, counts)         # position of each sorted entry within its level
    owner = (k + np.repeat(np.arange(len(counts)), counts)) % world
    return np.sort(order[owner == rank])


class _DevPtr:
    """Minimal __cuda_array_interface__ view of `count` doubles at a raw device pointer."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 3, "strides": None}


def torch_allreduce_callback(device=None, group=None):
    """Callback for Context.set_allreduce: SUM-reduce the device coefficient buffer in place with
    torch.distributed (RCCL) ON THE STREAM IT IS HANDED -- the context's own, or the auxiliary stream of the two-stream
    step driver -- which is made torch's current stream for the call when it is not already."""
    import torch
    import torch.distributed as dist

    views = {}          # (ptr, count) -> tensor view: the buffers are long-lived, wrap each once
    streams = {}        # raw stream -> torch.cuda.ExternalStream

    def fn(ptr: int, count: int, stream: int) -> None:
        t = views.get((ptr, count))
        if t is None:
            t = views[(ptr, count)] = torch.as_tensor(_DevPtr(ptr, count), device=device)
        if stream and stream != torch.cuda.current_stream(device).cuda_stream:
            ext = streams.get(stream)
            if ext is None:
                ext = streams[stream] = torch.cuda.ExternalStream(stream, device=device)
            with torch.cuda.stream(ext):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    return fn


def host_staged_allreduce_callback(group=None):
    """Callback for Context.set_allreduce that stages the buffer through the host: stream sync, D2H copy, a
    torch.distributed all-reduce of the host array (gloo, or whatever backend `group` has), H2D copy.  For ranks that
    SHARE a device (RCCL refuses that: test boxes, rehearsals of a multi-rank launch on one GPU) and for hosts whose
    only collective is a CPU one.  Costs two synchronous copies per reduction; not a production path."""
    import ctypes

    import torch
    import torch.distributed as dist
    hip = ctypes.CDLL("libamdhip64.so")

    def fn(ptr: int, count: int, stream: int) -> None:
        buf = np.empty(count)
        if hip.hipStreamSynchronize(ctypes.c_void_p(stream)) != 0:
            raise RuntimeError("hipStreamSynchronize failed")
        if hip.hipMemcpy(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(count * 8), 2) != 0:
            raise RuntimeError("hipMemcpy D2H failed")
        dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM, group=group)
        if hip.hipMemcpy(ctypes.c_void_p(ptr), buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(count * 8), 1) != 0:
            raise RuntimeError("hipMemcpy H2D failed")

    return fn


def allreduce_coefs_(coef: np.ndarray, group=None) -> np.ndarray:
    """Host-array flavour (gloo / CPU tests, or host-staged MPI-style callers): in place."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(coef)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return coef
