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
    """Static block partition [n0, n1) of ntot particles (every rank within one of equal)."""
    return ntot * rank // world, ntot * (rank + 1) // world


class _DevPtr:
    """Minimal __cuda_array_interface__ view of `count` doubles at a raw device pointer."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 3, "strides": None}


def torch_allreduce_callback(device=None, group=None):
    """Callback for Context.set_allreduce: SUM-reduce the device coefficient buffer in place with
    torch.distributed (RCCL).  The context must run on torch's current stream."""
    import torch
    import torch.distributed as dist

    views = {}          # (ptr, count) -> tensor view: the buffers are long-lived, wrap each once

    def fn(ptr: int, count: int, stream: int) -> None:
        t = views.get((ptr, count))
        if t is None:
            t = views[(ptr, count)] = torch.as_tensor(_DevPtr(ptr, count), device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    return fn


def allreduce_coefs_(coef: np.ndarray, group=None) -> np.ndarray:
    """Host-array flavour (gloo / CPU tests, or host-staged MPI-style callers): in place."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(coef)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return coef
