"""``pyEXP.util`` -- the two centre estimators that feed ``createFromReader(reader, center)`` and the particle iterator
(pyEXP/UtilWrappers.cc:58-140; expui/Centering.cc):

* ``getCenterOfMass(reader)``: sum m x / sum m over the reader's particles (all ranks);
* ``getDensityCenter(reader, stride=1, Nsort=0, Ndens=32)``: every sample point gets the density of its ``Ndens`` nearest
  neighbours -- ITSELF included: the samples are points of the set -- as (their summed mass) / (4 pi/3 r_N^3) / (total mass),
  r_N the distance to the farthest of them; the centre is the density-weighted mean position of the samples, or of the
  ``Nsort`` densest ones.  ``stride > 1`` takes the first nbods/stride of a random permutation as samples (the tree
  still holds every particle).  The reference builds its own KD tree (include/KDtree.H); the neighbours here come from
  scipy's cKDTree (exact k-nearest search, all cores);
* ``particleIterator(reader, func)``: ``func(mass, pos, vel, index)`` for every particle."""
from __future__ import annotations

import math
from typing import Callable, List, Optional

import numpy as np


def _all_ranks(a: np.ndarray) -> np.ndarray:
    """the rows of every rank, in rank order (the MPI_Bcast loop of expui/Centering.cc:36-63)"""
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return a
    parts: List[Optional[np.ndarray]] = [None] * dist.get_world_size()
    dist.all_gather_object(parts, a)
    return np.concatenate(parts)


def getCenterOfMass(reader) -> List[float]:
    a = reader.arrays()
    pm = _all_ranks(np.concatenate([a["pos"].astype(np.float64), a["mass"].astype(np.float64)[:, None]], axis=1))
    ctr, mastot = np.zeros(3), 0.0
    for k in range(3):
        ctr[k] = float(np.sum(pm[:, 3] * pm[:, k]))
    mastot = float(pm[:, 3].sum())
    if mastot > 0.0:
        ctr /= mastot
    return [float(v) for v in ctr]


def knn_density(pos: np.ndarray, mass: np.ndarray, samples: np.ndarray, Ndens: int):
    """density of every sample point (indices into pos) from its Ndens nearest neighbours, itself included -> (density,
    valid): 0 and False where the neighbour sphere has no volume"""
    from scipy.spatial import cKDTree
    tree = cKDTree(pos)
    k = min(int(Ndens), len(pos))
    d, j = tree.query(pos[samples], k=k, workers=-1)
    d, j = d.reshape(len(samples), k), j.reshape(len(samples), k)
    wgt = mass[j].sum(axis=1)
    volume = 4.0 * math.pi / 3.0 * d[:, -1] ** 3
    kdmass = float(mass.sum())
    ok = (volume > 0.0) & (kdmass > 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        dens = np.where(ok, wgt / volume / kdmass, 0.0)
    return dens, ok


def getDensityCenter(reader, stride: int = 1, Nsort: int = 0, Ndens: int = 32, seed: Optional[int] = None) -> List[float]:
    a = reader.arrays()
    pm = _all_ranks(np.concatenate([a["pos"].astype(np.float64), a["mass"].astype(np.float64)[:, None]], axis=1))
    pos, mass = np.ascontiguousarray(pm[:, :3]), np.ascontiguousarray(pm[:, 3])
    nbods = len(mass)
    if nbods == 0:
        raise RuntimeError("tree is empty")                   # kdtree::nearestN on an empty tree (include/KDtree.H:349)
    samples = np.arange(nbods)
    if stride > 1:
        # every rank evaluates the whole estimate on the gathered particle set: they must draw the SAME sample (the
        # reference splits one loop over the ranks and combines, expui/Centering.cc); an unseeded draw is rank 0's
        if seed is None:
            import sys
            dist = sys.modules.get("torch.distributed")
            box = [int(np.random.SeedSequence().entropy % (1 << 63))]
            if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.broadcast_object_list(box, src=0)
            seed = box[0]
        samples = np.random.default_rng(seed).permutation(nbods)[: nbods // stride]
    dens, ok = knn_density(pos, mass, samples, Ndens)
    samples, dens = samples[ok], dens[ok]
    if Nsort > 0 and len(dens) > Nsort:
        top = np.argsort(dens, kind="stable")[-Nsort:]
        samples, dens = samples[top], dens[top]
    ctr = (dens[:, None] * pos[samples]).sum(axis=0)
    dentot = float(dens.sum())
    if dentot > 0.0:
        ctr = ctr / dentot
    return [float(v) for v in ctr]


def particleIterator(reader, func: Callable) -> None:
    """``func(mass, pos, vel, index)`` for each particle of the reader (pyEXP/UtilWrappers.cc:108-140)"""
    a = reader.arrays()
    for i in range(len(a["mass"])):
        func(float(a["mass"][i]), [float(v) for v in a["pos"][i]], [float(v) for v in a["vel"][i]], int(a["indx"][i]))


def getVersionInfo():
    """(version, git branch, commit, build date) of THIS package where pyEXP reports EXP's"""
    from . import __version__ as v
    return {"version": v, "package": "exp_amd"}
