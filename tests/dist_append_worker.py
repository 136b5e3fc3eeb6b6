"""Worker of tests/test_dist_gpu.py::test_append_step_over_two_ranks: one rank of a run that SHARES one GPU (host-staged gloo
all-reduce, as tests/dist_worker.py), stepping its block of the particles with exp_amd_step_kdk in a chosen form of the fused
step: "off" (ordinary), "app" (append, every rank), "mixed" (rank 0: regions WITHOUT slack -- every pass runs out of room and
the force pass is redone from its source --, the other ranks: append with slack; lean payload on the odd ranks)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch.distributed as dist
    from exp_amd.dist import host_staged_allreduce_callback, shard_range
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Context, SphereSL
    from tests.conftest import make_grid
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    model, g = make_grid("plummer", 4, 8, 400)
    n = 60001
    m, pos, vel = sample_sphere(model, n, seed=67)
    pos[:, 2] *= 0.8
    n0, n1 = shard_range(n, rank, world)
    ctx = Context(0)
    if world > 1:
        ctx.set_allreduce(host_staged_allreduce_callback(), world, rank)
    ctx.set_append_min({"off": 0, "app": 1000, "mixed": -1000 if rank == 0 else 1000}[mode])
    ctx.set_append_lean(mode == "mixed" and rank % 2 == 1)
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m[n0:n1], pos[n0:n1], vel[n0:n1])
    f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
    nosort = []
    for k in range(9):
        ctx.profile(True); ctx.profile_reset()
        f.step_kdk(c, 0.01)
        nosort.append(not ctx.profile_report().get("k_scatter_adv", {}).get("launches", 0))
        ctx.profile(False)
    d = c.download(("pos", "vel", "acc", "pot"))
    info = ctx.comm_info()
    np.savez(out, coef=f.get_coefs(), used=f.Used(), n0=n0, n1=n1, nosort=np.array(nosort), calls=info["allreduce_calls"], **d)
    c.close(); f.close(); ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
