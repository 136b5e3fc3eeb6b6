"""Worker of tests/test_dist_gpu.py: one rank of a 2-process run that SHARES one GPU.  The data-path
collective (one all-reduce of a small device buffer) goes host-staged through gloo, so that two
ranks can sit on the same device (RCCL refuses that); everything else is the production path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch
    import torch.distributed as dist
    from exp_amd.dist import host_staged_allreduce_callback, shard_range
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Context, Orient, SphereSL
    from tests.conftest import make_grid
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    model, g = make_grid("plummer", 4, 8, 400)
    n = 40001
    m, pos, vel = sample_sphere(model, n, seed=61)
    pos = pos + np.array([0.1, 0.0, -0.05])
    n0, n1 = shard_range(n, rank, world)
    ctx = Context(0)
    if world > 1:
        ctx.set_allreduce(host_staged_allreduce_callback(), world, rank)
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m[n0:n1], pos[n0:n1], vel[n0:n1])
    f.determine_coefficients(c)
    coef0 = f.get_coefs()
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    o = Orient(ctx, 1, 3000, Orient.CENTER | Orient.AXIS, Orient.KE)
    o.accumulate(0.0, c)                      # global radix select: histograms all-reduced per pass
    st = o.state()
    for _ in range(3):
        f.step_kdk(c, 0.01)
    com = c.fix_positions(0)
    d = c.download(("pos", "vel", "acc", "pot"))
    np.savez(out, coef0=coef0, coef=f.get_coefs(), used=f.Used(), n0=n0, n1=n1, Ecurr=st["Ecurr"],
             oused=st["used"], center1=st["center1"], axis1=st["axis1"], com=com["com"], mtot=com["mtot"],
             **d)
    o.close(); c.close(); f.close()
    # block multistep over the ranks: per-level coefficient sets, level-change differencing (its
    # packed all-reduce runs on every rank whether or not it has movers), re-sorts
    from exp_amd.runtime import Simulation
    ms, dtime = 2, 0.04
    f = SphereSL(ctx, g, multistep=ms)
    c = Component.from_arrays(ctx, m[n0:n1], pos[n0:n1], vel[n0:n1])
    sim = Simulation(ctx, dtime, multistep=ms, dynfrac=[1000.0, 0.01, 0.01, 0.03, 0.05], shiftlevl=0)
    sim.add_component(c, f)
    sim.init()
    sim.step(2)
    d2 = c.download(("pos", "vel", "acc"))
    lev = c.download_levels()
    np.savez(out.replace(".npz", "_ms.npz"), coef=f.get_coefs(), lev=lev, **d2)
    sim.close(); c.close(); f.close()
    # the same run on a RADIUS-ORDERED input (what the reference's gensph writes) with the level-balanced strided
    # partition (exp_amd.dist.shard_indices): every rank must hold ~1/world of every time-step level
    from exp_amd.dist import shard_indices
    order = np.argsort(np.linalg.norm(pos, axis=1), kind="stable")
    mr, pr, vr = m[order], pos[order], vel[order]
    idx = shard_indices(n, rank, world)
    ms, dtime = 3, 0.04
    f = SphereSL(ctx, g, multistep=ms)
    c = Component.from_arrays(ctx, mr[idx], pr[idx], vr[idx])
    sim = Simulation(ctx, dtime, multistep=ms, dynfrac=[1000.0, 0.01, 0.01, 0.03, 0.05], shiftlevl=0)
    sim.add_component(c, f)
    sim.init()
    lev0 = c.download_levels()
    sim.step(1)
    d3 = c.download(("pos", "vel"))
    np.savez(out.replace(".npz", "_bal.npz"), idx=idx, lev0=lev0, lev=c.download_levels(), coef=f.get_coefs(), **d3)
    sim.close(); c.close(); f.close()
    # ---- the two-component run (BASELINE config 4 in miniature: sphereSL halo + cylinder disk, both self forces and both
    # cross forces, block multistep) with BOTH components sharded over the ranks by shard_indices: every rank holds ~1/world of
    # every level of each; the cylinder's cos / sin sets and its in-cut mass ride one all-reduce (exputil/EmpCylSL.cc:
    # 4355-4550 uses two, src/Cylinder.cc:1081-1098 two more).  With a callback the step driver keeps its two-stream
    # schedule (the callback is handed each stream); EXP_AMD_SIM_OVERLAP=0 in the environment runs the one-stream one.
    from exp_amd.runtime import Cylinder
    from tests import config4_util as c4
    inp = c4.config4_inputs(n_halo=600, n_disk=600)
    gs, cg = c4.grids()
    ms, dtime = 3, 2.5e-4
    ih, idk = shard_indices(600, rank, world), shard_indices(600, rank, world)
    fh = SphereSL(ctx, gs, multistep=ms, **c4.sph_window(gs, float(inp["scale"])))
    fd = Cylinder(ctx, cg, multistep=ms)
    ch = Component.from_arrays(ctx, inp["halo_mass"][ih], inp["halo_pos"][ih], inp["halo_vel"][ih])
    cd = Component.from_arrays(ctx, inp["disk_mass"][idk], inp["disk_pos"][idk], inp["disk_vel"][idk])
    sim = Simulation(ctx, dtime, multistep=ms, dynfrac=c4.DYN, shiftlevl=0)
    k1, k2 = sim.add_component(ch, fh), sim.add_component(cd, fd)
    sim.add_interaction(k1, k2)
    sim.add_interaction(k2, k1)
    sim.init()
    lev0 = (ch.download_levels(), cd.download_levels())
    sim.step(2)
    dh, dd = ch.download(("pos", "vel", "acc", "pot")), cd.download(("pos", "vel", "acc", "pot"))
    gc = fd.get_coefs()
    np.savez(out.replace(".npz", "_c4.npz"), ih=ih, idk=idk, lev0_h=lev0[0], lev0_d=lev0[1], lev_h=ch.download_levels(),
             lev_d=cd.download_levels(), coef_h=fh.get_coefs(), coef_dc=gc[0], coef_ds=gc[1], cylmass=fd.cylmass,
             used_h=fh.Used(), used_d=fd.Used(), switches=sim.step_switches, calls=ctx.comm_info()["allreduce_calls"],
             **{"h_" + k: v for k, v in dh.items()}, **{"d_" + k: v for k, v in dd.items()})
    sim.close()
    for o in (ch, cd, fh, fd):
        o.close()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
