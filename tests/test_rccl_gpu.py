"""The library's own RCCL communicator (exp_amd_comm_get_unique_id / exp_amd_comm_init_rank,
exp_amd/csrc/context.hip): the ONE on-stream ncclAllReduce of the coefficient buffer that replaces the
(L+1)^2 MPI_Allreduce calls of src/SphericalBasis.cc:864-903 and the two of
exputil/EmpCylSL.cc:4188-4222.  RCCL accepts a one-rank communicator, so the path -- symbol binding,
communicator, the collective on the compute stream inside accumulate / fused step / the multistep
driver -- runs on the single GPU of the test box; the sum over one rank must leave every result
bit-identical to the run without a communicator.  GPU only."""
import numpy as np
import pytest

from tests.conftest import make_grid

pytestmark = pytest.mark.gpu


def _run(with_rccl):
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Context, Simulation, SphereSL
    ctx = Context(0)
    if with_rccl:
        ctx.init_rccl(Context.rccl_unique_id(), 1, 0)
        info = ctx.comm_info()
        assert info["kind"] == "rccl" and info["nranks"] == 1 and info["rank"] == 0
    else:
        assert ctx.comm_info()["kind"] == "none"
    model, g = make_grid("plummer", 4, 8, 400)
    m, pos, vel = sample_sphere(model, 30000, seed=5)
    out = {}
    f = SphereSL(ctx, g)
    c = Component.from_arrays(ctx, m, pos, vel)
    f.determine_coefficients(c)
    out["coef"] = f.get_coefs().copy()
    c.zero_acceleration(0)
    f.get_acceleration_and_potential(c)
    for _ in range(3):
        f.step_kdk(c, 0.01)
    out["step"] = c.download()
    n_single = ctx.comm_info()["allreduce_calls"]
    c.close(); f.close()
    # block multistep: the per-sub-step level block and the level-change differences are reduced too
    f = SphereSL(ctx, g, multistep=2)
    c = Component.from_arrays(ctx, m, pos, vel)
    sim = Simulation(ctx, 0.05, multistep=2)
    sim.add_component(c, f)
    sim.init()
    sim.step(2)
    out["ms"] = c.download()
    out["lev"] = c.download_levels()
    out["calls"] = (n_single, ctx.comm_info()["allreduce_calls"])
    sim.close(); c.close(); f.close(); ctx.close()
    return out


def test_native_rccl_communicator_one_rank():
    ref = _run(False)
    got = _run(True)
    assert ref["calls"] == (0, 0)
    # accumulate + 3 fused steps = 4 reductions; the multistep run adds one per sub-step and one per
    # level sweep with changes
    assert got["calls"][0] == 4 and got["calls"][1] > got["calls"][0] + 2 * 4
    assert np.array_equal(got["coef"], ref["coef"]) or \
        np.abs(got["coef"] - ref["coef"]).max() <= 1e-13 * np.abs(ref["coef"]).max()   # (atomics: not bitwise)
    for key in ("step", "ms"):
        for k in ("pos", "vel", "acc", "pot"):
            assert np.abs(got[key][k] - ref[key][k]).max() <= 1e-10 * np.abs(ref[key][k]).max(), (key, k)
    assert (got["lev"] != ref["lev"]).mean() < 1e-3 and ref["lev"].max() > 0


@pytest.mark.parametrize("prekick", [True, False])
@pytest.mark.parametrize("with_rccl", [False, True])
def test_graph_replay_of_fused_steps_is_bit_identical(with_rccl, prekick):
    """exp_amd_step_kdk_n replays PAIRS of steady-state fused steps from a HIP graph captured on the context's
    stream -- with a communicator, the ncclAllReduce of the coefficient buffer is a node of that graph.  In
    deterministic mode (order-independent sums) the replayed run must equal the eager one bit for bit: odd and
    even step counts, a diagnostic in the middle, a change of dt (the graph is dropped and captured again),
    and the all-reduce count must be what eager stepping gives.  Without the pre-kicked store the diagnostic completes the
    deferred closing half-kick, so the pair captured after it is not periodic: it is run once and the next pair captured
    afresh (this used to be refused with an error; found by tests/fuzz/fuzz_kdk.py)."""
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Context, SphereSL
    model, g = make_grid("plummer", 4, 8, 400)
    m, pos, vel = sample_sphere(model, 40000, seed=15)

    def run(graph):
        ctx = Context(0)
        if with_rccl:
            ctx.init_rccl(Context.rccl_unique_id(), 1, 0)
        ctx.set_deterministic(True)
        ctx.set_prekick(prekick)
        f = SphereSL(ctx, g)
        c = Component.from_arrays(ctx, m, pos, vel)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)

        def steps(k, dt):
            if graph:
                f.step_kdk_n(c, dt, k)
            else:
                for _ in range(k):
                    f.step_kdk(c, dt)

        steps(7, 0.01)
        mid = c.fix_positions()                    # a read-only diagnostic: the keys of the next step stay valid
        steps(6, 0.01)
        steps(5, 0.004)                            # another dt: captured anew
        steps(1, 0.004)
        out = c.download()
        out["coef"] = f.get_coefs().copy()
        out["calls"] = ctx.comm_info()["allreduce_calls"]
        out["mid"] = mid["com"]
        c.close(); f.close(); ctx.close()
        return out

    eager, replay = run(False), run(True)
    for k in ("pos", "vel", "acc", "pot", "coef"):
        assert np.array_equal(eager[k], replay[k]), k
    assert eager["calls"] == replay["calls"] == (20 if with_rccl else 0)


def test_graph_replay_without_deterministic_mode():
    """... and in the default mode (atomics in arrival order) to rounding; EXP_AMD_STEP_GRAPH is honoured."""
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Context, SphereSL
    model, g = make_grid("plummer", 6, 10, 400)
    m, pos, vel = sample_sphere(model, 200000, seed=16)
    outs = []
    for graph in (False, True):
        ctx = Context(0)
        f = SphereSL(ctx, g)
        c = Component.from_arrays(ctx, m, pos, vel)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
        if graph:
            f.step_kdk_n(c, 0.005, 12)
        else:
            for _ in range(12):
                f.step_kdk(c, 0.005)
        outs.append(c.download())
        c.close(); f.close(); ctx.close()
    for k in ("pos", "vel"):
        assert np.abs(outs[0][k] - outs[1][k]).max() <= 1e-11 * np.abs(outs[0][k]).max(), k
    assert np.abs(outs[0]["acc"] - outs[1]["acc"]).max() <= 1e-9 * np.abs(outs[0]["acc"]).max()


def test_graph_replay_sees_changes_made_between_calls():
    """A captured pair of fused steps bakes in kernel ARGUMENTS the replay key of round 3 did not cover: the force's
    settings (exterior continuation, dsmall, M0 accumulation), the component's frame (body rotation, pseudo-acceleration,
    the mass array a re-upload replaces) and the scratch pointers an outside force pass on a larger target re-allocates.
    Every such call now bumps the library's mutation counter (common.h) and the graph is captured afresh; replayed and
    eager runs must stay bit-identical through each of them, with 0 and 2 eager steps in between (an even number of
    eager steps used to bring `cur` and the parity back to the captured values)."""
    from exp_amd.models import sample_sphere
    from exp_amd.runtime import Component, Context, SphereSL
    from exp_amd._lib import check
    model, g = make_grid("plummer", 4, 8, 400)
    m, pos, vel = sample_sphere(model, 30000, seed=21)
    pos[:40] *= 80.0 / np.linalg.norm(pos[:40], axis=1)[:, None]      # beyond rmax: the exterior continuation matters
    m2, pos2, _ = sample_sphere(model, 90000, seed=22)
    th = 0.3
    body = np.array([[np.cos(th), np.sin(th), 0.0], [-np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])

    def run(graph):
        ctx = Context(0)
        ctx.set_deterministic(True)
        f = SphereSL(ctx, g)
        c = Component.from_arrays(ctx, m, pos, vel)
        big = Component.from_arrays(ctx, m2, pos2)
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)

        def steps(k, dt=0.01, eager=False):
            if graph and not eager:
                f.step_kdk_n(c, dt, k)
            else:
                for _ in range(k):
                    f.step_kdk(c, dt)

        steps(6)
        check(ctx.lib.exp_amd_sph_set_exterior(f.h, 0), ctx.h)          # case 1: a force setting, no eager step between
        steps(4)
        check(ctx.lib.exp_amd_sph_set_dsmall(f.h, 1e-18), ctx.h)
        steps(4)
        c.set_pseudo_accel([1e-3, -2e-3, 5e-4])                         # case 2: the frame, two eager steps between
        steps(2, eager=True)
        steps(4)
        c.set_orientation(body)
        steps(2, eager=True)
        steps(4)
        st = c.download(("mass", "pos", "vel"))                         # ... a re-upload with other masses
        c.upload(st["mass"] * 1.25, st["pos"], st["vel"])
        f.determine_coefficients(c); c.zero_acceleration(0); f.get_acceleration_and_potential(c)
        steps(2, eager=True)
        steps(4)
        f.get_acceleration_and_potential(big, external=True)            # case 3: scratch of the force pass re-allocated
        steps(2, eager=True)
        steps(5)
        out = c.download()
        out["coef"] = f.get_coefs().copy()
        big.close(); c.close(); f.close(); ctx.close()
        return out

    eager, replay = run(False), run(True)
    for k in ("pos", "vel", "acc", "pot", "coef"):
        assert np.array_equal(eager[k], replay[k]), k


def _two_component_run(with_rccl):
    """the disk + halo miniature through the step driver (two streams)"""
    from exp_amd.runtime import Component, Context, Cylinder, Simulation, SphereSL
    from tests import config4_util as c4
    ctx = Context(0)
    if with_rccl:
        ctx.init_rccl(Context.rccl_unique_id(), 1, 0)
        assert ctx.comm_info()["streams"] == 2          # ncclCommSplit gave the auxiliary stream its own communicator
    inp = c4.config4_inputs(n_halo=500, n_disk=500)
    g, cg = c4.grids()
    fh = SphereSL(ctx, g, multistep=3, **c4.sph_window(g, float(inp["scale"])))
    fd = Cylinder(ctx, cg, multistep=3)
    ch = Component.from_arrays(ctx, inp["halo_mass"], inp["halo_pos"], inp["halo_vel"])
    cd = Component.from_arrays(ctx, inp["disk_mass"], inp["disk_pos"], inp["disk_vel"])
    sim = Simulation(ctx, 2.5e-4, multistep=3, dynfrac=c4.DYN)
    k1, k2 = sim.add_component(ch, fh), sim.add_component(cd, fd)
    sim.add_interaction(k1, k2)
    sim.add_interaction(k2, k1)
    sim.init()
    sim.step(2)
    out = dict(h=ch.download(), d=cd.download(), lh=ch.download_levels(), ld=cd.download_levels(),
               calls=ctx.comm_info()["allreduce_calls"], switches=sim.step_switches)
    sim.close()
    for o in (ch, cd, fh, fd):
        o.close()
    ctx.close()
    return out


def test_two_stream_step_driver_with_a_communicator_per_stream():
    """With an RCCL communicator the two-component step driver keeps its two-stream schedule: the auxiliary stream's
    collectives go through a second communicator split from the first (exp_amd/csrc/context.hip:
    expamd_comm_two_streams).  One rank: the sums over one rank leave every result what it is without a communicator, to
    the order of the atomics."""
    ref, got = _two_component_run(False), _two_component_run(True)
    assert ref["calls"] == 0 and got["calls"] > 2 * 2 * 8 and ref["switches"] > 0
    assert np.array_equal(ref["lh"], got["lh"]) and np.array_equal(ref["ld"], got["ld"])
    for comp in ("h", "d"):
        assert np.abs(ref[comp]["pos"] - got[comp]["pos"]).max() <= 1e-12
        for k in ("vel", "acc", "pot"):
            assert np.abs(ref[comp][k] - got[comp][k]).max() <= 1e-10 * np.abs(ref[comp][k]).max(), (comp, k)
