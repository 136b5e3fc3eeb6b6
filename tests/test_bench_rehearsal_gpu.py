"""Dress rehearsal of the command the driver runs on a multi-GPU node,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

on the ONE GPU of the test box: `--rehearse-shared-gpu` puts every rank on device 0 with a gloo process group and a
host-staged coefficient all-reduce (RCCL refuses two ranks on one device); the sharding of the particle set, the
communicator vote and its fallback, the MAX-reduce of the timed regions and the rank-0-only JSON line are bench.py's
production code (bench.py main()).  What is replaced: the MPI_Allreduce of src/SphericalBasis.cc:864-903.

Every bench run is a CHILD process (spawned, never exec'ed from a process that has touched the GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--nbodies", "4e5", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs",
         "--no-sustained"]


def _env(extra=None):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("EXP_AMD_BENCH_FAIL_RCCL", None)
    if extra:
        env.update(extra)
    return env


def _json_lines(out):
    lines = []
    for ln in out.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                lines.append(json.loads(ln))
            except ValueError:
                pass
    return lines


def _bench(args, world=1, port=None, env=None, timeout=900):
    if world > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
               "--gpus", str(world)] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args
    p = subprocess.run(cmd, cwd=ROOT, env=_env(env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, ("exactly one JSON line on stdout (rank 0 only)", p.stdout[-3000:])
    return lines[0], p.stderr


@pytest.fixture(scope="module")
def one_rank():
    line, _ = _bench(SMALL)
    assert line["n_gpus"] == 1 and line["config"]["comm"]["kind"] == "none"
    return line


def _check_two(line, one, graph):
    assert line["metric"] == one["metric"] and line["unit"] == "particle-steps/s"
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    cfg = line["config"]
    assert cfg["nbodies_total"] == 400000 and cfg["nbodies_per_gpu"] == 200000
    assert cfg["parallelism"].startswith("particle-shard x2")
    comm = cfg["comm"]
    assert comm["nranks"] == 2 and comm["rank"] == 0 and comm["kind"] == "callback"
    assert comm["allreduce_calls"] > 0
    assert "fallback" in comm["path"] and "voted down" in comm["note"]        # bench.py's vote + fallback ran
    assert cfg["rehearsal"]
    # the MAX over the ranks of the wall-clock region is what `value` is made of
    assert line["value"] == pytest.approx(400000 / (line["ms_per_step"] * 1e-3), rel=1e-9)
    assert line["scaling"] == "strong" and line["roofline"]["kernel"]
    sc, sc1 = line["selfcheck"], one["selfcheck"]
    assert sc["particles_rank0"] == 200000 and 0 < sc["used_rank0"] <= 200000
    # the 2-rank run IS the 1-rank run sharded: same particle set, same number of steps; the sums differ in order only
    assert sc["mtot"] == pytest.approx(sc1["mtot"], rel=1e-12)
    if not graph:
        assert sc["coef_00_0"] == pytest.approx(sc1["coef_00_0"], rel=1e-12)
        for a, b in zip(sc["center_of_mass"], sc1["center_of_mass"]):
            assert abs(a - b) <= 1e-12
    else:
        # --graph runs 4 + K more steps before the eager region: a different point of the same run
        assert sc["coef_00_0"] == pytest.approx(sc1["coef_00_0"], rel=1e-3)
        assert "exp_amd_step_kdk_n" in line["stepping"]


def test_two_ranks_through_torchrun(one_rank):
    port = 29600 + (os.getpid() % 300)
    line, err = _bench(SMALL + ["--rehearse-shared-gpu"], world=2, port=port)
    _check_two(line, one_rank, graph=False)
    assert "falling back" in err


def test_two_ranks_through_torchrun_graph(one_rank):
    port = 29950 + (os.getpid() % 300)
    line, _ = _bench(SMALL + ["--rehearse-shared-gpu", "--graph"], world=2, port=port)
    _check_two(line, one_rank, graph=True)


def test_two_ranks_the_driver_s_spelling_with_every_region():
    """The scaling command as the driver spells it -- `--gpus N --steps K --warmup W` and nothing else that switches a region
    off -- on two ranks with shards above 2^20 particles: the append form of the fused step over the all-reduce callback, the
    `sustained` region (whose step count must be the SAME on every rank: each step carries an all-reduce -- formed from a
    rank's own clock it differed between the ranks and the run never came back) and the lean A/B region."""
    port = 31000 + (os.getpid() % 300)
    line, _ = _bench(["--nbodies", "2.4e6", "--steps", "3", "--warmup", "4", "--no-cpu-baseline", "--no-other-configs",
                      "--rehearse-shared-gpu"], world=2, port=port, timeout=600)
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["nbodies_per_gpu"] == 1200000 and cfg["comm"]["nranks"] == 2
    assert cfg["step_form"] == "append" and cfg["append_payload"] == "full"
    assert line["sustained"]["steps"] >= 3 and line["sustained"]["seconds"] > 0.5
    assert line["append_lean_ab"]["steps"] == 3 and line["append_lean_ab"]["ms_per_step"] > 0
    assert line["selfcheck"]["mtot"] == pytest.approx(1.0, rel=1e-12)


def test_weak_scaling_flag_two_ranks():
    port = 30300 + (os.getpid() % 300)
    line, _ = _bench(["--nbodies", "2e5", "--scaling", "weak", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                      "--no-other-configs", "--no-sustained", "--rehearse-shared-gpu"], world=2, port=port)
    assert line["scaling"] == "weak" and line["config"]["nbodies_total"] == 400000
    assert line["config"]["nbodies_per_gpu"] == 200000 and line["config"]["comm"]["nranks"] == 2


def test_native_rccl_failure_takes_the_fallback(one_rank):
    """One rank, real RCCL (nranks 1) set up and probed, then voted down by the test hook: the torch.distributed callback
    must carry the run, and the result must be the plain single-rank one."""
    line, err = _bench(SMALL + ["--force-comm"], env={"EXP_AMD_BENCH_FAIL_RCCL": "all", "MASTER_PORT": str(
        30650 + (os.getpid() % 300))})
    comm = line["config"]["comm"]
    assert "fallback" in comm["path"] and "EXP_AMD_BENCH_FAIL_RCCL" in comm["note"]
    assert comm["kind"] == "callback" and comm["allreduce_calls"] > 0 and comm["nranks"] == 1
    assert "falling back" in err
    assert line["selfcheck"]["coef_00_0"] == pytest.approx(one_rank["selfcheck"]["coef_00_0"], rel=1e-12)


def test_native_rccl_one_rank(one_rank):
    """... and without the hook the library's communicator is the one used."""
    line, _ = _bench(SMALL + ["--force-comm"], env={"MASTER_PORT": str(31000 + (os.getpid() % 300))})
    comm = line["config"]["comm"]
    assert comm["kind"] == "rccl" and comm["allreduce_calls"] > 0 and comm["note"] is None
    assert line["selfcheck"]["coef_00_0"] == pytest.approx(one_rank["selfcheck"]["coef_00_0"], rel=1e-12)


def test_two_ranks_with_no_launcher(one_rank):
    """`python bench.py --gpus 2` as the driver spells the N = 1 command -- no torchrun around it: bench.py starts its own
    ranks (bench.py: self_launch) instead of measuring one GPU and printing n_gpus 1."""
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu"] + SMALL
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout[-3000:]
    _check_two(lines[0], one_rank, graph=False)


def test_split_step_flag_is_the_same_run(one_rank):
    """`bench.py --split` (the opt-in two-half-shard fused step, sort passes on a second stream: profiles/r06_overlap_ab.txt)
    is the same run: the coefficient set differs from the plain step's only in the order of its sums."""
    line, _ = _bench(SMALL + ["--split"])
    assert "SPLIT" in line["stepping"] and line["n_gpus"] == 1
    sc, sc1 = line["selfcheck"], one_rank["selfcheck"]
    assert sc["used_rank0"] == sc1["used_rank0"]
    assert sc["coef_00_0"] == pytest.approx(sc1["coef_00_0"], rel=1e-12)
    for a, b in zip(sc["center_of_mass"], sc1["center_of_mass"]):
        assert abs(a - b) <= 1e-12
    assert "k_scatter_adv" in line["roofline"]["kernels_ms_per_step"]
