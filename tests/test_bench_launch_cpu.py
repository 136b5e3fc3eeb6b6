"""bench.py's launch logic that needs no GPU: a world that contradicts --gpus is refused before anything is measured, and
the self-launcher (no WORLD_SIZE, --gpus N > 1) reports a failing rank with a non-zero exit code.  One rank <-> one GPU:
/root/reference/src/begin.cc:146-210."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_world_size_that_contradicts_gpus_is_refused():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr and "{" not in p.stdout
    p = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_self_launch_passes_a_failing_rank_on(tmp_path):
    """EXP_AMD_BENCH_LAUNCH_PROBE makes each child print its rank environment and exit with the given code instead of
    running the bench: the parent must start exactly N children with the torchrun layout and exit non-zero when one fails."""
    p = _run(["--gpus", "3"], {"EXP_AMD_BENCH_LAUNCH_PROBE": "0"})
    assert p.returncode == 0, p.stderr
    seen = sorted(ln for ln in p.stdout.splitlines() if ln.startswith("probe "))
    assert seen == [f"probe rank={r} local_rank={r} world=3 addr=127.0.0.1" for r in range(3)]
    p = _run(["--gpus", "2"], {"EXP_AMD_BENCH_LAUNCH_PROBE": "7:1"})       # rank 1 exits with 7
    assert p.returncode == 7 and "rank 1 exited with 7" in p.stderr
