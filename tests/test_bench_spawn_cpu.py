"""`python bench.py --gpus N` outside torchrun starts its N ranks itself (bench.py, spawn_ranks): the parent never touches
a GPU, relays rank 0's single JSON line and exits with the children's code.  Exercised here without a GPU: with
BENCH_SPAWN_SELFTEST the ranks only rendezvous over gloo (127.0.0.1) and sum their ranks."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(mode, n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_SPAWN_SELFTEST"] = mode
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"],
                          env=env, capture_output=True, text=True, timeout=600)


def test_two_ranks_are_spawned_and_rank0_line_is_relayed():
    p = _run("1", 2)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line == {"selftest": True, "n_gpus": 2, "rank_sum": 1.0}


def test_a_failing_rank_makes_the_parent_fail():
    p = _run("fail", 2)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")]
