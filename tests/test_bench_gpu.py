"""bench.py end to end on the GPU box: the single-rank line, and the multi-rank code path (RCCL process group,
blocking gather, overlapped gathers) forced onto ONE rank with BENCH_FORCE_DIST=1 so that the driver exercises it
even where only one GPU is leased."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args):
    env = dict(os.environ)
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"bench.py must print ONE line on stdout, got {len(lines)}"
    return json.loads(lines[0])


def test_bench_line_is_verified_and_carries_roofline_and_cpu_baseline(gpu):
    line = _run({}, "--steps", "10", "--warmup", "2")
    assert line["verified"] is True and line["verify_states"] >= 1000
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["dtype"] == "f32"
    r = line["roofline"]
    # `bound` names the resource achieved / peak / frac are figures of; what the counters say binds the kernel is a key of its own
    assert r["bound"] == "hbm" and r["binding_measured"] == "valu_issue" and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["kernel_ms"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["single_thread"] > 0 and c["passes"] >= 5
    # every other BASELINE config rides in the same line as a compact sub-record
    recs = {(x["workload"], x["algo"], x["dtype"]): x for x in line["configs"] if not x.get("split")}
    for key in [("revolute_rotor_chain", "aba", "f64"), ("mini_cheetah", "aba", "f64"), ("mit_humanoid", "rnea", "f32"), ("tello", "aba", "f32"),
                ("tello", "rnea", "f32"), ("jvrc1_humanoid", "aba", "f32"), ("jvrc1_humanoid", "fd_derivatives", "f32"),
                ("jvrc1_humanoid", "fd_derivatives", "f64"), ("mit_humanoid", "fd_derivatives", "f32"), ("four_bar", "aba", "f32"),
                ("six_bar", "aba", "f32")]:
        x = recs[key]
        assert "error" not in x, x
        assert x["verified"] is True and x["ms"] > 0 and x["evals_per_s"] > 0 and 0 < x["roofline"]["frac"] < 1, x
        assert x["kernel"]
    # one GPU's share of the 8-way strong split of configs 3 and 4, with the efficiency that split would have
    shares = {x["workload"]: x for x in line["configs"] if x.get("split") == 8 and x["algo"] == "aba"}
    assert set(shares) == {"mit_humanoid", "tello"}
    rnea_share = [x for x in line["configs"] if x.get("split") == 8 and x["algo"] == "rnea"]
    assert len(rnea_share) == 1 and rnea_share[0]["verified"] is True and "rnea_chain_lm_kernel<float, 4>" in rnea_share[0]["kernel"]
    for x in shares.values():
        assert "error" not in x, x
        assert x["verified"] is True and 0.2 < x["split_efficiency"] < 1.2 and x["batch"] * 8 in (262144, 1048576), x
    assert "lm_kernel<float, 4" in shares["mit_humanoid"]["kernel"]


def test_bench_multi_rank_path_on_one_rank(gpu):
    import socket

    with socket.socket() as sk:  # a port nobody holds right now (a fixed one can linger in TIME_WAIT between runs)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {"BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1",
           "LOCAL_RANK": "0"}
    line = _run(env, "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--workload", "tello", "--scaling", "strong",
                "--batch", "131072")
    assert line["verified"] is True
    assert line["gather_ms"] > 0 and line["end_to_end"]["value"] > 0
    # (the two figures come from separate timed loops of 6 steps each; either can catch a stall of the box, so they are
    # not compared -- that assertion failed once in 30 runs with the compute-only loop 5 x slower than usual)
    assert line["value"] > 0


def test_bench_weak_run_carries_the_strong_split_of_config_4(gpu):
    """N > 1 weak runs append a `strong` sub-record: BASELINE config 4's ONE batch of 1 048 576 TelloWithArms states split over the ranks.
    Here the multi-rank path is forced on one rank (the whole batch on it)."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {"BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1",
           "LOCAL_RANK": "0"}
    line = _run(env, "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--replays", "3")
    assert line["verified"] is True and line["scaling"] == "weak"
    st = line["strong"]
    assert st["scaling"] == "strong" and st["batch_global"] == 1048576 and st["batch_per_gpu"] == 1048576 and st["n_gpus"] == 1
    assert st["verified"] is True and st["value"] > 0 and st["ms_per_step"] > 0
