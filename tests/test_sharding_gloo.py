"""N > 1 path on CPU: world_size-2 gloo processes shard a batch, each computes its slab (the oracle
stands in for the HIP kernel, which cannot run here) and the gathered result must equal the
single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_py as O
from generalized_rbda_amd.sharding import shard_range, sharded_apply
from generalized_rbda_amd.states import random_states
from models import zoo


def test_shard_range_partitions_exactly():
    for B in (0, 1, 7, 64, 1000, 262144):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(B, r, world)
                assert 0 <= lo <= hi <= B
                cover += list(range(lo, hi)) if B <= 1000 else []
                sizes = [shard_range(B, rr, world)[1] - shard_range(B, rr, world)[0] for rr in range(world)]
                assert max(sizes) - min(sizes) <= 1 and sum(sizes) == B
            if B <= 1000:
                assert cover == list(range(B))


def _worker(rank, world, port, blob, q, qd, tau, ref, ok):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        compute = lambda a, b, c: torch.from_numpy(O.forward_dynamics(blob, a.numpy(), b.numpy(), c.numpy()))
        out = sharded_apply(compute, torch.from_numpy(q), torch.from_numpy(qd), torch.from_numpy(tau))
        ok[rank] = int(np.array_equal(out.numpy(), ref))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_forward_dynamics_matches_single_process():
    blob = zoo()["tree_rotor_float"]
    q, qd, tau = random_states(blob, 101, config_index=31)  # odd: slabs of 51 and 50 states
    ref = O.forward_dynamics(blob, q, qd, tau)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ok = mp.get_context("spawn").Array("i", [0, 0])
    mp.spawn(_worker, args=(2, port, blob, q, qd, tau, ref, ok), nprocs=2, join=True)
    assert list(ok) == [1, 1]
