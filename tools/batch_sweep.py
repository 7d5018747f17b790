"""Kernel time of the forward dynamics over a sweep of batch sizes (startup cost vs steady-state rounds; latency mode below
one tile per SIMD: GRBDA_NO_LATENCY_MODE=1 switches it off).   usage: python tools/batch_sweep.py [model] [f32|f64]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
if os.environ.get("GRBDA_LIB"):
    G.LIB_PATH = os.path.abspath(os.environ["GRBDA_LIB"])
from generalized_rbda_amd.states import random_states
model = sys.argv[1] if len(sys.argv) > 1 else "mit_humanoid"
dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
for B in (64, 4096, 16384, 32768, 49152, 65536, 65600, 98304, 131072, 196608, 262144, 393216, 524288, 1048576, 2097152):
    q, qd, tau = random_states(plan.blob, B, 2)
    t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    out = torch.empty((B, plan.nv), dtype=dt, device="cuda:0")
    plan.time_kernel("aba", tq, tqd, tt, out, iters=5)
    ms = min(plan.time_kernel("aba", tq, tqd, tt, out, iters=30) for _ in range(3))
    tiles = (B + 63) // 64
    print(f"B {B:8d} tiles {tiles:6d} rounds {tiles / 2048:6.2f}  {ms:.4f} ms  {B / ms / 1e6:8.3f} G evals/s", flush=True)
