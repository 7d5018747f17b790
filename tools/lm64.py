"""Latency mode of the fp64 forward dynamics: four against two wavefronts per tile (GRBDA_LM_WAVES=2) against the one-wavefront kernel (GRBDA_NO_LATENCY_MODE=1),
kernel ms at small batches.   usage: python tools/lm64.py [model ...]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import random_states

for model in (sys.argv[1:] or ["mini_cheetah", "mit_humanoid"]):
    path = os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf")
    plans = {}
    for label, env in (("four", {}), ("two", {"GRBDA_LM_WAVES": "2"}), ("one", {"GRBDA_NO_LATENCY_MODE": "1"})):
        os.environ.update(env)
        plans[label] = G.Plan.from_urdf(path)
        for k in env:
            del os.environ[k]
    for B in (64, 8192, 16384, 32768, 65536):
        q, qd, tau = random_states(plans["four"].blob, B, 2)
        t = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda:0")
        tq, tqd, tt = t(q), t(qd), t(tau)
        out = torch.empty((B, plans["four"].nv), dtype=torch.float64, device="cuda:0")
        row = []
        for label, plan in plans.items():
            plan.time_kernel("aba", tq, tqd, tt, out, iters=5)
            ms = min(plan.time_kernel("aba", tq, tqd, tt, out, iters=30) for _ in range(3))
            row.append(f"{label} {ms:.4f} ms ({plan.kernel_name('aba', 'f64', B).split('::')[-1].replace('aba_chain_', '')})")
        print(f"{model:14s} fp64 B {B:6d} tiles {(B + 63) // 64:4d}  " + "  ".join(row), flush=True)
