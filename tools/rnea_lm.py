"""Latency mode of the inverse dynamics: parity against the oracle at small batches and kernel ms, latency mode (the library's choice), two wavefronts per tile
(GRBDA_LM_WAVES=2) and the one-wavefront kernels (GRBDA_NO_LATENCY_MODE=1).   usage: python tools/rnea_lm.py [model ...]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import generalized_rbda_amd as G
import oracle_py as O
from generalized_rbda_amd.states import random_states

for model in (sys.argv[1:] or ["mit_humanoid", "mini_cheetah"]):
    path = os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf")
    plans = {}
    for label, env in (("lm", {}), ("two", {"GRBDA_LM_WAVES": "2"}), ("one", {"GRBDA_NO_LATENCY_MODE": "1"})):
        os.environ.update(env)
        plans[label] = G.Plan.from_urdf(path)
        for k in env:
            del os.environ[k]
    for dt in (torch.float32, torch.float64):
        dn = "f32" if dt == torch.float32 else "f64"
        for B in (1, 65, 1000):
            q, qd, ydd = random_states(plans["lm"].blob, B, 3)
            c = (lambda a: a.astype(np.float32).astype(np.float64)) if dt == torch.float32 else (lambda a: a)
            t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
            ref = O.inverse_dynamics(plans["lm"].blob, c(q), c(qd), c(ydd))
            errs = []
            for label, plan in plans.items():
                got = plan.inverse_dynamics(t(q), t(qd), t(ydd)).double().cpu().numpy()
                errs.append(f"{label} {np.abs(got - ref).max() / (1 + np.abs(ref).max()):.1e}")
            print(f"{model} {dn} B {B}: " + "  ".join(errs) + "   " + plans["lm"].kernel_name("rnea", dn, B).split("::")[-1], flush=True)
        for B in (64, 16384, 32768, 65536):
            q, qd, ydd = random_states(plans["lm"].blob, B, 2)
            t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
            tq, tqd, tx = t(q), t(qd), t(ydd)
            out = torch.empty((B, plans["lm"].nv), dtype=dt, device="cuda:0")
            row = []
            for label, plan in plans.items():
                plan.time_kernel("rnea", tq, tqd, tx, out, iters=5)
                ms = min(plan.time_kernel("rnea", tq, tqd, tx, out, iters=30) for _ in range(3))
                row.append(f"{label} {ms:.4f} ms ({plan.kernel_name('rnea', dn, B).split('::')[-1].replace('rnea_chain_', '')})")
            print(f"{model:14s} {dn} B {B:6d}  " + "  ".join(row), flush=True)
