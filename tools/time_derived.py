"""Time the derived entry points (mass matrix, H^-1, inverse OSIM) on one model.  usage: python tools/time_derived.py [model] [B]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import random_states
model = sys.argv[1] if len(sys.argv) > 1 else "mit_humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
q, qd, tau = random_states(plan.blob, B, 2)
for dt in (torch.float32, torch.float64):
    t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    def timed(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    res = {"mass_matrix": timed(lambda: plan.mass_matrix(tq)), "fd_dtau": timed(lambda: plan.fd_dtau(tq)),
           "aba": timed(lambda: plan.forward_dynamics(tq, tqd, tt)), "rnea": timed(lambda: plan.inverse_dynamics(tq, tqd, tt)),
           "inv_osim(2)": timed(lambda: plan.inv_osim(tq, [plan.n_bodies - 1, plan.n_bodies // 2], [[0.05, -0.02, 0.1], [0, 0.03, -0.2]]), 2)}
    print(model, B, str(dt).split(".")[1], "  ".join(f"{k}={v:.3f}ms" for k, v in res.items()), f"env NO_CRBA={os.environ.get('GRBDA_NO_CRBA', '0')}", flush=True)
