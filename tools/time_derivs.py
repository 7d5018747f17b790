"""Time the derivative entry points of config 5 (d ydd / d tau, d qd, d q) on one model.
usage: python tools/time_derivs.py [model] [B]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
if os.environ.get("GRBDA_LIB"):
    G.LIB_PATH = os.path.abspath(os.environ["GRBDA_LIB"])
from generalized_rbda_amd.states import random_states
model = sys.argv[1] if len(sys.argv) > 1 else "jvrc1_humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
q, qd, tau = random_states(plan.blob, B, 2)
for dt in (torch.float32, torch.float64):
    t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    def timed(fn, n=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    res = {"aba": timed(lambda: plan.forward_dynamics(tq, tqd, tt)), "rnea": timed(lambda: plan.inverse_dynamics(tq, tqd, tt)),
           "mass_matrix": timed(lambda: plan.mass_matrix(tq)), "fd_dtau": timed(lambda: plan.fd_dtau(tq)),
           "fd_dqd": timed(lambda: plan.fd_dqd(tq, tqd, tt)), "fd_dq": timed(lambda: plan.fd_dq(tq, tqd, tt)),
           "all three": timed(lambda: plan.fd_derivatives(tq, tqd, tt))}
    print(model, B, "nv", plan.nv, str(dt).split(".")[1], "  ".join(f"{k}={v:.3f}ms" for k, v in res.items()), flush=True)
