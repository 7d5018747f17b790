import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import valid_random_states_device
for model in ("four_bar", "six_bar"):
    plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
    B = 1048576
    q, qd, tau, nd = valid_random_states_device(plan, B, 5, "cuda:0")
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    def timed(fn, n=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    print(model, "nv", plan.nv, "aba %.3f ms" % timed(lambda: plan.forward_dynamics(tq, tqd, tt)), "all three %.3f ms" % timed(lambda: plan.fd_derivatives(tq, tqd, tt)), flush=True)
