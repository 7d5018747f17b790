"""Analytic derivatives on the constraint manifold (manifold_kernels.hip) against the difference batches (GRBDA_NO_MANIFOLD=1)
and timing of both.  usage: manifold_check.py [B]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import valid_random_states_device
from generalized_rbda_amd.robots import tello_with_arms
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
models = {"four_bar": None, "six_bar": None, "planar_leg_linkage": None, "tello_with_arms": tello_with_arms}
for name, build in models.items():
    res = {}
    for mode in ("manifold", "differences"):
        if mode == "differences": os.environ["GRBDA_NO_MANIFOLD"] = "1"
        else: os.environ.pop("GRBDA_NO_MANIFOLD", None)
        plan = G.Plan.from_model(build()) if build else G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", name + ".urdf"))
        q, qd, tau, _ = valid_random_states_device(plan, B, 5, "cuda:0")
        for dt in (torch.float64, torch.float32):
            t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
            tq, tqd, tt = t(q), t(qd), t(tau)
            out = plan.fd_derivatives(tq, tqd, tt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): out = plan.fd_derivatives(tq, tqd, tt)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            H = plan.mass_matrix(tq)
            res[(mode, dt)] = [out[k].double().cpu().numpy() for k in ("dq", "dqd", "dtau")] + [H.double().cpu().numpy()]
            print(f"{name:20s} {mode:12s} {str(dt):14s} analytic={plan.info().analytic_derivatives} B={B} {ms:9.3f} ms")
    for dt in (torch.float64, torch.float32):
        a, b = res[("manifold", dt)], res[("differences", dt)]
        errs = [np.abs(x - y).max() / (1 + np.abs(y).max()) for x, y in zip(a, b)]
        print(f"{name:20s} {str(dt):14s} max rel diff dq {errs[0]:.2e} dqd {errs[1]:.2e} dtau {errs[2]:.2e} H {errs[3]:.2e}")
