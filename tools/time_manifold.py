"""Time fd_derivatives on TelloWithArms (analytic route).  usage: time_manifold.py B [f32|f64]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import valid_random_states_device
from generalized_rbda_amd.robots import tello_with_arms
B = int(sys.argv[1]); precs = sys.argv[2:] or ["f32", "f64"]
plan = G.Plan.from_model(tello_with_arms())
q, qd, tau, _ = valid_random_states_device(plan, B, 5, "cuda:0")
for prec in precs:
    dt = torch.float32 if prec == "f32" else torch.float64
    t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    plan.fd_derivatives(tq, tqd, tt); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): out = plan.fd_derivatives(tq, tqd, tt)
    torch.cuda.synchronize()
    print(f"tello_with_arms B={B} {prec} fd_derivatives {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms  (GRBDA_WORK_WANT_MB={os.environ.get('GRBDA_WORK_WANT_MB', '-')})", flush=True)
