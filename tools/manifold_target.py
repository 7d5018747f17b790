"""A few calls of fd_derivatives on an implicit model (rocprofv3 target).  usage: manifold_target.py model B f32|f64"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import valid_random_states_device
name, B, prec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
if name == "tello":
    from generalized_rbda_amd.robots import tello_with_arms
    plan = G.Plan.from_model(tello_with_arms())
else:
    plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", name + ".urdf"))
q, qd, tau, _ = valid_random_states_device(plan, B, 5, "cuda:0")
dt = torch.float32 if prec == "f32" else torch.float64
t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
tq, tqd, tt = t(q), t(qd), t(tau)
for _ in range(4):
    out = plan.fd_derivatives(tq, tqd, tt)
torch.cuda.synchronize()
