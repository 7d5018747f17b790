"""A few launches of one kernel (target program of tools/pmc_run.sh / tools/profile_round.sh).
usage: pmc_target.py aba|rnea 32|64 [model]   env: PMC_BATCH, PMC_LAUNCHES"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import random_states
kind = sys.argv[1] if len(sys.argv) > 1 else "aba"
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 32
model = sys.argv[3] if len(sys.argv) > 3 else "mit_humanoid"
if model == "tello":
    from generalized_rbda_amd.robots import tello_with_arms
    plan = G.Plan.from_model(tello_with_arms())
else:
    plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
B = int(os.environ.get("PMC_BATCH", "262144"))
from generalized_rbda_amd.states import valid_random_states_device
q, qd, tau, _ = valid_random_states_device(plan, B, 2, "cuda:0")  # (implicit models: projected + gated, as bench.py)
dt = torch.float32 if prec == 32 else torch.float64
t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
tq, tqd, tt = t(q), t(qd), t(tau)
out = torch.empty((B, plan.nv), dtype=dt, device="cuda:0")
for _ in range(int(os.environ.get("PMC_LAUNCHES", "6"))):
    (plan.forward_dynamics if kind == "aba" else plan.inverse_dynamics)(tq, tqd, tt, out=out)
torch.cuda.synchronize()
