"""A few derivative passes on JVRC-1 (target program of the rocprofv3 counter passes over deriv_kernels.hip)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
if os.environ.get("GRBDA_LIB"):
    G.LIB_PATH = os.path.abspath(os.environ["GRBDA_LIB"])
from generalized_rbda_amd.states import random_states
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models/jvrc1_humanoid.urdf"))
B = int(os.environ.get("PMC_BATCH", "65536"))
q, qd, tau = random_states(plan.blob, B, 2)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
tq, tqd, tt = t(q), t(qd), t(tau)
for _ in range(3):
    plan.fd_derivatives(tq, tqd, tt)
torch.cuda.synchronize()
