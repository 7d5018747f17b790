import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import generalized_rbda_amd as G
G.LIB_PATH = os.path.abspath(sys.argv[1])
import oracle_py as O
from generalized_rbda_amd.states import random_states
for m in ("mit_humanoid", "jvrc1_humanoid"):
    blob = G.urdf_to_blob(os.path.join(ROOT, "tests/golden/robot-models", m + ".urdf"))
    plan = G.Plan(blob)
    for scale in (1.0, 20.0):
        q, qd, tau = random_states(blob, 2000, 7)
        q[:, 7:] *= scale
        q32, qd32, t32 = (a.astype(np.float32).astype(np.float64) for a in (q, qd, tau))
        ref = O.forward_dynamics(blob, q32, qd32, t32)
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
        got = plan.forward_dynamics(t(q32), t(qd32), t(t32)).double().cpu().numpy()
        err = (np.abs(got - ref).max(axis=1) / (1 + np.abs(ref).max(axis=1)))
        print(m, "scale", scale, "max rel err %.2e  median %.2e" % (err.max(), np.median(err)))
