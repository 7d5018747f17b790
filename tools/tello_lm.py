"""TelloWithArms forward dynamics (fp32) in latency mode against the one-wavefront kernel: kernel ms by batch size (lm: the library's choice; two: GRBDA_LM_WAVES=2;
one: GRBDA_NO_LATENCY_MODE=1).   usage: python tools/tello_lm.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import valid_random_states_device
from generalized_rbda_amd.robots import tello_with_arms
plans = {}
for label, env in (("lm", {}), ("two", {"GRBDA_LM_WAVES": "2"}), ("one", {"GRBDA_NO_LATENCY_MODE": "1"})):
    os.environ.update(env)
    plans[label] = G.Plan.from_model(tello_with_arms())
    for k in env: del os.environ[k]
for B in (64, 16384, 32768, 49152, 65536, 131072):
    q, qd, tau, _ = valid_random_states_device(plans["lm"], B, 3, "cuda:0")
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    out = torch.empty((B, plans["lm"].nv), dtype=torch.float32, device="cuda:0")
    row = []
    for label, plan in plans.items():
        plan.time_kernel("aba", tq, tqd, tt, out, iters=5)
        ms = min(plan.time_kernel("aba", tq, tqd, tt, out, iters=30) for _ in range(3))
        row.append(f"{label} {ms:.4f} ms ({plan.kernel_name('aba', 'f32', B).split('::')[-1].replace('aba_chain_', '')})")
    print(f"tello B {B:6d}  " + "  ".join(row), flush=True)
