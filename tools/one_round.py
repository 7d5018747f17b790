"""Per-tile latency of the forward dynamics against wavefronts per SIMD: kernel ms at one tile per wavefront slot for GRBDA_WAVES_PER_CU_ABA32 = 4 / 8
(one / two wavefronts per SIMD) and at several tiles per wavefront.   usage: python tools/one_round.py [tello|model.urdf name]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import valid_random_states_device
from generalized_rbda_amd.robots import tello_with_arms

name = sys.argv[1] if len(sys.argv) > 1 else "tello"
for wpc in ("4", "8"):
    os.environ["GRBDA_WAVES_PER_CU_ABA32"] = wpc
    os.environ["GRBDA_NO_LATENCY_MODE"] = "1"
    plan = G.Plan.from_model(tello_with_arms()) if name == "tello" else G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", name + ".urdf"))
    for B in (65536, 131072, 262144, 524288, 1048576):
        q, qd, tau, _ = valid_random_states_device(plan, B, 3, "cuda:0")
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
        tq, tqd, tt = t(q), t(qd), t(tau)
        out = torch.empty((B, plan.nv), dtype=torch.float32, device="cuda:0")
        plan.time_kernel("aba", tq, tqd, tt, out, iters=5)
        ms = min(plan.time_kernel("aba", tq, tqd, tt, out, iters=30) for _ in range(3))
        slots = 256 * int(wpc)
        print(f"{name} waves per CU {wpc}: B {B:8d} = {B // 64 / slots:5.2f} tiles per wavefront slot  {ms:.4f} ms  ({ms / (B / 131072):.4f} per 131 072 states)", flush=True)
