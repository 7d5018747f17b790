"""Single-GPU prediction of the 1 -> 8 GPU strong-scaling curve (BASELINE configs 3, 4, 5): the kernel time at B, B/2, B/4, B/8 states
(hipEvents around the kernel, grbda_time_kernel) and the implied efficiency of an N-way batch split, t(B) / N / t(B / N) -- the
batch-sharded design has no data-path collective, so a rank's step IS its kernel.  usage: python tools/strong_scaling_proxy.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import valid_random_states_device
from generalized_rbda_amd.robots import tello_with_arms

CASES = [("mit_humanoid", 262144, "aba", torch.float32, 2), ("mit_humanoid", 262144, "rnea", torch.float32, 2),
         ("tello", 1048576, "aba", torch.float32, 3), ("tello", 1048576, "rnea", torch.float32, 3),
         ("jvrc1_humanoid", 1048576, "aba", torch.float32, 4), ("four_bar", 1048576, "aba", torch.float32, 5),
         ("six_bar", 1048576, "aba", torch.float32, 6), ("mini_cheetah", 65536, "aba", torch.float64, 1)]
print("# kernel ms (min of 3 x 30 launches) at B / N states and the efficiency of an N-way split, t(B) / (N t(B / N))")
for name, B, algo, dt, cfg in CASES:
    plan = G.Plan.from_model(tello_with_arms()) if name == "tello" else G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", name + ".urdf"))
    q, qd, tau, _ = valid_random_states_device(plan, B, cfg, "cuda:0")
    t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    times = {}
    for n in (1, 2, 4, 8):
        b = B // n
        out = torch.empty((b, plan.nv), dtype=dt, device="cuda:0")
        a, v, x = tq[:b].contiguous(), tqd[:b].contiguous(), tt[:b].contiguous()
        plan.time_kernel(algo, a, v, x, out, iters=5)
        times[n] = min(plan.time_kernel(algo, a, v, x, out, iters=30) for _ in range(3))
    dts = "f32" if dt == torch.float32 else "f64"
    line = f"{name:15s} {algo:4s} {dts} B={B:8d} " + " ".join(f"N={n}: {times[n]:.4f} ms ({plan.kernel_name(algo, dts, B // n)})" for n in (1, 2, 4, 8))
    eff = " ".join(f"eff{n}={times[1] / n / times[n]:.3f}" for n in (2, 4, 8))
    print(line)
    print(f"{'':15s} {'':4s}    {eff}", flush=True)
