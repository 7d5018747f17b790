"""In-kernel cycle accounting of the derivative recursion (make expd NAME=dprof DEFS=-DGRBDA_DERIV_PROFILE): s_memtime deltas per phase.
usage: GRBDA_HIP_LIB=build/exp/libgrbda_dprof.so python tools/deriv_prof.py [model] [B]"""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import random_states
model = sys.argv[1] if len(sys.argv) > 1 else "jvrc1_humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
q, qd, tau = random_states(plan.blob, B, 2)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
tq, tqd, tt = t(q), t(qd), t(tau)
L = G.lib()
L.grbda_debug_deriv_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
plan.fd_derivatives(tq, tqd, tt); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L.grbda_debug_deriv_profile(buf, 1)
n = 3
for _ in range(n): plan.fd_derivatives(tq, tqd, tt)
torch.cuda.synchronize()
L.grbda_debug_deriv_profile(buf, 0)
names = ["pass 1 (kinematics of bodies with children, root first)", "pass 2 body: records, kinematics, composites", "pass 2 body: joint terms, in-cluster ancestors, hand-over",
         "pass 2 cluster: parent accumulator row, own entries stored", "base cluster", "walk: one ancestor cluster (rows, dot products, stores)", "walk: base columns + exit", "tile end"]
tiles = (B + 63) // 64
tot = sum(buf[i] for i in range(8))
print(f"{model} B={B} fp32 rnea_deriv_kernel, s_memtime ticks per TILE of 64 states (profiling build):")
for i, nm in enumerate(names):
    cnt = buf[32 + i] / n / tiles
    print(f"  {nm:62s} {buf[i]/n/tiles:10.0f} ticks/tile {100*buf[i]/max(tot,1):5.1f}%   events/tile {cnt:6.1f}   ticks/event {buf[i]/n/tiles/max(cnt,1e-9):8.0f}")
print(f"  total {tot/n/tiles:.0f} ticks per tile")
