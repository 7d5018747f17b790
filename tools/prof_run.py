import ctypes, os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import generalized_rbda_amd as G
G.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "prof", "libgrbda_hip_prof.so")
from generalized_rbda_amd.states import random_states
plan = G.Plan.from_urdf("tests/golden/robot-models/mit_humanoid.urdf")
B = 262144
q, qd, tau = random_states(plan.blob, B, 2)
dev = torch.device("cuda:0")
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
tq, tqd, tt = t(q), t(qd), t(tau)
out = torch.empty((B, plan.nv), dtype=torch.float32, device=dev)
L = G.lib()
L.grbda_debug_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3): plan.forward_dynamics(tq, tqd, tt, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
L.grbda_debug_profile(buf, 1)
n = 10
for _ in range(n): plan.forward_dynamics(tq, tqd, tt, out=out)
torch.cuda.synchronize()
L.grbda_debug_profile(buf, 0)
names = ["stage", "step:group wait+issue", "fwd", "bwd(other)", "acc", "bwd:loop-top", "bwd:body-record", "bwd:consts+kinematics", "bwd:bias+acc-loads", "bwd:math+handover", "bwd:joint-terms+push", "bwd:solve+K-store", "step:records", "acc:entry drain", "acc:K/y0/ap loads", "acc:ydd+out store", "acc:body record", "acc:consts+sincos", "tile epilogue", "tile entry drain"]
tot = sum(buf[i] for i in range(13))  # buckets >= 13 are sub-buckets of acc
print("per-wave-per-launch s_memtime ticks (shader cycles), MIT humanoid ABA f32, profiling build:")
for i, nm in enumerate(names):
    print(f"  {nm:36s} {buf[i]/n/2048:12.0f}  {100*buf[i]/tot:5.1f}%")
print("  total", tot / n / 2048)
