"""Experiment: a plan-specialised, fully unrolled ABA kernel (build/spec/spec.hip, tables dumped by
grbda_debug_dump_plan) against the interpreter kernel: same results? how much faster?"""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import random_states

plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models/mit_humanoid.urdf"))
S = ctypes.CDLL(os.path.join(ROOT, "build/spec/libspec.so"))
S.spec_launch_aba_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
B = 262144
q, qd, tau = random_states(plan.blob, B, 2)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
tq, tqd, tt = t(q), t(qd), t(tau)
ref = plan.forward_dynamics(tq, tqd, tt)
grid = 2048
rows = S.spec_scratch_rows()
scratch = torch.zeros(grid * rows * 64 + 64, dtype=torch.float32, device="cuda:0")
lds = max(S.spec_lds_slots() * 256, 64 * (plan.nq + 2 * plan.nv) * 4)
out = torch.empty_like(ref)
st = torch.cuda.current_stream().cuda_stream
def run():
    rc = S.spec_launch_aba_f32(tq.data_ptr(), tqd.data_ptr(), tt.data_ptr(), out.data_ptr(), B, scratch.data_ptr(), grid, lds, ctypes.c_void_p(st))
    assert rc == 0, rc
run(); torch.cuda.synchronize()
err = ((out - ref).abs().max(dim=1).values / (1 + ref.abs().max(dim=1).values)).max().item()
print("max rel diff vs interpreter kernel: %.3e" % err)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print("specialised kernel: %.4f ms" % (e0.elapsed_time(e1) / 20))
print("interpreter kernel: %.4f ms" % plan.time_kernel("aba", tq, tqd, tt, out, iters=20))
