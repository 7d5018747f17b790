"""Phase ablation of minv_mfma_kernel (experiment build with -DGRBDA_EXP_MV_ABL): time of d ydd / d tau alone and of all three with one phase
removed at a time (results are wrong; only the times mean something).  usage: GRBDA_LIB=build/exp/libgrbda_mvabl.so python tools/mv_ablate.py [B]"""
import os, sys, time, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
G.LIB_PATH = os.path.abspath(os.environ["GRBDA_LIB"])
from generalized_rbda_amd.states import random_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models/jvrc1_humanoid.urdf"))
q, qd, tau = random_states(plan.blob, B, 2)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
tq, tqd, tt = t(q), t(qd), t(tau)
def timed(fn, n=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
L = G.lib()
for bits, what in [(0, "everything"), (1, "no result stores"), (2, "no H^-1 store"), (3, "no stores at all"), (4, "no walk"), (8, "no second product"),
                   (16, "no right-hand-side copy"), (32, "no first product"), (64, "no record copy"), (127, "nothing but barriers and the tile clear")]:
    L.grbda_debug_mv_abl(bits)
    print(f"{what:42s} dtau {timed(lambda: plan.fd_dtau(tq)):7.3f} ms   all three {timed(lambda: plan.fd_derivatives(tq, tqd, tt)):7.3f} ms", flush=True)
