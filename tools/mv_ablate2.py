import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
G.LIB_PATH = os.path.abspath(os.environ["GRBDA_LIB"])
from generalized_rbda_amd.states import random_states
B = 1048576
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models/jvrc1_humanoid.urdf"))
q, qd, tau = random_states(plan.blob, B, 2)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
tq, tqd, tt = t(q), t(qd), t(tau)
def timed(fn, n=4):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
L = G.lib()
for bits, what in [(0, "everything"), (4+8+32, "memory only (no walk, no products)"), (1+2+16+64, "compute only (no copies, no stores)"), (127, "skeleton"),
                   (1+2, "no stores"), (16+64, "no loads"), (8+32, "no products")]:
    L.grbda_debug_mv_abl(bits)
    print(f"{what:42s} all three {timed(lambda: plan.fd_derivatives(tq, tqd, tt)):7.3f} ms", flush=True)
