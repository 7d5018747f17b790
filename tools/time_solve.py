"""Time the derivative entry points that isolate the SPD solve: d ydd / d tau (H^-1 only), + d/dqd (one right-hand side),
all three.  usage: GRBDA_LIB=lib.so python tools/time_solve.py [model] [B]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
if os.environ.get("GRBDA_LIB"):
    G.LIB_PATH = os.path.abspath(os.environ["GRBDA_LIB"])
from generalized_rbda_amd.states import random_states
model = sys.argv[1] if len(sys.argv) > 1 else "jvrc1_humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
q, qd, tau = random_states(plan.blob, B, 2)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
tq, tqd, tt = t(q), t(qd), t(tau)
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
res = {"fd_dtau": timed(lambda: plan.fd_dtau(tq)), "dqd only": timed(lambda: plan.fd_derivatives(tq, tqd, tt, want=("dqd",))),
       "all three": timed(lambda: plan.fd_derivatives(tq, tqd, tt))}
print(os.path.basename(G.LIB_PATH), model, B, "  ".join(f"{k}={v:.3f}ms" for k, v in res.items()), flush=True)

try:
    import ctypes
    L = G.lib()
    L.grbda_debug_mf_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 8)()
    L.grbda_debug_mf_prof(buf, 1)
    plan.fd_derivatives(tq, tqd, tt); torch.cuda.synchronize()
    L.grbda_debug_mf_prof(buf, 0)
    names = ["H block copy + wait", "H row from LDS + rhs copy issue", "Cholesky", "L^-1", "GEMM1 + copy wait + H^-1 stores", "barrier + GEMM2", "result stores"]
    tot = sum(buf[:7])
    print("s_memtime ticks per state (all three):", "  ".join(f"{n}={buf[i] / B:.0f} ({100 * buf[i] / tot:.0f}%)" for i, n in enumerate(names)), " total", tot / B)
except AttributeError:
    pass

try:
    import ctypes
    L = G.lib()
    L.grbda_debug_mv_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 8)()
    L.grbda_debug_mv_prof(buf, 1)
    plan.fd_derivatives(tq, tqd, tt); torch.cuda.synchronize()
    L.grbda_debug_mv_prof(buf, 0)
    names = ["record copy + tile clear + wait", "walk", "barrier + rhs copy issue + GEMM1", "copy wait + H^-1 to LDS and out", "barrier", "GEMM2", "result stores"]
    tot = sum(buf[:7])
    print("minv_mfma_kernel, s_memtime ticks per state (all three):", "  ".join(f"{n}={buf[i] / B:.0f} ({100 * buf[i] / tot:.0f}%)" for i, n in enumerate(names)), " total", tot / B)
except AttributeError:
    pass
