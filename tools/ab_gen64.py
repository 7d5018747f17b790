"""fp64 forward dynamics of models whose chain programs carry generic clusters (aba_chain_kernel<double, WPS, 2>): time per call.
usage: python tools/ab_gen64.py [B]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import generalized_rbda_amd as G
from models import zoo, valid_states
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
z = zoo()
names = sys.argv[2].split(",") if len(sys.argv) > 2 else [k for k in z]
for name in names:
    if name not in z: continue
    plan = G.Plan(z[name])
    q, qd, tau = valid_states(z[name], 512, 3)
    rep = (B + 511) // 512
    q, qd, tau = (np.tile(a, (rep, 1))[:B] for a in (q, qd, tau))
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    fn = lambda: plan.forward_dynamics(tq, tqd, tt)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(name, plan.kernel_name("aba", "f64", B), "fp64 FD %.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3), flush=True)
