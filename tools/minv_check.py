"""A/B of the derivative pipeline: H^-1 = W^T W from the articulated-body quantities (minv_kernels.hip) against the dense factorisation
route (GRBDA_NO_MINV=1), same inputs, both precisions; then timings.   usage: python tools/minv_check.py [B]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import random_states

def plans(make):
    os.environ["GRBDA_NO_MINV"] = "1"; old = make()
    os.environ["GRBDA_NO_MINV"] = "0"; new = make()
    return old, new

def check(name, make, B, time_it=False):
    old, new = plans(make)
    q, qd, tau = random_states(new.blob, B, 2)
    for dt in (torch.float64, torch.float32):
        t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
        tq, tqd, tt = t(q), t(qd), t(tau)
        a = old.fd_derivatives(tq, tqd, tt); b = new.fd_derivatives(tq, tqd, tt)
        torch.cuda.synchronize()
        errs = []
        for k in ("dq", "dqd", "dtau"):
            d = (a[k].double() - b[k].double()).abs().amax(dim=(1, 2)) / (1.0 + a[k].double().abs().amax(dim=(1, 2)))
            errs.append(f"{k} {d.max().item():.2e}{'' if torch.isfinite(b[k]).all() else ' NONFINITE'}")
        only = new.fd_dtau(tq); d = (only.double() - a["dtau"].double()).abs().max() / (1 + a["dtau"].double().abs().max())
        msg = f"{name:28s} nv {new.nv:3d} B {B:8d} {str(dt).split('.')[1]}: new vs old  " + "  ".join(errs) + f"  dtau alone {d.item():.2e}"
        if time_it:
            def timed(fn, n=3):
                fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n): fn()
                torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
            del a, b, only
            msg += f" | all three old {timed(lambda: old.fd_derivatives(tq, tqd, tt)):.3f} ms new {timed(lambda: new.fd_derivatives(tq, tqd, tt)):.3f} ms" \
                   f" | dtau old {timed(lambda: old.fd_dtau(tq)):.3f} new {timed(lambda: new.fd_dtau(tq)):.3f}"
        print(msg, flush=True)

if __name__ == "__main__":
    from models import zoo
    urdf = lambda n: (lambda: G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", n + ".urdf")))
    z = zoo()
    for name in sorted(z):
        p = G.Plan(z[name])
        info = p.info()
        if not info.analytic_derivatives or info.spanning_tree_route: continue
        from generalized_rbda_amd.states import has_implicit_clusters
        if has_implicit_clusters(z[name]): continue
        check(name, (lambda blob=z[name]: G.Plan(blob)), 257)
    big = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    check("jvrc1_humanoid", urdf("jvrc1_humanoid"), big, True)
    check("mit_humanoid", urdf("mit_humanoid"), max(big // 4, 1024), True)
    check("mini_cheetah", urdf("mini_cheetah"), max(big // 16, 1024), True)
