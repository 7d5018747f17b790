"""Time ABA / RNEA kernels of one or more builds of the library (experiments).
usage: python tools/exp_run.py lib1.so [lib2.so ...]   -- each is timed in a child process"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def one(path):
    import torch
    import generalized_rbda_amd as G
    G.LIB_PATH = os.path.abspath(path)
    from generalized_rbda_amd.states import random_states
    res = []
    for urdf, prec in [(m, 32) for m in os.environ.get("EXP_MODELS", "mit_humanoid,mini_cheetah,jvrc1_humanoid").split(",")] + [(m, 64) for m in os.environ.get("EXP_MODELS64", "mit_humanoid,mini_cheetah").split(",") if m]:
        if urdf == "tello":
            from generalized_rbda_amd.robots import tello_with_arms
            plan = G.Plan.from_model(tello_with_arms())
        else:
            plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", urdf + ".urdf"))
        B = int(os.environ.get("EXP_B", "262144"))
        q, qd, tau = random_states(plan.blob, B, 2)
        if urdf == "tello":  # valid spanning positions: Newton projection on the device, failures replaced
            import numpy as np
            t64 = torch.as_tensor(q, dtype=torch.float64, device="cuda:0")
            ok = plan.project_positions(t64).cpu().numpy()
            q = t64.cpu().numpy()
            good, bad = np.flatnonzero(ok), np.flatnonzero(~ok)
            q[bad] = q[good[np.arange(bad.size) % good.size]]
        dt = torch.float32 if prec == 32 else torch.float64
        t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
        tq, tqd, tt = t(q), t(qd), t(tau)
        out = torch.empty((B, plan.nv), dtype=dt, device="cuda:0")
        for kind in ("aba", "rnea"):
            plan.time_kernel(kind, tq, tqd, tt, out, iters=3)
            ms = plan.time_kernel(kind, tq, tqd, tt, out, iters=20)
            res.append(f"{urdf}/f{prec}/{kind}={ms:.4f}ms")
    print(os.path.basename(path), " ".join(res), flush=True)

if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        for spec in sys.argv[1:]:  # [VAR=val,VAR=val:]lib.so
            envs, _, p = spec.rpartition(":")
            env = dict(os.environ)
            for kv in filter(None, envs.split(",")):
                k, _, v = kv.partition("=")
                env[k] = v
            print(envs, end=" ", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", p], check=False, env=env, stderr=subprocess.DEVNULL)
