"""Marginal cost of the phases of the chain kernel in the LOADED machine: kernel time with one phase removed (GRBDA_CHAIN_DEBUG bits of an
ablation build, make variant VARIANT=abl VFLAGS=-DGRBDA_EXP; results are meaningless then).  usage: python tools/chain_ablate2.py [model]"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def one(model, B):
    import torch
    import generalized_rbda_amd as G
    from generalized_rbda_amd.states import random_states
    plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
    q, qd, tau = random_states(plan.blob, B, 2)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    out = torch.empty((B, plan.nv), dtype=torch.float32, device="cuda:0")
    plan.time_kernel("aba", tq, tqd, tt, out, iters=3)
    print(f"{plan.time_kernel('aba', tq, tqd, tt, out, iters=30):.4f}", end=" ", flush=True)

if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--one":
        one(sys.argv[2], int(sys.argv[3]))
    else:
        model = sys.argv[1] if len(sys.argv) > 1 else "mit_humanoid"
        for dbg, name in ((0, "full"), (1, "no prologue"), (4, "no epilogue"), (32, "no forward segments"), (64, "no backward segments"),
                          (128, "no acceleration segments"), (160, "no forward, no acceleration"), (224, "no segments at all (loop only)"),
                          (5, "segments only"), (8, "full, slab rows aliased"), (0, "full (again)")):
            print(f"{name:34s}", end=" ", flush=True)
            for B in (131072, 262144, 1048576):
                env = dict(os.environ, GRBDA_CHAIN_DEBUG=str(dbg))
                subprocess.run([sys.executable, os.path.abspath(__file__), "--one", model, str(B)], env=env, stderr=subprocess.DEVNULL)
            print("ms at B = 131072 262144 1048576", flush=True)
