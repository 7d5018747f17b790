"""Phase ablation of the chain kernel (MIT humanoid, f32): kernel time with the tile prologue / the segments / the
epilogue switched off (GRBDA_CHAIN_DEBUG bits 0 / 1 / 2; results are meaningless then), for 1, 2 and 4 rounds of the
persistent grid.  usage: python tools/chain_ablate.py   (spawns one child per configuration)"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def one(B):
    import torch
    import generalized_rbda_amd as G
    from generalized_rbda_amd.states import random_states
    plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", os.environ.get("ABL_MODEL", "mit_humanoid") + ".urdf"))
    q, qd, tau = random_states(plan.blob, B, 2)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    out = torch.empty((B, plan.nv), dtype=torch.float32, device="cuda:0")
    plan.time_kernel("aba", tq, tqd, tt, out, iters=3)
    print(f"{plan.time_kernel('aba', tq, tqd, tt, out, iters=30):.4f}", end=" ", flush=True)

if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        one(int(sys.argv[2]))
    else:
        for dbg, name in ((0, "full"), (1, "no prologue"), (4, "no epilogue"), (5, "segments only"), (2, "prologue+epilogue only"), (6, "prologue only"), (3, "epilogue only"),
                          (8, "full, slab rows aliased"), (13, "segments only, aliased")):
            print(f"{name:24s}", end=" ", flush=True)
            for B in (131072, 262144, 524288, 1048576):
                env = dict(os.environ, GRBDA_CHAIN_DEBUG=str(dbg))
                subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(B)], env=env, stderr=subprocess.DEVNULL)
            print("ms at B = 131072 262144 524288 1048576")
