"""Forward / inverse dynamics of the reference's parallel-chain family through the spanning-tree route: evaluations per second per
(depth, loop size), fp32 and fp64.  usage (GPU box): python tools/big_cluster_bench.py [B]"""
import os, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import generalized_rbda_amd as G
from models import valid_states
from parallel_chains import parallel_chain_urdf

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ONLY = os.environ.get("BIG_ONLY")  # e.g. "1,20,31": one model (implicit flag, depth, loop size)
for implicit, depth, loop in ([tuple(int(x) for x in ONLY.split(","))] if ONLY else []) or ((False, 10, 4), (False, 10, 12), (False, 10, 16), (False, 20, 30), (True, 10, 5), (True, 10, 13), (True, 10, 17), (True, 20, 31), (False, 40, 40), (True, 40, 41)):
    with tempfile.NamedTemporaryFile("w", suffix=".urdf", delete=False) as f:
        f.write(parallel_chain_urdf(depth, loop, implicit))
    plan = G.Plan.from_urdf(f.name)
    os.unlink(f.name)
    q, qd, tau = valid_states(plan.blob, 512, config_index=3, big=True, scale=0.5 if depth < 40 else 0.25)
    rep = (B + 511) // 512
    q, qd, tau = (np.tile(a, (rep, 1))[:B] for a in (q, qd, tau))
    for dt in (torch.float32, torch.float64):
        t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
        tq, tqd, tt = t(q), t(qd), t(tau)
        res = {}
        for name, fn in (("FD", lambda: plan.forward_dynamics(tq, tqd, tt)), ("ID", lambda: plan.inverse_dynamics(tq, tqd, tt))):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): fn()
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 3
        print(f"{'implicit' if implicit else 'explicit'} depth {depth} loop {loop} route {'spanning' if plan.info().spanning_tree_route else 'structured'} "
              f"{str(dt).split('.')[1]} B {B}: FD {res['FD'] * 1e3:.2f} ms ({B / res['FD']:.3g}/s)  ID {res['ID'] * 1e3:.2f} ms ({B / res['ID']:.3g}/s)", flush=True)
