import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import random_states, parse_clusters
from models import zoo
np.set_printoptions(linewidth=250, precision=3, suppress=True)
z = zoo()
for name in sys.argv[1:]:
    blob = z[name]
    os.environ["GRBDA_NO_MINV"] = "1"; old = G.Plan(blob)
    os.environ["GRBDA_NO_MINV"] = "0"; new = G.Plan(blob)
    m = parse_clusters(blob)
    print(name, "nv", new.nv)
    for i, c in enumerate(m["clusters"]):
        (pc, fb, k, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) = c
        print(f"  cluster {i}: parent cluster {pc} first_body {fb} k {k} v_index {vi} n {nvel} type {ctype}")
    q, qd, tau = random_states(blob, 4, 2)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device="cuda:0")
    a = old.fd_dtau(t(q))[0].cpu().numpy(); b = new.fd_dtau(t(q))[0].cpu().numpy()
    bad = np.abs(a - b) > 1e-9 * (1 + np.abs(a).max())
    print("wrong entries (rows, then cols):", sorted(set(np.where(bad)[0])), sorted(set(np.where(bad)[1])))
    Hm = old.mass_matrix(t(q))[0].cpu().numpy()
    print("old H^-1 H - 1:", np.abs(a @ Hm - np.eye(new.nv)).max(), " new:", np.abs(b @ Hm - np.eye(new.nv)).max())
    print((b @ Hm - np.eye(new.nv)))
