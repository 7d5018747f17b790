"""Stress run (not part of the test suite): many freshly drawn random models through forward / inverse dynamics (against
the oracle), the analytic derivatives (against the difference batches of a second plan) and the force-propagation
inverse OSIM (against the unit-wrench route).  usage: python tools/stress_random_models.py [n_seeds]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import generalized_rbda_amd as G
import oracle_py as O
from models import chain_test_tree, random_cluster_tree, valid_states

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
rel = lambda a, b: float(np.abs(a - b).max() / (1.0 + np.abs(b).max()))
bad = 0
shapes = {}
for seed in range(1000, 1000 + n):
    for kind in ("chain", "tree_float", "tree_fixed", "chain_fixed_like", "tree_big"):
        if kind == "chain":
            m = chain_test_tree(seed, n_limbs=1 + seed % 5, ori_repr="rpy" if seed % 4 == 0 else "quaternion", rotors=seed % 3 != 0,
                                deep_pairs=seed % 2 == 1)
        elif kind == "tree_float":
            m = random_cluster_tree(seed, n_clusters=3 + seed % 9, floating=True, ori_repr="rpy" if seed % 7 == 0 else "quaternion")
        elif kind == "tree_fixed":
            m = random_cluster_tree(seed, n_clusters=2 + seed % 8, floating=False)
        elif kind == "tree_big":  # Generic clusters of 9-20 bodies / 5-12 coordinates among ordinary ones: the spanning-tree route
            m = random_cluster_tree(seed, n_clusters=2 + seed % 3, floating=seed % 2 == 0, kinds=("generic_big", "rev", "rotor", "generic_big", "generic"))
        else:
            m = random_cluster_tree(seed, n_clusters=2 + seed % 10, floating=False, kinds=("rev", "axirotor", "rotor"))
        blob = m.serialize()
        os.environ.pop("GRBDA_NO_ANALYTIC", None); os.environ.pop("GRBDA_NO_EFPA", None)
        plan = G.Plan(blob)
        info = plan.info()
        key = (kind, info.chain_aba_f32, info.chain_rnea_f32, info.analytic_derivatives, "diffs>0" if info.n_chain_differentials else "")
        shapes[key] = shapes.get(key, 0) + 1
        B = 70
        big = kind == "tree_big"  # (the oracle build with room for 48 bodies per cluster)
        if big and plan.nv > 64:
            continue
        q, qd, tau = valid_states(blob, B, config_index=seed, big=big)
        errs = {}
        for dt, tol in ((torch.float64, 1e-9), (torch.float32, 1e-3)):
            c = (lambda a: a) if dt == torch.float64 else (lambda a: a.astype(np.float32).astype(np.float64))
            errs[f"aba{dt}"] = (rel(plan.forward_dynamics(t(q, dt), t(qd, dt), t(tau, dt)).double().cpu().numpy(), O.forward_dynamics(blob, c(q), c(qd), c(tau), big=big)), tol)
            errs[f"rnea{dt}"] = (rel(plan.inverse_dynamics(t(q, dt), t(qd, dt), t(tau, dt)).double().cpu().numpy(), O.inverse_dynamics(blob, c(q), c(qd), c(tau), big=big)), tol)
        if big:
            H64 = plan.mass_matrix(t(q[:8])).cpu().numpy()
            Hinv = plan.fd_dtau(t(q[:8])).cpu().numpy()
            errs["H Hinv"] = (float(np.abs(np.einsum("bij,bjk->bik", H64, Hinv) - np.eye(plan.nv)).max()), 1e-7)
            fails = {k: v for k, v in errs.items() if not (v[0] < v[1])}
            if fails:
                bad += 1
                print("FAIL", kind, seed, {k: f"{v[0]:.2e}" for k, v in fails.items()}, "info", key, flush=True)
            continue
        d = plan.fd_derivatives(t(q[:6]), t(qd[:6]), t(tau[:6]))
        os.environ["GRBDA_NO_ANALYTIC"] = "1"; os.environ["GRBDA_NO_EFPA"] = "1"
        slow = G.Plan(blob)
        errs["dtau"] = (rel(d["dtau"].cpu().numpy(), slow.fd_dtau(t(q[:6])).cpu().numpy()), 1e-8)
        errs["dqd"] = (rel(d["dqd"].cpu().numpy(), slow.fd_dqd(t(q[:6]), t(qd[:6]), t(tau[:6])).cpu().numpy()), 1e-8)
        errs["dq"] = (rel(d["dq"].cpu().numpy(), slow.fd_dq(t(q[:6]), t(qd[:6]), t(tau[:6]), step=1e-6).cpu().numpy()), 2e-5)
        # mass matrix: fp64 (plain CRBA) against the oracle, fp32 (packed rows + in-place unpack) against fp64
        H64 = plan.mass_matrix(t(q[:66])).cpu().numpy()
        z4 = np.zeros((4, plan.nv))
        c0 = O.inverse_dynamics(blob, q[:4], z4, z4)
        Href = np.stack([O.inverse_dynamics(blob, q[:4], z4, np.tile(np.eye(plan.nv)[k], (4, 1))) - c0 for k in range(plan.nv)], axis=2)
        errs["H64"] = (rel(H64[:4], Href), 1e-9)
        errs["H32"] = (rel(plan.mass_matrix(t(q[:66], torch.float32)).double().cpu().numpy(), H64), 1e-4)
        if info.analytic_derivatives:
            d32 = plan.fd_derivatives(t(q[:66], torch.float32), t(qd[:66], torch.float32), t(tau[:66], torch.float32))
            d64 = plan.fd_derivatives(t(q[:66]), t(qd[:66]), t(tau[:66]))
            errs["dtau32"] = (rel(d32["dtau"].double().cpu().numpy(), d64["dtau"].cpu().numpy()), 2e-2)
        names = [b.name for b in m.bodies]
        cand = [i for i, nm in enumerate(names) if not nm.startswith("r")]
        rng = np.random.default_rng(seed)
        bodies = [int(x) for x in rng.choice(cand, size=min(3, len(cand)), replace=False)]
        off = rng.uniform(-0.2, 0.2, size=(len(bodies), 3))
        L1 = plan.inv_osim(t(q[:6]), bodies, off).cpu().numpy()
        L2 = slow.inv_osim(t(q[:6]), bodies, off).cpu().numpy()
        errs["osim"] = (rel(L1, L2), 1e-7)
        fails = {k: v for k, v in errs.items() if not (v[0] < v[1])}
        if fails:
            bad += 1
            print("FAIL", kind, seed, {k: f"{v[0]:.2e}" for k, v in fails.items()}, "info", key, flush=True)
print("models", sum(shapes.values()), "failures", bad)
for k, v in sorted(shapes.items()):
    print("  (kind, chain_aba_f32, chain_rnea_f32, analytic, explicit pairs as differentials):", k, "x", v)
sys.exit(1 if bad else 0)
