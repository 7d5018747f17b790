"""A THIRD evaluation of d ydd / d q on the constraint manifold near singular poses (review of round 4, soft spot 1b): the analytic
route (manifold_kernels.hip) and the difference batches (GRBDA_NO_MANIFOLD=1) disagree by up to 22 % on four_bar states, both in fp64.
Reference here: the oracle compiled in x87 extended precision (oracle/_build/libgrbda_oracle_ld.so, eps 1.1e-19), central differences
along an INDEPENDENT position with the dependent ones re-projected by Newton to |phi| < 1e-17, Richardson-extrapolated over two steps
(h, h / 2): truncation O(h^4), round-off eps / h.  States are binned by the condition number of the dependent constraint Jacobian K_d.
usage: python tools/manifold_third_eval.py [n_draw]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import generalized_rbda_amd as G
import oracle_py as O
from generalized_rbda_amd.states import parse_clusters, random_states

LD = np.longdouble


def reference_dq(blob, m, q, qd, tau, h):
    """d ydd / d (independent position k) by Richardson-extrapolated central differences in long double, on the manifold"""
    nv = qd.shape[0]
    q0 = O.project_positions_ld(blob, q[None])[0][0]
    cols = []
    for c in m["clusters"]:
        (pc, fb, kk, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) = c
        flags = m["ints"][io + 1: io + 1 + nsv] if ctype == 2 else (m["ints"][io: io + nsv] if ctype == 3 else None)
        for a in range(nvel):
            cols.append((qi + ([j for j in range(nsv) if flags[j]][a] if flags is not None else a)))
    J = np.empty((nv, nv), dtype=LD)
    def D(step, k):
        out = []
        for sgn in (+1, -1):
            qq = q0.copy()
            qq[cols[k]] += sgn * LD(step)
            qq, ok = O.project_positions_ld(blob, qq[None])
            assert ok[0]
            out.append(O.forward_dynamics_ld(blob, qq, qd[None].astype(LD), tau[None].astype(LD))[0])
        return (out[0] - out[1]) / (2 * LD(step))
    for k in range(nv):
        J[:, k] = (4 * D(h / 2, k) - D(h, k)) / 3
    return J.astype(np.float64), q0.astype(np.float64)


def main():
    n_draw = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    for name in ("four_bar", "planar_leg_linkage", "six_bar"):
        path = os.path.join(ROOT, "tests/golden/robot-models", name + ".urdf")
        os.environ.pop("GRBDA_NO_MANIFOLD", None)
        plan = G.Plan.from_urdf(path)
        os.environ["GRBDA_NO_MANIFOLD"] = "1"
        plan_d = G.Plan.from_urdf(path)
        os.environ.pop("GRBDA_NO_MANIFOLD", None)
        blob = plan.blob
        m = parse_clusters(blob)
        q, qd, tau = random_states(blob, n_draw, config_index=5)
        t64 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
        tq = t64(q)
        ok = plan.project_positions(tq).cpu().numpy()
        q = tq.cpu().numpy()[ok]; qd = qd[ok]; tau = tau[ok]
        kcond = O.spanning_state(blob, q, qd)[3]
        print(f"{name}: {ok.sum()} valid states of {n_draw}; cond(K_d) quantiles 50 / 99 / max: {np.quantile(kcond, .5):.3g} {np.quantile(kcond, .99):.3g} {kcond.max():.3g}")
        print(f"  {'cond(K_d) bin':>16s} {'states':>6s} {'analytic vs LD':>16s} {'differences vs LD':>18s} {'analytic vs diff':>17s} {'|J| max':>10s}")
        for lo, hi in ((1, 10), (10, 100), (100, 1e3), (1e3, 1e4), (1e4, 1e6)):
            idx = np.flatnonzero((kcond >= lo) & (kcond < hi))[:12]
            if idx.size == 0:
                continue
            ea, ed, ead, jm = 0.0, 0.0, 0.0, 0.0
            for i in idx:
                J_ref, q0 = reference_dq(blob, m, q[i], qd[i], tau[i], 1e-7)
                qa = t64(q0[None]); va = t64(qd[i][None]); ta = t64(tau[i][None])
                Ja = plan.fd_dq(qa, va, ta).cpu().numpy()[0]
                Jd = plan_d.fd_dq(qa, va, ta, step=1e-6).cpu().numpy()[0]
                sc = 1.0 + np.abs(J_ref).max()
                ea = max(ea, np.abs(Ja - J_ref).max() / sc); ed = max(ed, np.abs(Jd - J_ref).max() / sc)
                ead = max(ead, np.abs(Ja - Jd).max() / sc); jm = max(jm, np.abs(J_ref).max())
            print(f"  [{lo:7.0e},{hi:7.0e}) {idx.size:6d} {ea:16.2e} {ed:18.2e} {ead:17.2e} {jm:10.3g}", flush=True)


if __name__ == "__main__":
    main()
