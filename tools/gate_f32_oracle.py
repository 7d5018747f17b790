"""Evidence for the conditioning gate (generalized_rbda_amd/states.py): on the states the gate REJECTS, is single precision
itself the problem?  Per bucket of the constraint gain / condition number: max relative error against the fp64 oracle of
(a) the product's fp32 forward dynamics and (b) the ORACLE compiled in single precision (oracle/_build/libgrbda_oracle_f32.so:
the dense restatement of the reference's algorithm in `float`), on the same fp32-rounded inputs.
usage (GPU box): python tools/gate_f32_oracle.py [B] > profiles/r4_gate_f32_oracle.txt"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import generalized_rbda_amd as G
import oracle_py as O
from generalized_rbda_amd.states import random_states, accept, GATE_GMAX, GATE_KCOND, GATE_QMAX
from generalized_rbda_amd.robots import tello_with_arms

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
THREADS = len(os.sched_getaffinity(0))
print(f"# gate now: gain < {GATE_GMAX}, cond < {GATE_KCOND}, |q| < {GATE_QMAX}; round 3's: gain < 50, cond < 1000 -- severity below is relative to round 3's")
print(f"# B = {B} random states per model (reference sampling law + ONE Newton projection, no gate); states are listed when ROUND 3's gate rejects them")
print("# err = max_i |ydd_i - ydd64_i| / (1 + max_i |ydd64_i|) per state, against the fp64 ORACLE on the fp32-rounded inputs")
for name, build, cfg in (("tello_with_arms", tello_with_arms, 3), ("four_bar", None, 5), ("six_bar", None, 6)):
    plan = G.Plan.from_model(build()) if build else G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", name + ".urdf"))
    blob = plan.blob
    q, qd, tau = random_states(blob, B, cfg)
    t64 = torch.as_tensor(q, dtype=torch.float64, device="cuda:0")
    conv = plan.project_positions(t64).cpu().numpy()
    q = t64.cpu().numpy()
    q, qd, tau = q[conv], qd[conv], tau[conv]
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device="cuda:0")
    gmax, kcond, status = plan.constraint_gain(t(q, torch.float64))   # (the gate looks at the fp64 states, as bench.py and the tests do)
    gm, kc = gmax.cpu().numpy(), kcond.cpu().numpy()
    ok = accept(blob, q, np.where(gm < 50.0, gm, np.inf), np.where(kc < 1000.0, kc, np.inf)) & (status.cpu().numpy() == 0)  # round 3's gate
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    q, qd, tau = r32(q), r32(qd), r32(tau)   # what an fp32 caller hands over
    y32 = plan.forward_dynamics(t(q, torch.float32), t(qd, torch.float32), t(tau, torch.float32)).double().cpu().numpy()
    y64d = plan.forward_dynamics(t(q, torch.float64), t(qd, torch.float64), t(tau, torch.float64)).cpu().numpy()
    rej = np.flatnonzero(~ok)
    acc = np.flatnonzero(ok)
    rng = np.random.default_rng(1)
    sample_acc = rng.choice(acc, size=min(acc.size, 60000), replace=False)
    idx = np.concatenate([rej, sample_acc])
    ref = np.empty((idx.size, plan.nv))
    o32 = np.empty((idx.size, plan.nv))
    ref[:] = O.forward_dynamics_mt(blob, q[idx], qd[idx], tau[idx], THREADS)
    o32[:] = O.forward_dynamics_mt_f32(blob, q[idx], qd[idx], tau[idx], THREADS)
    den = 1.0 + np.abs(ref).max(axis=1)
    e_k32 = np.abs(y32[idx] - ref).max(axis=1) / den
    e_o32 = np.abs(o32 - ref).max(axis=1) / den
    e_k64 = np.abs(y64d[idx] - ref).max(axis=1) / den
    e_o32[~np.isfinite(e_o32)] = np.inf
    e_k32[~np.isfinite(e_k32)] = np.inf
    is_rej = np.arange(idx.size) < rej.size
    print(f"\n== {name}: {q.shape[0]} converged states, gate rejects {rej.size} ({100.0 * rej.size / q.shape[0]:.2f} %); fp64 kernel vs oracle on ALL {idx.size} compared states: max {e_k64.max():.2e}")
    print(f"accepted sample ({sample_acc.size}): kernel fp32 max {e_k32[~is_rej].max():.2e}   oracle fp32 max {e_o32[~is_rej].max():.2e}")
    sev = np.maximum(gm[idx] / 50.0, kc[idx] / 1000.0)
    sev = np.maximum(sev, np.abs(q[idx][:, -plan.nq:]).max(axis=1) / 1e9)
    print("states outside round 3's gate by severity = max(gain / 50, cond / 1000) (|q| >= 32 rad counted in its own row; the gate now ends at severity 3):")
    print(f"{'bucket':>22s} {'n':>7s} {'kernel f32 max':>15s} {'oracle f32 max':>15s} {'kernel>1e-3':>12s} {'oracle>1e-3':>12s} {'kernel>1e-3 & oracle<=1e-3':>28s}")
    qbig = np.zeros(idx.size, bool)
    from generalized_rbda_amd.states import parse_clusters
    for c in parse_clusters(blob)["clusters"]:
        if c[9] >= 2:
            qbig |= np.abs(q[idx][:, c[3]: c[3] + c[4]]).max(axis=1) >= GATE_QMAX
    edges = [1, 2, 5, 10, 100, 1e3, np.inf]
    lo = 0
    for hi in edges:
        sel = is_rej & ~qbig & (sev >= max(lo, 1e-30)) & (sev < hi) if lo else is_rej & ~qbig & (sev < hi)
        if sel.any():
            print(f"{f'[{lo:g}, {hi:g})':>22s} {int(sel.sum()):7d} {e_k32[sel].max():15.2e} {e_o32[sel].max():15.2e} {int((e_k32[sel] > 1e-3).sum()):12d} {int((e_o32[sel] > 1e-3).sum()):12d} {int(((e_k32[sel] > 1e-3) & (e_o32[sel] <= 1e-3)).sum()):28d}")
        lo = hi
    sel = is_rej & qbig
    if sel.any():
        print(f"{'|q| >= 32 rad':>22s} {int(sel.sum()):7d} {e_k32[sel].max():15.2e} {e_o32[sel].max():15.2e} {int((e_k32[sel] > 1e-3).sum()):12d} {int((e_o32[sel] > 1e-3).sum()):12d} {int(((e_k32[sel] > 1e-3) & (e_o32[sel] <= 1e-3)).sum()):28d}")
    both = is_rej
    print(f"all rejected: kernel fp32 > 1e-3 on {int((e_k32[both] > 1e-3).sum())}, oracle fp32 > 1e-3 on {int((e_o32[both] > 1e-3).sum())}, "
          f"kernel fails where the fp32 oracle passes: {int(((e_k32[both] > 1e-3) & (e_o32[both] <= 1e-3)).sum())}, "
          f"oracle fails where the kernel passes: {int(((e_o32[both] > 1e-3) & (e_k32[both] <= 1e-3)).sum())}")
