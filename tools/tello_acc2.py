"""fp32 accuracy of the TelloWithArms forward / inverse dynamics of several builds of the library against the fp64 ORACLE, on the
gated synthetic states of bench.py.  usage: python tools/tello_acc2.py lib1.so [lib2.so ...]"""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))


def one(path):
    import numpy as np, torch
    import generalized_rbda_amd as G
    G.LIB_PATH = os.path.abspath(path)
    import oracle_py as O
    from generalized_rbda_amd.robots import tello_with_arms
    from generalized_rbda_amd.states import valid_random_states_device
    plan = G.Plan.from_model(tello_with_arms())
    B = int(os.environ.get("ACC_B", "262144"))
    q, qd, tau, nd = valid_random_states_device(plan, B, 3, "cuda:0")
    r32 = lambda a: a.astype(np.float32).astype(np.float64)
    q, qd, tau = r32(q), r32(qd), r32(tau)
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device="cuda:0")
    y32 = plan.forward_dynamics(t(q, torch.float32), t(qd, torch.float32), t(tau, torch.float32)).double().cpu().numpy()
    i32 = plan.inverse_dynamics(t(q, torch.float32), t(qd, torch.float32), t(tau, torch.float32)).double().cpu().numpy()
    n = min(B, 60000)
    th = len(os.sched_getaffinity(0))
    ref = O.forward_dynamics_mt(plan.blob, q[:n], qd[:n], tau[:n], th)
    o32 = O.forward_dynamics_mt_f32(plan.blob, q[:n], qd[:n], tau[:n], th)
    den = 1 + np.abs(ref).max(axis=1)
    ek = np.abs(y32[:n] - ref).max(axis=1) / den
    eo = np.abs(o32 - ref).max(axis=1) / den
    i64 = plan.inverse_dynamics(t(q, torch.float64), t(qd, torch.float64), t(tau, torch.float64)).cpu().numpy()
    ei = np.abs(i32 - i64).max(axis=1) / (1 + np.abs(i64).max(axis=1))
    qs = [0.5, 0.9, 0.99, 0.999, 1.0]
    print(f"{os.path.basename(path):28s} ABA f32 kernel q{qs} = {np.quantile(ek, qs)}  | float oracle {np.quantile(eo, qs)} | RNEA f32 vs f64 kernel {np.quantile(ei, qs)}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        for p in sys.argv[1:]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", p], check=False)
