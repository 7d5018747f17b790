"""fp32 accuracy of the TelloWithArms chain kernels against the device's own fp64 results on the same fp32 inputs.
usage: python tools/tello_acc.py lib1.so [lib2.so ...]   (each library in a child process)
Writes the worst states of the first library to gpurun_out/tello_worst.npz for analysis on the host."""
import os, subprocess, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(path, dump):
    import numpy as np, torch
    import generalized_rbda_amd as G
    G.LIB_PATH = os.path.abspath(path)
    from generalized_rbda_amd.robots import tello_with_arms
    from generalized_rbda_amd.states import random_states
    plan = G.Plan.from_model(tello_with_arms())
    B = int(os.environ.get("ACC_B", "1048576"))
    q, qd, tau = random_states(plan.blob, B, 3)
    t64 = torch.as_tensor(q, dtype=torch.float64, device="cuda:0")
    ok = plan.project_positions(t64).cpu().numpy()
    q = t64.cpu().numpy()
    good, bad = np.flatnonzero(ok), np.flatnonzero(~ok)
    q[bad] = q[good[np.arange(bad.size) % good.size]]
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device="cuda:0")
    q32, qd32, tau32 = t(q, torch.float32), t(qd, torch.float32), t(tau, torch.float32)
    ydd32 = plan.forward_dynamics(q32, qd32, tau32)
    tau32_out = plan.inverse_dynamics(q32, qd32, tau32)
    ydd_ref = plan.forward_dynamics(q32.double(), qd32.double(), tau32.double())
    tau_ref = plan.inverse_dynamics(q32.double(), qd32.double(), tau32.double())
    torch.cuda.synchronize()
    e_aba = ((ydd32.double() - ydd_ref).abs().amax(dim=1) / (1.0 + ydd_ref.abs().amax(dim=1))).cpu().numpy()
    e_rnea = ((tau32_out.double() - tau_ref).abs().amax(dim=1) / (1.0 + tau_ref.abs().amax(dim=1))).cpu().numpy()
    qs = [0.5, 0.99, 0.999, 0.9999, 1.0]
    print(os.path.basename(path), "good", good.size, "aba q", np.quantile(e_aba, qs), "n>1e-3", int((e_aba > 1e-3).sum()),
          "rnea q", np.quantile(e_rnea, qs), "n>1e-3", int((e_rnea > 1e-3).sum()), flush=True)
    gmax, kcond, status = plan.constraint_gain(t(q, torch.float64))
    gm = gmax.cpu().numpy()
    qmax = np.abs(q[:, 7:]).max(axis=1)
    print("status nonzero", int((status != 0).sum().item()), "gmax q", np.quantile(gm, [0, 0.5, 0.9, 0.99, 0.999, 1.0]))
    for thr in (10, 20, 30, 50, 100, 200, 500, 1e9):
        for qthr in (1e9, 20.0):
            keep = (gm < thr) & (qmax < qthr)
            print(f"  gmax < {thr:g} |q| < {qthr:g}: keep {keep.mean():.4f}  max e_aba {e_aba[keep].max():.2e}  max e_rnea {e_rnea[keep].max():.2e}"
                  f"  n>1e-3 {int((e_aba[keep] > 1e-3).sum())}/{int((e_rnea[keep] > 1e-3).sum())}", flush=True)
    if dump:
        w = np.argsort(-e_aba)[:300]
        w2 = np.argsort(-e_rnea)[:300]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez(os.path.join(ROOT, "gpurun_out", "tello_worst.npz"), q=q[w], qd=qd[w], tau=tau[w], e=e_aba[w],
                 ydd32=ydd32[w].cpu().numpy(), ydd64=ydd_ref[w].cpu().numpy(),
                 q_r=q[w2], qd_r=qd[w2], ydd_r=tau[w2], e_r=e_rnea[w2], tau32=tau32_out[w2].cpu().numpy(),
                 tau64=tau_ref[w2].cpu().numpy())


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(sys.argv[2], sys.argv[3] == "1")
    else:
        for i, p in enumerate(sys.argv[1:]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", p, "1" if i == 0 else "0"], check=False)
