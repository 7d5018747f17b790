"""Time applyTestForce and the inverse OSIM on one model, force-propagation route vs unit-wrench route (GRBDA_NO_EFPA=1).
usage: python tools/time_contact.py [mit_humanoid|mini_cheetah|tello] [B]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import random_states
model = sys.argv[1] if len(sys.argv) > 1 else "mit_humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
for no in ("0", "1"):
    os.environ["GRBDA_NO_EFPA"] = no
    if model == "tello":
        from generalized_rbda_amd.robots import tello_with_arms
        plan = G.Plan.from_model(tello_with_arms())
        bodies = [10, 20]
    else:
        plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
        import struct
        blob = plan.blob
        _, _, nb, nc = struct.unpack_from("<II2i", blob, 0)
        n_ints, n_dbls, n_names = struct.unpack_from("<3i", blob, 28)
        off = 96 + 416 * nb + 64 * nc + 4 * ((n_ints + 1) & ~1) + 8 * n_dbls
        names = [n.decode() for n in blob[off: off + n_names].split(b"\0")[:nb]]
        links = [i for i, n in enumerate(names) if "rotor" not in n.lower()]
        bodies = [links[-1], links[len(links) // 2]]  # two links (a contact on a rotor has no force-propagation path)
    q, qd, tau = random_states(plan.blob, B, 3)
    if model == "tello":
        t64 = torch.as_tensor(q, dtype=torch.float64, device="cuda:0"); ok = plan.project_positions(t64).cpu().numpy(); q = t64.cpu().numpy()
        good, bad = np.flatnonzero(ok), np.flatnonzero(~ok); q[bad] = q[good[np.arange(bad.size) % good.size]]
    tq = torch.as_tensor(q, dtype=torch.float32, device="cuda:0")
    f = torch.as_tensor(np.random.default_rng(1).uniform(-1, 1, (B, 3)), dtype=torch.float32, device="cuda:0")
    def timed(fn, n=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(model, B, "f32 NO_EFPA=" + no, "apply_test_force %.3f ms" % timed(lambda: plan.apply_test_force(tq, bodies[0], [0, 0, -0.05], f)),
          "inv_osim(2) %.3f ms" % timed(lambda: plan.inv_osim(tq, bodies, [[0, 0, -0.05], [0, 0.03, -0.02]])), flush=True)
