"""One d ydd / d (q, qd, tau) call of a model per precision under rocprofv3 --kernel-trace: per-kernel totals of the LAST call.
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/deriv_trace.py [model] [B];  then  python3 tools/deriv_trace.py --sum DIR"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--sum":
    import csv, glob, collections
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # marker kernels: the calls are separated by a torch fill kernel (see below)
    calls, cur = [], []
    for r in rows:
        n = r["Kernel_Name"]
        if "grbda_hip" not in n:
            if cur: calls.append(cur); cur = []
            continue
        cur.append((n.split("(")[0].replace("void grbda_hip::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    if cur: calls.append(cur)
    for c in calls[-2:]:
        tot = collections.OrderedDict()
        for n, t in c: tot[n] = tot.get(n, 0.0) + t
        span = sum(tot.values())
        print("  ".join(f"{n.split('<')[0]}<{n.split('<')[1][:14]} {t:.3f} ms x{sum(1 for a, _ in c if a == n)}" for n, t in tot.items()), f"| sum {span:.3f} ms")
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
from generalized_rbda_amd.states import random_states
model = sys.argv[1] if len(sys.argv) > 1 else "jvrc1_humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
q, qd, tau = random_states(plan.blob, B, 2)
for dt in (torch.float64, torch.float32):
    t = lambda a: torch.as_tensor(a, dtype=dt, device="cuda:0")
    tq, tqd, tt = t(q), t(qd), t(tau)
    for _ in range(2):
        d = plan.fd_derivatives(tq, tqd, tt)
        torch.cuda.synchronize()
        del d
    marker = torch.zeros(16, device="cuda:0"); marker.fill_(1.0); torch.cuda.synchronize()
    d = plan.fd_derivatives(tq, tqd, tt)
    torch.cuda.synchronize()
    marker.fill_(2.0); torch.cuda.synchronize()
    del d
