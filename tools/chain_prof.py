"""In-kernel cycle accounting of the chain kernel (make expc NAME=cprof DEFS=-DGRBDA_CHAIN_PROFILE): s_memtime deltas per tile phase and
segment type, summed over wavefronts.  usage: python tools/chain_prof.py [model] [B]"""
import ctypes, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import generalized_rbda_amd as G
G.LIB_PATH = os.environ.get("GRBDA_HIP_LIB", os.path.join(ROOT, "build", "exp", "libgrbda_cprof.so"))
from generalized_rbda_amd.states import random_states
model = sys.argv[1] if len(sys.argv) > 1 else "mit_humanoid"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
plan = G.Plan.from_urdf(os.path.join(ROOT, "tests/golden/robot-models", model + ".urdf"))
q, qd, tau = random_states(plan.blob, B, 2)
dev = torch.device("cuda:0")
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
tq, tqd, tt = t(q), t(qd), t(tau)
out = torch.empty((B, plan.nv), dtype=torch.float32, device=dev)
L = G.lib()
L.grbda_debug_chain_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(3): plan.forward_dynamics(tq, tqd, tt, out=out)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
L.grbda_debug_chain_profile(buf, 1)
n = 10
for _ in range(n): plan.forward_dynamics(tq, tqd, tt, out=out)
torch.cuda.synchronize()
L.grbda_debug_chain_profile(buf, 0)
ms = plan.time_kernel("aba", tq, tqd, tt, out, iters=20)
names = {0: "stage inputs", 1: "epilogue", 2: "free fwd", 3: "run fwd", 4: "run bwd (leaf/slot head)", 5: "free bwd", 6: "free acc", 7: "run acc", 8: "pair acc",
         9: "diff fwd", 10: "diff bwd", 11: "diff acc", 13: "gen fwd", 14: "gen bwd", 15: "gen acc", 20: "run bwd (pair head)"}
tiles = (B + 63) // 64
tot = sum(buf[i] for i in range(32))
print(f"{model} B={B} kernel {ms:.4f} ms (profiling build); s_memtime ticks per TILE, phases and segment types:")
for i, nm in names.items():
    if buf[i]:
        cnt = buf[32 + i] / n / tiles
        print(f"  {nm:28s} {buf[i]/n/tiles:10.0f} ticks/tile {100*buf[i]/tot:5.1f}%   units/tile {cnt:5.1f}   ticks/unit {buf[i]/n/tiles/cnt:8.0f}")
print(f"  total {tot/n/tiles:.0f} ticks per tile")
marks = {"run fwd": (0, ["top: next record, prefetch issue, g0 (scalar waits)", "sin / cos", "transform (constants + math)", "LDS store", "copies (vm wait)"], 3),
         "run bwd": (8, ["top: record, prefetch issue, LDS blk, chat", "bias force, IA sum, u, D", "link_up (transform)", "rotor", "rcp, K store", "rank-1 update", "copies (vm wait)"], None),
         "run acc": (20, ["top: record, prefetch issue, ydd", "put", "child transform", "copies (vm wait)"], 7)}
if os.environ.get("CPROF_CHUNKED"):
    marks["run acc"] = (20, ["chunk: records + loads issued + parent (v, a)", "wait for the first [K | y0] block", "phase A (constants, sin / cos, E)", "phase B (recursion, all links of the chunk)"], 7)
nb = buf[32 + 4] + buf[32 + 20]
for nm, (base, labels, cnt_bucket) in marks.items():
    links = (buf[32 + cnt_bucket] if cnt_bucket is not None else nb) / n / tiles
    print(f"  {nm}: per link ({links:.0f} links per tile)")
    for k, lab in enumerate(labels):
        print(f"      {lab:55s} {buf[64 + base + k]/n/tiles/links:8.0f} ticks")
