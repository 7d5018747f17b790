#!/usr/bin/env python3
"""bench.py -- forward-dynamics evals/sec on batched random states, MIT Humanoid cluster model.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (batched cluster ABA, ClusterTreeModel::forwardDynamics)
over one batch of B synthetic states per GPU, inputs and outputs resident in HBM.  Scaling: weak by
default (every rank owns its own B-state shard); `--scaling strong` splits the workload's batch over the
ranks (BASELINE config 4: 1 048 576 Tello states over 8 GPUs).  States are independent and the model plan is
replicated, so there is no data-path collective inside a step; `value` times the K steps alone.  The one
exchange the path has -- results to rank 0 over RCCL -- is measured separately: `gather_ms` for one
blocking gather, and `end_to_end` for K steps that each gather their results, the gather of step i running
beside the kernel of step i + 1 (two result buffers).  After the timed region a strided sample of the results is
checked against the CPU oracle (`verified`: max error of the sample under the tolerance).
`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (torch.distributed.run as a child
process, before this process touches a GPU) and relays rank 0's line and the children's exit code.
The K steps are timed `--replays` times (default 7), every repetition bracketed by barrier + synchronize; `value` is
the MEDIAN repetition, `spread` carries min / max.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (guides: MI355X_MICROARCH.md chip table)
VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}
# what a pure v_fma_f32 stream reaches on this chip (tools/valu_ubench.hip, profiles/r2_valu_ubench.txt: 1.31 ns per
# wave-instruction per SIMD at 8 waves/SIMD, the clock sagging to ~1.9 GHz under that load; the guide quotes 103 TF)
VALU_ATTAINABLE_TFLOPS = {"f32": 100.0}

WORKLOADS = {
    # name: (urdf, batch per GPU, dtype, config index for the RNG seed)
    "mit_humanoid": ("mit_humanoid.urdf", 262144, "f32", 2),
    "mini_cheetah": ("mini_cheetah.urdf", 65536, "f64", 1),
    "revolute_rotor_chain": ("revolute_rotor_chain.urdf", 1024, "f64", 0),
    "jvrc1_humanoid": ("jvrc1_humanoid.urdf", 1048576, "f32", 4),
    # hand-built TelloWithArms (the URDF carries no constraints): implicit differentials
    "tello": ("<TelloWithArms>", 1048576, "f32", 3),
    # BASELINE config 5's URDF+ loop clusters
    "four_bar": ("four_bar.urdf", 1048576, "f32", 5),
    "six_bar": ("six_bar.urdf", 1048576, "f32", 6),
}


def kernel_sources_sha():
    """sha256 (16 hex digits) over the device-side sources of the library: the PMC files under profiles/ carry the value they were
    taken on (tools/collect_profiles.py), and counters of other sources are refused below instead of printed next to a fresh time."""
    import glob, hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "generalized_rbda_amd", "csrc")
    # (device code AND the host code that picks kernels, waves per CU, chunking and the plan layout: capi.cpp, plan.cpp)
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def physical_cores():
    """distinct (package, core) pairs of /proc/cpuinfo; falls back to the logical count"""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def usable_cpus():
    """hardware threads this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one
    (a GPU box can show 256 logical CPUs and grant a handful)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(blob, q, qd, tau, budget_s=20.0, passes=5):
    """Oracle (CPU restatement of the reference algorithm, fp64) on the host cores: single thread and all hardware
    threads, median of `passes` passes each, on a bounded sample of the same batch."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    threads = usable_cpus()

    def rate(fn, n_threads, n):
        t0 = time.perf_counter()
        fn(blob, q[:n], qd[:n], tau[:n], n_threads)
        return n / max(time.perf_counter() - t0, 1e-9)

    def median_rate(fn, n_threads, budget):
        r = rate(fn, n_threads, min(q.shape[0], 256 * n_threads))  # sizing pass
        n = int(min(q.shape[0], max(64 * n_threads, r * budget / (2 * passes))))
        rs = sorted(rate(fn, n_threads, n) for _ in range(passes))
        return rs[len(rs) // 2], n

    single, n1 = median_rate(O.forward_dynamics_mt, 1, budget_s)
    multi, nm = median_rate(O.forward_dynamics_mt, threads, budget_s)
    # the same restatement compiled in single precision (oracle/Makefile), a shorter sample
    single32, _ = median_rate(O.forward_dynamics_mt_f32, 1, budget_s / 3)
    multi32, _ = median_rate(O.forward_dynamics_mt_f32, threads, budget_s / 3)
    return {"value": multi, "unit": "evals/s", "cores": threads, "physical_cores": physical_cores(),
            "logical_cpus_visible": os.cpu_count(), "kind": "port",
            "single_thread": single, "per_thread_all_core": multi / threads, "scaling_over_single_thread": multi / single,
            "passes": passes, "dtype": "f64", "f32": {"value": multi32, "single_thread": single32},
            "sample": f"first {nm} ({n1} single-thread) states of the same batch, median of {passes} passes; "
                      f"oracle/ = dense 6k x 6k plain-C restatement of the reference algorithm (the parity checker)",
            "reference_chart_evals_per_s_per_core": 2.0e4,
            "reference_chart_note": "images/ForwardDynamicsBenchmark.png of the reference (C-ABA, MIT Humanoid, "
                                    "hardware unstated), NOT re-measured: the reference cannot be built here (DESIGN.md)"}


def verify_sample(blob, q, qd, x, out, algo, dtype_name, n=1024):
    """strided sample of the results of the LAST timed step against the oracle (fed the inputs as the device saw them)"""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    B = q.shape[0]
    idx = np.unique(np.concatenate([np.arange(min(64, B)), np.linspace(0, B - 1, num=min(n, B), dtype=np.int64),
                                    np.arange(max(0, B - 64), B)]))
    c = (lambda a: a.astype(np.float32).astype(np.float64)) if dtype_name == "f32" else (lambda a: a)
    fn = O.forward_dynamics if algo == "aba" else O.inverse_dynamics
    if algo == "aba":
        ref = O.forward_dynamics_mt(blob, c(q[idx]), c(qd[idx]), c(x[idx]), os.cpu_count() or 1)
    else:
        ref = fn(blob, c(q[idx]), c(qd[idx]), c(x[idx]))
    got = out[idx].double().cpu().numpy()
    err = np.abs(got - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))
    tol = 1e-3 if dtype_name == "f32" else 1e-9
    return {"verified": bool(err.max() < tol and np.isfinite(got).all()), "verify_states": int(idx.size),
            "verify_max_rel_err": float(err.max()), "verify_states_over_tol": int((err >= tol).sum()), "verify_tol": tol}


def load_plan(G, workload):
    """the plan of a WORKLOADS entry (TelloWithArms is hand-built: its URDF carries no constraints)"""
    urdf = WORKLOADS[workload][0]
    if workload == "tello":
        from generalized_rbda_amd.robots import tello_with_arms

        return G.Plan.from_model(tello_with_arms())
    return G.Plan.from_urdf(os.path.join(ROOT, "tests", "golden", "robot-models", urdf))


def committed_counters(sha, workload, algo, dtype_name, B):
    """(traffic bytes per launch, source, flops per eval, source) from the PMC files committed under profiles/ -- only counters
    taken on THESE kernel sources (a file records the sha of the sources it was measured on; others are refused, not printed
    next to a fresh time)"""
    traffic, traffic_src, stale = None, None, []
    for fname in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", fname)) as f:
                doc = json.load(f)
            if doc.get("kernel_sources_sha") != sha:
                stale.append(fname)
                continue
            for e in doc["entries"]:
                if traffic is None and (e["workload"], e["algo"], e["dtype"], e["batch"]) == (workload, algo, dtype_name, B):
                    traffic = e["bytes_per_launch"]
                    traffic_src = f"rocprofv3 FETCH_SIZE + WRITE_SIZE per launch, profiles/{fname} (kernel sources {sha})"
        except (OSError, KeyError, ValueError):
            pass
    if traffic is None and stale:
        traffic_src = f"no counters for kernel sources {sha}: {', '.join('profiles/' + x for x in stale)} were taken on other sources (refused)"
    flops_pmc, flops_pmc_src = None, None
    for fname in ("r6_pmc_flops.json", "r5_pmc_flops.json", "r4_pmc_flops.json", "r3_pmc_flops.json"):
        try:
            with open(os.path.join(ROOT, "profiles", fname)) as f:
                doc = json.load(f)
            if doc.get("kernel_sources_sha") != sha:
                continue
            for e in doc["entries"]:
                if flops_pmc is None and (e["workload"], e["algo"], e["dtype"]) == (workload, algo, dtype_name):
                    flops_pmc, flops_pmc_src = e["flops_per_eval"], f"profiles/{fname}: " + e["source"]
        except (OSError, KeyError, ValueError):
            pass
    return traffic, traffic_src, flops_pmc, flops_pmc_src


def verify_derivatives(plan, q, qd, tau, d, dtype_name, n=24):
    """d ydd / d (q, qd, tau) of a strided sample against differences of the ORACLE's forward dynamics along the reference's
    tangent step (the reference's own yardstick, testRigidBodyDynamicsAlgosDerivatives.cpp:309-380): d tau exact (affine),
    d qd exact (quadratic), d q central differences with h = 1e-6."""
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from generalized_rbda_amd.states import parse_clusters, tangent_step

    blob, nv, B = plan.blob, plan.nv, q.shape[0]
    idx = np.unique(np.concatenate([[0, min(63, B - 1), min(64, B - 1)], np.linspace(0, B - 1, num=min(n, B), dtype=np.int64), [B - 1]]))
    c = (lambda a: a.astype(np.float32).astype(np.float64)) if dtype_name == "f32" else (lambda a: a)
    qs, qds, ts = c(q[idx]), c(qd[idx]), c(tau[idx])
    m = parse_clusters(blob)
    threads = os.cpu_count() or 1
    fd = lambda a, b, x: O.forward_dynamics_mt(blob, a, b, x, threads)
    h, ns = 1e-6, idx.size
    ref = {k: np.empty((ns, nv, nv)) for k in d}
    f0 = fd(qs, qds, ts)
    for k in range(nv):
        e = np.zeros(nv)
        e[k] = 1.0
        if "dq" in d:
            qp = np.stack([tangent_step(m, qs[i], k, +h) for i in range(ns)])
            qm = np.stack([tangent_step(m, qs[i], k, -h) for i in range(ns)])
            ref["dq"][:, :, k] = (fd(qp, qds, ts) - fd(qm, qds, ts)) / (2 * h)
        if "dqd" in d:
            ref["dqd"][:, :, k] = (fd(qs, qds + e, ts) - fd(qs, qds - e, ts)) / 2.0
        if "dtau" in d:
            ref["dtau"][:, :, k] = fd(qs, qds, ts + e) - f0
    tol = 1e-3 if dtype_name == "f32" else 2e-5  # (fp64: the central differences themselves, the reference's 2e-5)
    worst, finite = 0.0, True
    sel = torch.as_tensor(idx, device=next(iter(d.values())).device)
    for k in d:
        got = d[k][sel].double().cpu().numpy()
        finite = finite and bool(np.isfinite(got).all())
        worst = max(worst, float((np.abs(got - ref[k]).max(axis=(1, 2)) / (1.0 + np.abs(ref[k]).max(axis=(1, 2)))).max()))
    return {"verified": bool(finite and worst < tol), "verify_states": int(ns), "verify_max_rel_err": worst, "verify_tol": tol}


# The BASELINE configs beside the headline one, each as a compact sub-record of the default single-GPU line (`configs`): the
# driver's run is the only measurement anyone but the builder takes, so every config is in it.
CONFIG_RECORDS = [
    # (BASELINE config, workload, algo, dtype, batch)
    (1, "revolute_rotor_chain", "aba", "f64", 1024),
    (2, "mini_cheetah", "aba", "f64", 65536),
    (3, "mit_humanoid", "rnea", "f32", 262144),
    (3, "mit_humanoid", "fd_derivatives", "f32", 262144),
    (4, "tello", "aba", "f32", 1048576),
    (4, "tello", "rnea", "f32", 1048576),
    (5, "jvrc1_humanoid", "aba", "f32", 1048576),
    (5, "jvrc1_humanoid", "fd_derivatives", "f32", 1048576),
    (5, "jvrc1_humanoid", "fd_derivatives", "f64", 1048576),
    (5, "four_bar", "aba", "f32", 1048576),
    (5, "six_bar", "aba", "f32", 1048576),
    # one GPU's share of an 8-way STRONG split of configs 3 and 4 (SURVEY 8e's 0.9 is about these batches): the step of a rank is its kernel on B / 8 states,
    # so `split_efficiency` = t(B) / (8 t(B / 8)) with t(B) from this same line is what an 8-GPU node would show (tools/strong_scaling_proxy.py)
    (3, "mit_humanoid", "aba", "f32", 32768, 8),
    (3, "mit_humanoid", "rnea", "f32", 32768, 8),
    (4, "tello", "aba", "f32", 131072, 8),
]


def config_records(G, dev, steps, sha, only=None, headline_kernel_ms=None):
    """one sub-record per CONFIG_RECORDS row: K launches bracketed by synchronize (median of 3 repetitions), the kernel's own
    duration from hipEvents (grbda_time_kernel), a strided sample against the oracle after the timed region"""
    import numpy as np
    import torch

    from generalized_rbda_amd.states import valid_random_states_device

    out_recs, cache = [], {}
    for row in CONFIG_RECORDS:
        cfg_no, workload, algo, dtype_name, B = row[:5]
        split = row[5] if len(row) > 5 else None
        if only and workload not in only:
            continue
        rec = {"config": cfg_no, "workload": workload, "algo": algo, "dtype": dtype_name, "batch": B}
        if split:
            rec["split"] = split
        try:
            if workload not in cache:
                cache.clear()  # (one workload's inputs at a time: a million-state batch in fp64 on the host is ~1 GB)
                plan = load_plan(G, workload)
                cache[workload] = (plan,) + tuple(valid_random_states_device(plan, B, WORKLOADS[workload][3], dev)[:3])
            plan, q, qd, x = cache[workload]
            q, qd, x = q[:B], qd[:B], x[:B]  # (a split record takes the head of its workload's batch)
            tdt = torch.float32 if dtype_name == "f32" else torch.float64
            elem = 4 if dtype_name == "f32" else 8
            tq, tqd, tx = (torch.as_tensor(a, dtype=tdt, device=dev) for a in (q, qd, x))
            info = plan.info()
            if algo == "fd_derivatives":
                # all three matrices of d ydd / d (q, qd, tau): ms per batch (the bar of this row is quoted in ms per 1 048 576 states)
                d = plan.fd_derivatives(tq, tqd, tx)
                torch.cuda.synchronize()
                reps = []
                for _ in range(3):
                    del d
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    d = plan.fd_derivatives(tq, tqd, tx)
                    torch.cuda.synchronize()
                    reps.append(time.perf_counter() - t0)
                ms = sorted(reps)[1] * 1e3
                result_bytes = (3 * plan.nv * plan.nv + plan.nq + 2 * plan.nv) * elem
                rec.update({"ms": ms, "evals_per_s": B / (ms * 1e-3), "ms_min": min(reps) * 1e3, "ms_max": max(reps) * 1e3,
                            "what": "d ydd / d q, d qd, d tau (three [B, nv, nv] arrays) in one call, analytic",
                            "kernel": "aba + rnea_deriv_kernel + spd solve (deriv_kernels.hip), chunked",
                            "roofline": {"bound": "hbm", "bytes_per_eval": result_bytes, "achieved": B / (ms * 1e-3) * result_bytes / 1e9,
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": B / (ms * 1e-3) * result_bytes / 1e9 / HBM_PEAK_GBS},
                            "valu": None})
                rec.update(verify_derivatives(plan, q, qd, x, d, dtype_name))
                del d
            else:
                run = plan.forward_dynamics if algo == "aba" else plan.inverse_dynamics
                out = torch.empty((B, plan.nv), dtype=tdt, device=dev)
                for _ in range(3):
                    run(tq, tqd, tx, out=out)
                torch.cuda.synchronize()
                reps = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        run(tq, tqd, tx, out=out)
                    torch.cuda.synchronize()
                    reps.append((time.perf_counter() - t0) / steps)
                ms = sorted(reps)[1] * 1e3
                kernel_ms = plan.time_kernel(algo, tq, tqd, tx, out, iters=max(5, min(steps, 50)))
                bytes_per_eval = (plan.nq + 3 * plan.nv) * elem
                _, _, flops_pmc, _ = committed_counters(sha, workload, algo, dtype_name, B)
                flops = flops_pmc if flops_pmc is not None else (info.flops_aba if algo == "aba" else info.flops_rnea)
                k_rate = B / (kernel_ms * 1e-3)
                rec.update({"ms": ms, "evals_per_s": B / (ms * 1e-3), "kernel_ms": kernel_ms, "steps": steps,
                            "kernel": plan.kernel_name(algo, dtype_name, B, dev.index or 0),
                            "roofline": {"bound": "hbm", "bytes_per_eval": bytes_per_eval, "achieved": k_rate * bytes_per_eval / 1e9,
                                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k_rate * bytes_per_eval / 1e9 / HBM_PEAK_GBS},
                            "valu": {"flops_per_eval": flops, "flops_source": "pmc" if flops_pmc is not None else "plan model",
                                     "achieved_tflops": k_rate * flops / 1e12, "peak_tflops": VALU_PEAK_TFLOPS[dtype_name],
                                     "frac": k_rate * flops / 1e12 / VALU_PEAK_TFLOPS[dtype_name]}})
                rec.update(verify_sample(plan.blob, q, qd, x, out, algo, dtype_name))
                del out
            del tq, tqd, tx
        except Exception as e:  # noqa: BLE001 -- a failing sub-record must not take the headline line with it
            rec.update({"error": f"{type(e).__name__}: {e}", "verified": False})
        out_recs.append(rec)
    # the strong-split shares: efficiency of the N-way split from the kernel times of THIS run (the headline's for config 3)
    for rec in out_recs:
        if not rec.get("split") or "kernel_ms" not in rec:
            continue
        full = [r for r in out_recs if r is not rec and r.get("workload") == rec["workload"] and r.get("algo") == rec["algo"] and
                r.get("dtype") == rec["dtype"] and r.get("batch") == rec["batch"] * rec["split"] and "kernel_ms" in r]
        full_ms = full[0]["kernel_ms"] if full else (headline_kernel_ms if rec["workload"] == "mit_humanoid" and rec["algo"] == "aba" else None)
        if full_ms:
            rec["full_batch_kernel_ms"] = full_ms
            rec["split_efficiency"] = full_ms / (rec["split"] * rec["kernel_ms"])
    return out_recs


def strong_split_record(G, dist, rank, world, dev, steps, replays, warmup):
    """BASELINE config 4 as it is named: ONE global batch of 1 048 576 TelloWithArms states split over the ranks (contiguous slabs,
    plan replicated, no data-path collective).  Timed like the main line: K steps (one hipGraph replay when it captures), bracketed by
    barrier + synchronize, MAX over the ranks, median of the repetitions.  Every rank calls this; rank 0 gets the record.  A weak-scaling
    run shows ~1.0 by construction; this sub-record is what a driver's N-GPU run needs to see the STRONG split of config 4."""
    import numpy as np
    import torch

    from generalized_rbda_amd.robots import tello_with_arms
    from generalized_rbda_amd.sharding import shard_range
    from generalized_rbda_amd.states import valid_random_states_device

    _, B_global, dtype_name, cfg = WORKLOADS["tello"]
    plan = G.Plan.from_model(tello_with_arms())
    # every rank draws the SAME global batch itself (counter-based RNG, one seed; the Newton projection of the dependent
    # coordinates runs on its own GPU) and keeps its contiguous slab: no broadcast of a gigabyte of fp64 states through rank 0
    # and no extra barriers in the headline run.  Bounded: at most 20 steps, 3 repetitions.
    steps, replays = min(steps, 20), min(replays, 3)
    q, qd, x, _ = valid_random_states_device(plan, B_global, cfg, dev)
    lo, hi = shard_range(B_global, rank, world)
    tdt = torch.float32
    tq, tqd, tx = (torch.as_tensor(a[lo:hi], dtype=tdt, device=dev) for a in (q, qd, x))
    out = torch.empty((hi - lo, plan.nv), dtype=tdt, device=dev)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(1, warmup)):
        plan.forward_dynamics(tq, tqd, tx, out=out)
    barrier()
    reps = []
    for _ in range(max(1, replays)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            plan.forward_dynamics(tq, tqd, tx, out=out)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        reps.append(t.item())
    elapsed = sorted(reps)[len(reps) // 2]
    kernel_ms = plan.time_kernel("aba", tq, tqd, tx, out, iters=max(5, min(steps, 50)))
    if rank != 0:
        return None
    rec = {"what": "BASELINE config 4: ONE batch of 1 048 576 TelloWithArms states split over the ranks (strong scaling), cluster ABA fp32",
           "scaling": "strong", "n_gpus": world, "batch_global": B_global, "batch_per_gpu": hi - lo, "steps": steps,
           "value": B_global * steps / elapsed, "unit": "evals/s", "ms_per_step": elapsed / steps * 1e3, "launch": "loop",
           "kernel_ms_rank0": kernel_ms, "kernel": plan.kernel_name("aba", dtype_name, hi - lo, dev.index or 0),
           "replays": len(reps), "ms_per_step_min": min(reps) / steps * 1e3, "ms_per_step_max": max(reps) / steps * 1e3}
    rec.update(verify_sample(plan.blob, q[lo:hi], qd[lo:hi], x[lo:hi], out, "aba", dtype_name))
    return rec


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as children of THIS process (which has not touched
    a GPU and does not: no torch.cuda call before or after), relay rank 0's JSON line, exit with the children's code."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.lstrip().startswith("{")]
    if p.returncode != 0 or len(lines) != 1:
        sys.stderr.write(f"bench: {n}-rank run failed (exit code {p.returncode}, {len(lines)} result lines)\n")
        sys.stderr.write(p.stdout[-2000:])
        raise SystemExit(p.returncode or 1)
    print(lines[0])
    raise SystemExit(0)


def spawn_selftest():
    """BENCH_SPAWN_SELFTEST=1 (tests/test_bench_spawn_cpu.py): the ranks only rendezvous over gloo and sum their ranks --
    checks the self-spawn / relay plumbing where there is no GPU.  Never taken by a measurement."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    if os.environ.get("BENCH_SPAWN_SELFTEST") == "fail" and rank == world - 1:
        raise SystemExit(3)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "rank_sum": t.item()}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="mit_humanoid", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="states per GPU (default: the workload's)")
    ap.add_argument("--dtype", default="", choices=["", "f32", "f64"])
    ap.add_argument("--algo", default="aba", choices=["aba", "rnea"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time a plain loop of K launches instead of one hipGraph replay")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's batch per GPU; strong: the workload's batch split over the GPUs")
    ap.add_argument("--replays", type=int, default=7, help="repetitions of the K timed steps (median reported)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` sub-records (the other BASELINE configs)")
    ap.add_argument("--configs", default="", help="comma-separated workloads to keep in `configs` (default: all)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])  # does not return
    if os.environ.get("BENCH_SPAWN_SELFTEST"):
        return spawn_selftest()

    import numpy as np
    import torch

    import generalized_rbda_amd as G
    from generalized_rbda_amd.sharding import shard_range
    from generalized_rbda_amd.states import valid_random_states_device

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # (BENCH_FORCE_DIST=1: take the multi-rank code path even with one rank -- a self-test of the RCCL plumbing)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        # RCCL prints a version banner on the process's stdout when the communicator comes up; the contract is
        # ONE JSON line on stdout, so file descriptor 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=dev)
            dist.barrier()
            warm = torch.zeros(8, device=dev)
            dist.gather(warm, [torch.zeros(8, device=dev) for _ in range(world)] if rank == 0 else None, dst=0)
            torch.cuda.synchronize()
        finally:
            # the banner sits in the C library's stdio buffer (fully buffered on a pipe) until someone flushes it:
            # flush it NOW, while descriptor 1 still points at stderr
            sys.stdout.flush()
            try:
                import ctypes

                ctypes.CDLL(None).fflush(None)
            except OSError:
                pass
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    urdf, B, dtype_name, cfg = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    B_global = B if args.scaling == "strong" else B * world
    if args.dtype:
        dtype_name = args.dtype
    tdt = torch.float32 if dtype_name == "f32" else torch.float64
    plan = load_plan(G, args.workload)
    blob = plan.blob
    info = plan.info()

    # Synthetic inputs: reference sampling law, counter-based RNG (generalized_rbda_amd/states.py).  Implicit clusters take
    # spanning positions on the constraint manifold: Newton projection of the dependent coordinates (GenericJoint.cpp:289-385)
    # on the device, then the conditioning gate of states.py; rejected states are replaced by accepted ones (input
    # generation only, outside the timed region).
    if args.scaling == "strong":
        # ONE global batch; rank 0 draws (and Newton-projects) it, the others receive it; this rank computes its contiguous slab
        # (sharding.shard_range)
        import numpy as np

        if rank == 0 or dist is None:
            q, qd, x, n_distinct = valid_random_states_device(plan, B_global, cfg, dev)
        else:
            q, qd, x, n_distinct = np.empty((B_global, plan.nq)), np.empty((B_global, plan.nv)), np.empty((B_global, plan.nv)), B_global
        if dist is not None and world > 1:
            for a in (q, qd, x):
                tb = torch.as_tensor(a, dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
                dist.broadcast(tb, src=0)
                a[...] = tb.cpu().numpy()
        lo, hi = shard_range(B_global, rank, world)
        q, qd, x = q[lo:hi], qd[lo:hi], x[lo:hi]
        B = hi - lo
    else:
        # weak: every rank owns its own B-state shard of a world * B batch, drawn from its own RNG stream
        q, qd, x, n_distinct = valid_random_states_device(plan, B, cfg + 1000 * rank, dev)
    tq = torch.as_tensor(q, dtype=tdt, device=dev)
    tqd = torch.as_tensor(qd, dtype=tdt, device=dev)
    tx = torch.as_tensor(x, dtype=tdt, device=dev)
    out = torch.empty((B, plan.nv), dtype=tdt, device=dev)
    run = plan.forward_dynamics if args.algo == "aba" else plan.inverse_dynamics

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run(tq, tqd, tx, out=out)
    barrier()
    # The K timed steps are K launches of the same kernel: captured once into a hipGraph (torch.cuda.CUDAGraph on ROCm) and
    # replayed, which takes the host's per-launch work out of the gaps between kernels (0.170 -> 0.164 ms per step on the
    # headline workload).  The C ABI only enqueues on the stream it is given, so it captures as it is.  Any failure to
    # capture falls back to the plain loop; `launch` in the JSON says which one was timed.
    graph, launch = None, "loop"
    if not args.no_graph and args.steps > 0:
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # the library keeps its per-wave slab per (device, stream): allocate it before capture
                run(tq, tqd, tx, out=out, stream=side)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                for _ in range(args.steps):
                    run(tq, tqd, tx, out=out, stream=torch.cuda.current_stream(dev))
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            g.replay()  # untimed: instantiation / upload of the graph
            torch.cuda.synchronize()
            graph, launch = g, f"hipGraph of {args.steps} kernel launches, one replay"
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"bench: hipGraph capture failed ({e}); timing the plain launch loop\n")
            graph = None
            torch.cuda.synchronize()
    # R repetitions of EXACTLY K steps, each bracketed by barrier + synchronize on both sides and reduced with MAX over the
    # ranks; `value` is the median repetition
    reps = []
    for _ in range(max(1, args.replays)):
        barrier()
        t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(args.steps):
                run(tq, tqd, tx, out=out)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        reps.append(el)
    elapsed = sorted(reps)[len(reps) // 2]

    # kernel-only duration: hipEvents on the launch stream (torch's current stream here)
    kernel_ms = plan.time_kernel(args.algo, tq, tqd, tx, out, iters=max(5, min(args.steps, 50)))

    # the path's one exchange step: results to rank 0 over RCCL.  (a) one blocking gather; (b) K steps that each gather
    # their results, the gather of step i overlapping the kernel of step i + 1 (two result buffers)
    gather_ms, e2e_elapsed = None, None
    if dist is not None:
        bufs = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
        barrier()
        g0 = time.perf_counter()
        dist.gather(out, bufs, dst=0)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        outs = [out, torch.empty_like(out)]
        pending = [None, None]
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if pending[i % 2] is not None:
                pending[i % 2].wait()  # the gather that last read this buffer is ordered before the kernel below
            run(tq, tqd, tx, out=outs[i % 2])
            pending[i % 2] = dist.gather(outs[i % 2], bufs, dst=0, async_op=True)
        for w in pending:
            if w is not None:
                w.wait()
        barrier()
        e2e_elapsed = time.perf_counter() - t0
        t = torch.tensor([e2e_elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e2e_elapsed = t.item()
        run(tq, tqd, tx, out=out)  # `out` holds the results of rank 0's own shard again for the check below
        torch.cuda.synchronize()

    # N > 1 (weak, the default workload): the strong split of BASELINE config 4 beside the weak value
    strong = None
    if dist is not None and args.scaling == "weak" and args.workload == "mit_humanoid" and args.algo == "aba" and \
            os.environ.get("BENCH_NO_STRONG") != "1":
        strong = strong_split_record(G, dist, rank, world, dev, args.steps, args.replays, args.warmup)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    elem = 4 if dtype_name == "f32" else 8
    bytes_per_eval = (plan.nq + 3 * plan.nv) * elem  # q, qd, tau in + ydd out
    total_states = B_global if args.scaling == "strong" else world * B
    evals_per_s = total_states * args.steps / elapsed
    kernel_evals_per_s = B / (kernel_ms * 1e-3)
    achieved_gbs = kernel_evals_per_s * bytes_per_eval / 1e9
    flops = info.flops_aba if args.algo == "aba" else info.flops_rnea
    # measured HBM-side traffic of the same launch configuration, if a PMC run is committed (profiles/)
    # (only counters taken on THESE kernel sources: a file records the sha of the sources it was measured on)
    sha = kernel_sources_sha()
    traffic, traffic_src, flops_pmc, flops_pmc_src = committed_counters(sha, args.workload, args.algo, dtype_name, B)
    from generalized_rbda_amd.states import parse_clusters

    general = any(c[9] >= 2 for c in parse_clusters(blob)["clusters"])
    # the kernel this batch ran on, from the library's own selection (grbda_kernel_name)
    kernel_name = plan.kernel_name(args.algo, dtype_name, B, dev.index or 0)
    line = {
        "metric": "forward-dynamics evals/sec (batched random states), MIT Humanoid cluster model"
        if args.workload == "mit_humanoid" and args.algo == "aba"
        else f"{'forward' if args.algo == 'aba' else 'inverse'}-dynamics evals/sec, {args.workload}",
        "value": evals_per_s,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": dtype_name,
        "data": "synthetic",
        "launch": launch,
        "config": {"workload": f"{urdf} cluster-{'ABA' if args.algo == 'aba' else 'RNEA'}, {B} random states per GPU",
                   "batch_per_gpu": B, "batch_global": total_states, "nq": plan.nq, "nv": plan.nv, "n_bodies": plan.n_bodies,
                   "n_clusters": plan.n_clusters, "parallelism": f"batch-sharded x{world}, plan replicated"},
        # `bound` names the resource `achieved` / `peak` / `frac` are figures OF: the HBM roofline of the ALGORITHMIC bytes, as the
        # contract asks.  `binding_measured` names the resource the measurements say binds the kernel (per-wavefront latency: dependent
        # VALU issue and scalar / LDS / slab waits at two wavefronts per SIMD -- profiles/r5_chain_phase_profile.txt); its figures are
        # under `valu`
        "roofline": {"bound": "hbm", "binding_measured": "valu_issue", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": bytes_per_eval * B,
                     "frac_traffic": None if traffic is None else traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "kernel": kernel_name,
                     "kernel_ms": kernel_ms, "bytes_per_eval": bytes_per_eval,
                     "note": "`achieved`/`frac` price the ALGORITHMIC bytes as the contract asks; `frac_traffic` prices the "
                             "HBM-side bytes rocprofv3 counted; what binds is instruction issue -- two wavefronts per SIMD run at ~85 % of the rate this instruction mix (dependent VALU chains, scalar / LDS / slab waits) issues at, profiles/r5_chain_phase_profile.txt -- not bytes and not flops (see valu; DESIGN.md 4)",
                     "valu": {"flops_model": flops, "flops_pmc": flops_pmc, "flops_pmc_source": flops_pmc_src,
                              "flops_per_eval": flops_pmc if flops_pmc is not None else flops,
                              "achieved_tflops": kernel_evals_per_s * (flops_pmc if flops_pmc is not None else flops) / 1e12,
                              "peak_tflops": VALU_PEAK_TFLOPS[dtype_name],
                              "frac": kernel_evals_per_s * (flops_pmc if flops_pmc is not None else flops) / 1e12
                              / VALU_PEAK_TFLOPS[dtype_name],
                              "attainable_tflops": VALU_ATTAINABLE_TFLOPS.get(dtype_name),
                              "note": "flops_model = the plan compiler's operation count; flops_pmc = executed, from the "
                                      "committed SQ_INSTS_VALU_* counters; achieved_tflops / frac use flops_pmc when present"}},
        "spread": {"replays": len(reps), "ms_per_step_min": min(reps) / max(args.steps, 1) * 1e3,
                   "ms_per_step_median": elapsed / max(args.steps, 1) * 1e3, "ms_per_step_max": max(reps) / max(args.steps, 1) * 1e3,
                   "value_min": total_states * args.steps / max(reps), "value_max": total_states * args.steps / min(reps)},
        "inputs": {"distinct_states": n_distinct, "gate": "implicit clusters: max |Kd^-1 Ki| < 150, cond(Kd) < 3000 and |q_span| < 32 rad "
                                                          "(generalized_rbda_amd/states.py; profiles/r4_gate_f32_oracle.txt)" if general else None},
    }
    line.update(verify_sample(blob, q, qd, x, out, args.algo, dtype_name))
    if gather_ms is not None:
        line["gather_ms"] = gather_ms
        line["end_to_end"] = {"value": total_states * args.steps / e2e_elapsed, "unit": "evals/s",
                              "ms_per_step": e2e_elapsed / args.steps * 1e3,
                              "what": "K steps, each followed by the RCCL gather of its results to rank 0; the gather of "
                                      "step i overlaps the kernel of step i + 1"}
    if strong is not None:
        line["strong"] = strong
    if world == 1 and dist is None and not args.no_configs and args.workload == "mit_humanoid" and args.algo == "aba":
        del tq, tqd, tx, out
        line["configs"] = config_records(G, dev, max(5, min(args.steps, 20)), sha,
                                         only=set(args.configs.split(",")) if args.configs else None, headline_kernel_ms=kernel_ms)
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(blob, q, qd, x)
    elif world == 1:
        line["cpu_baseline"] = None
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
