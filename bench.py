#!/usr/bin/env python3
"""bench.py -- forward-dynamics evals/sec on batched random states, MIT Humanoid cluster model.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (batched cluster ABA, ClusterTreeModel::forwardDynamics)
over one batch of B synthetic states per GPU, inputs and outputs resident in HBM.  Weak scaling:
every rank owns its own B-state shard (states are independent, the model plan is replicated);
there is no data-path collective inside a step -- results are gathered to rank 0 over RCCL once
after the timed region (reported as gather_ms, not part of `value`).
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

WORKLOADS = {
    # name: (urdf, batch per GPU, dtype, config index for the RNG seed)
    "mit_humanoid": ("mit_humanoid.urdf", 262144, "f32", 2),
    "mini_cheetah": ("mini_cheetah.urdf", 65536, "f64", 1),
    "revolute_rotor_chain": ("revolute_rotor_chain.urdf", 1024, "f64", 0),
    "jvrc1_humanoid": ("jvrc1_humanoid.urdf", 1048576, "f32", 4),
    # hand-built TelloWithArms (the URDF carries no constraints): implicit differentials
    "tello": ("<TelloWithArms>", 1048576, "f32", 3),
}


def cpu_baseline(blob, q, qd, tau, budget_s=12.0):
    """Oracle (CPU restatement of the reference algorithm) on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O

    cores = os.cpu_count() or 1
    n = min(q.shape[0], 2048)
    t0 = time.perf_counter()
    O.forward_dynamics_mt(blob, q[:n], qd[:n], tau[:n], cores)
    dt = max(time.perf_counter() - t0, 1e-6)
    n2 = int(min(q.shape[0], max(n, n / dt * budget_s)))
    t0 = time.perf_counter()
    O.forward_dynamics_mt(blob, q[:n2], qd[:n2], tau[:n2], cores)
    dt = time.perf_counter() - t0
    return {"value": n2 / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"first {n2} states of the same batch, fp64 dense cluster-ABA restatement (oracle/), "
                      f"{cores} pthreads, one pass"}


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
    args = ap.parse_args()

    import numpy as np
    import torch

    import generalized_rbda_amd as G
    from generalized_rbda_amd.states import random_states

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
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
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    urdf, B, dtype_name, cfg = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    if args.dtype:
        dtype_name = args.dtype
    tdt = torch.float32 if dtype_name == "f32" else torch.float64
    if args.workload == "tello":
        from generalized_rbda_amd.robots import tello_with_arms

        plan = G.Plan.from_model(tello_with_arms())
    else:
        plan = G.Plan.from_urdf(os.path.join(ROOT, "tests", "golden", "robot-models", urdf))
    blob = plan.blob
    info = plan.info()

    # synthetic inputs: reference sampling law, counter-based RNG, distinct stream per rank
    q, qd, x = random_states(blob, B, config_index=cfg + 1000 * rank)
    if args.workload == "tello":
        # implicit clusters take spanning positions on the constraint manifold: Newton projection of the
        # dependent coordinates (GenericJoint.cpp:289-385) on the device (grbda_project_positions_f64) --
        # input generation only; states that do not converge are replaced by converged ones
        t64 = torch.as_tensor(q, dtype=torch.float64, device=dev)
        ok = plan.project_positions(t64).cpu().numpy()
        q = t64.cpu().numpy()
        good = np.flatnonzero(ok)
        if good.size == 0:
            raise SystemExit("no valid Tello state could be generated")
        bad = np.flatnonzero(~ok)
        q[bad] = q[good[np.arange(bad.size) % good.size]]
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
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run(tq, tqd, tx, out=out)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # kernel-only duration: hipEvents on the launch stream (torch's current stream here)
    kernel_ms = plan.time_kernel(args.algo, tq, tqd, tx, out, iters=max(5, min(args.steps, 50)))

    # results gathered to rank 0 over RCCL (outside the timed region)
    gather_ms = None
    if dist is not None:
        bufs = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
        barrier()
        g0 = time.perf_counter()
        dist.gather(out, bufs, dst=0)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    elem = 4 if dtype_name == "f32" else 8
    bytes_per_eval = (plan.nq + 3 * plan.nv) * elem  # q, qd, tau in + ydd out
    evals_per_s = world * B * args.steps / elapsed
    kernel_evals_per_s = B / (kernel_ms * 1e-3)
    achieved_gbs = kernel_evals_per_s * bytes_per_eval / 1e9
    flops = info.flops_aba if args.algo == "aba" else info.flops_rnea
    # measured HBM-side traffic of the same launch configuration, if a PMC run is committed (profiles/)
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
            for e in json.load(f)["entries"]:
                if (e["workload"], e["algo"], e["dtype"], e["batch"]) == (args.workload, args.algo, dtype_name, B):
                    traffic = e["bytes_per_launch"]
                    traffic_src = "rocprofv3 FETCH_SIZE + WRITE_SIZE per launch, profiles/r1_pmc_traffic.json"
    except (OSError, KeyError, ValueError):
        pass
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
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype_name,
        "data": "synthetic",
        "config": {"workload": f"{urdf} cluster-{'ABA' if args.algo == 'aba' else 'RNEA'}, {B} random states per GPU",
                   "batch_per_gpu": B, "nq": plan.nq, "nv": plan.nv, "n_bodies": plan.n_bodies,
                   "n_clusters": plan.n_clusters, "parallelism": f"batch-sharded x{world}, plan replicated"},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": bytes_per_eval * B,
                     "kernel": f"{args.algo}_kernel<{'float' if dtype_name == 'f32' else 'double'}, false>",
                     "kernel_ms": kernel_ms, "bytes_per_eval": bytes_per_eval,
                     "note": "algorithmic bytes; the path is VALU-bound (see valu)",
                     "valu": {"flops_per_eval": flops,
                              "achieved_tflops": kernel_evals_per_s * flops / 1e12,
                              "peak_tflops": VALU_PEAK_TFLOPS[dtype_name],
                              "frac": kernel_evals_per_s * flops / 1e12 / VALU_PEAK_TFLOPS[dtype_name]}},
    }
    if gather_ms is not None:
        line["gather_ms"] = gather_ms
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(blob, q, qd, x)
    elif world == 1:
        line["cpu_baseline"] = None
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
