"""GPU parity at BASELINE.json's full batch sizes (run with -m gpu on an MI355X).

Forward dynamics: EVERY state of the batch against the oracle (round 5: the multi-threaded oracle does 0.3-0.8 M states per second on the
box's sixteen usable cores, so a million states are seconds) -- fp64 at 1e-9 and the config's fp32 at 1e-3, as MAXIMA over the batch; the
inverse dynamics against the oracle on a strided sample (first tile, ragged last tile, one state from every stretch in between) and over
the whole batch through size-independent properties: ID(FD(tau)) == tau in fp64 (testRigidBodyDynamicsAlgos.cpp:221,235) and
fp32 == fp64 within the fp32 tolerance.  Tolerances: north_star (fp64 1e-6 relative -- 1e-9 used here, fp32 1e-3)."""
import os

import numpy as np
import pytest

import oracle_py as O
import generalized_rbda_amd as G
from generalized_rbda_amd.states import valid_random_states_device
from models import ROBOT_MODELS

pytestmark = pytest.mark.gpu
TOL64, TOL32 = 1e-9, 1e-3


def rel_err(a, b):
    return float((np.abs(a - b).max(axis=1) / (1.0 + np.abs(b).max(axis=1))).max())


def usable_threads():
    """hardware threads this process may really use (affinity mask cut by the cgroup quota, as bench.py counts them)"""
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


def sample_indices(B, n=2048):
    """first tile, last 101 states (covers the ragged tail), and a stride over the rest"""
    head = np.arange(min(64, B))
    tail = np.arange(max(0, B - 101), B)
    mid = np.linspace(64, max(64, B - 102), num=max(0, n - head.size - tail.size), dtype=np.int64)
    return np.unique(np.concatenate([head, mid, tail]))


def _plan_and_states(workload, B, gpu):
    if workload == "tello":
        from generalized_rbda_amd.robots import tello_with_arms

        plan = G.Plan.from_model(tello_with_arms())
        cfg = 3
    else:
        plan = G.Plan.from_urdf(os.path.join(ROBOT_MODELS, workload + ".urdf"))
        cfg = {"mit_humanoid": 2, "mini_cheetah": 1, "jvrc1_humanoid": 4, "revolute_rotor_chain": 0, "four_bar": 5, "six_bar": 6}[workload]
    # implicit-loop models: spanning positions on the constraint manifold (Newton projection on the device,
    # GenericJoint.cpp:289-385) that pass the conditioning gate of generalized_rbda_amd/states.py -- as bench.py does
    q, qd, tau, n_distinct = valid_random_states_device(plan, B, cfg, gpu)
    assert n_distinct == B  # (rejected draws are replaced by fresh draws, never by copies)
    return plan, plan.blob, q, qd, tau


CASES = [
    # (workload, batch, dtype of the BASELINE config)        BASELINE.json configs[0..4]
    ("revolute_rotor_chain", 1024, "f64"),
    ("mini_cheetah", 65536, "f64"),
    ("mit_humanoid", 262144, "f32"),
    ("mit_humanoid", 262144 + 37, "f32"),   # the same with a ragged last tile
    ("tello", 1048576, "f32"),
    ("jvrc1_humanoid", 1048576, "f32"),
    ("four_bar", 1048576, "f32"),           # config 5's loop clusters
    ("six_bar", 1048576, "f32"),
]


@pytest.mark.parametrize("workload,B,dtype", CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}" for c in CASES])
def test_full_size_batch_matches_oracle_and_properties(workload, B, dtype, gpu):
    import torch

    plan, blob, q, qd, tau = _plan_and_states(workload, B, gpu)
    t = lambda a, dt: torch.as_tensor(a, dtype=dt, device=gpu)
    idx = sample_indices(B)
    # ---- fp64 on the device: whole-batch round trip + oracle on the sample -------------------------
    q64, qd64, tau64 = t(q, torch.float64), t(qd, torch.float64), t(tau, torch.float64)
    ydd64 = plan.forward_dynamics(q64, qd64, tau64)
    back = plan.inverse_dynamics(q64, qd64, ydd64)
    torch.cuda.synchronize()
    err_rt = ((back - tau64).abs().amax(dim=1) / (1.0 + ydd64.abs().amax(dim=1))).max().item()
    assert err_rt < 1e-7, f"ID(FD(tau)) over the whole batch: {err_rt:.2e}"
    # EVERY state against the oracle (fp64)
    ref = O.forward_dynamics_mt(blob, q, qd, tau, usable_threads())
    assert rel_err(ydd64.cpu().numpy(), ref) < TOL64, "fp64 ABA vs the oracle, maximum over the whole batch"
    ref_t = O.inverse_dynamics(blob, q[idx[:512]], qd[idx[:512]], tau[idx[:512]])
    tau_gpu = plan.inverse_dynamics(q64, qd64, tau64)
    assert rel_err(tau_gpu[idx[:512]].cpu().numpy(), ref_t) < TOL64, "fp64 RNEA vs oracle sample"
    if dtype == "f64":
        return
    # ---- fp32 (the dtype of the config): every state against the device's fp64 result of the same fp32 inputs ----
    q32, qd32, tau32 = t(q, torch.float32), t(qd, torch.float32), t(tau, torch.float32)
    ydd32 = plan.forward_dynamics(q32, qd32, tau32)
    tau32_out = plan.inverse_dynamics(q32, qd32, tau32)
    ydd_ref = plan.forward_dynamics(q32.double(), qd32.double(), tau32.double())
    tau_ref = plan.inverse_dynamics(q32.double(), qd32.double(), tau32.double())
    torch.cuda.synchronize()
    e_aba = ((ydd32.double() - ydd_ref).abs().amax(dim=1) / (1.0 + ydd_ref.abs().amax(dim=1)))
    e_rnea = ((tau32_out.double() - tau_ref).abs().amax(dim=1) / (1.0 + tau_ref.abs().amax(dim=1)))
    assert e_aba.max().item() < TOL32, f"fp32 ABA vs fp64 over the whole batch: {e_aba.max().item():.2e}"
    assert e_rnea.max().item() < TOL32, f"fp32 RNEA vs fp64 over the whole batch: {e_rnea.max().item():.2e}"
    # and the oracle itself on EVERY state, fed the fp32-rounded inputs
    c = lambda a: a.astype(np.float32).astype(np.float64)
    ref32 = O.forward_dynamics_mt(blob, c(q), c(qd), c(tau), usable_threads())   # EVERY state, fed the fp32-rounded inputs
    got32 = ydd32.double().cpu().numpy()
    e = np.abs(got32 - ref32).max(axis=1) / (1.0 + np.abs(ref32).max(axis=1))
    assert e.max() < TOL32, "fp32 ABA vs the oracle, maximum over the whole batch"


def test_full_size_derivatives_jvrc1(gpu):
    """BASELINE config 5 at its full size: d ydd / d (q, qd, tau) of 1 048 576 JVRC-1 states in fp32 (three [B, 38, 38]
    arrays, the chunked analytic pipeline of deriv_kernels.hip).  A strided sample is checked against central differences
    of the ORACLE's forward dynamics along the reference's tangent step (testRigidBodyDynamicsAlgosDerivatives.cpp:309-380,
    tolerance of the fp32 path 1e-3); the whole batch through size-independent properties: d ydd / d tau = H^-1 is symmetric,
    and the sampled states give the same matrices when they are evaluated alone (chunk seams)."""
    import torch
    from generalized_rbda_amd.states import parse_clusters, random_states
    from test_gpu_parity import _reference_plus

    plan = G.Plan.from_urdf(os.path.join(ROBOT_MODELS, "jvrc1_humanoid.urdf"))
    blob, nv = plan.blob, plan.nv
    assert plan.info().analytic_derivatives
    B = 1048576
    q, qd, tau = random_states(blob, B, config_index=4)
    c32 = lambda a: a.astype(np.float32)
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    d = plan.fd_derivatives(t32(q), t32(qd), t32(tau))
    torch.cuda.synchronize()
    for k in ("dq", "dqd", "dtau"):
        assert d[k].shape == (B, nv, nv)
    asym = (d["dtau"] - d["dtau"].transpose(1, 2)).abs().amax(dim=(1, 2)) / (1.0 + d["dtau"].abs().amax(dim=(1, 2)))
    assert asym.max().item() < TOL32, f"H^-1 symmetric over the whole batch: {asym.max().item():.2e}"
    for k in ("dq", "dqd", "dtau"):
        assert torch.isfinite(d[k]).all()
    idx = np.unique(np.concatenate([[0, 63, 64], np.linspace(65, B - 66, num=252, dtype=np.int64), [B - 65, B - 1]]))  # >= 256 states
    alone = plan.fd_derivatives(t32(q[idx]), t32(qd[idx]), t32(tau[idx]))
    for k in ("dq", "dqd", "dtau"):
        a, b = d[k][torch.as_tensor(idx, device=gpu)].double(), alone[k].double()
        assert ((a - b).abs().max() / (1.0 + b.abs().max())).item() < 1e-5, f"{k}: in the batch vs alone"
    # oracle central differences on the fp32-rounded inputs
    m = parse_clusters(blob)
    qs, qds, ts = (c32(a[idx]).astype(np.float64) for a in (q, qd, tau))
    h = 1e-6
    fd = lambda qq, vv, tt: O.forward_dynamics_mt(blob, qq, vv, tt, os.cpu_count() or 1)
    n = idx.size
    ref = {k: np.empty((n, nv, nv)) for k in ("dq", "dqd", "dtau")}
    for k in range(nv):
        qp = np.stack([_reference_plus(m, qs[i], k, +h) for i in range(n)])
        qm = np.stack([_reference_plus(m, qs[i], k, -h) for i in range(n)])
        ref["dq"][:, :, k] = (fd(qp, qds, ts) - fd(qm, qds, ts)) / (2 * h)
        e = np.zeros(nv)
        e[k] = 1.0
        ref["dqd"][:, :, k] = (fd(qs, qds + e, ts) - fd(qs, qds - e, ts)) / 2.0   # exact: quadratic in qd
        ref["dtau"][:, :, k] = fd(qs, qds, ts + e) - fd(qs, qds, ts)              # exact: affine in tau
    for k in ("dq", "dqd", "dtau"):
        got = d[k][torch.as_tensor(idx, device=gpu)].double().cpu().numpy()
        err = np.abs(got - ref[k]).max(axis=(1, 2)) / (1.0 + np.abs(ref[k]).max(axis=(1, 2)))
        assert err.max() < TOL32, f"{k} vs oracle differences: {err.max():.2e}"


@pytest.mark.parametrize("workload,B", [("four_bar", 1048576), ("six_bar", 1048576), ("tello", 131072)])
def test_full_size_derivatives_on_the_constraint_manifold(workload, B, gpu):
    """BASELINE config 5's loop clusters (and Tello's differentials): d ydd / d (q, qd, tau) of the implicit models by the
    analytic route through the spanning tree (manifold_kernels.hip), fp32, at size.  A strided sample against central
    differences of the ORACLE taken ON the constraint manifold (an independent position moves, the dependent ones are
    re-projected: the reference's yardstick, testRigidBodyDynamicsAlgosDerivatives.cpp:271-383, at the fp32 tolerance);
    the whole batch: finite, H^-1 symmetric, and H^-1 times the mass matrix of the same states = 1."""
    import torch
    from generalized_rbda_amd.states import parse_clusters
    from test_gpu_parity import _reference_plus_on_manifold

    plan, blob, q, qd, tau = _plan_and_states(workload, B, gpu)
    nv = plan.nv
    assert plan.info().analytic_derivatives == 1
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    d = plan.fd_derivatives(t32(q), t32(qd), t32(tau))
    H = plan.mass_matrix(t32(q))
    torch.cuda.synchronize()
    for k in ("dq", "dqd", "dtau"):
        assert d[k].shape == (B, nv, nv) and torch.isfinite(d[k]).all()
    asym = (d["dtau"] - d["dtau"].transpose(1, 2)).abs().amax(dim=(1, 2)) / (1.0 + d["dtau"].abs().amax(dim=(1, 2)))
    assert asym.max().item() < TOL32
    eye = torch.eye(nv, device=gpu, dtype=torch.float64)
    res = (torch.bmm(d["dtau"].double(), H.double()) - eye).abs().amax(dim=(1, 2))
    assert res.max().item() < 5e-2, f"H^-1 H = 1 over the whole batch (fp32 solve and fp32 H, cond(H) up to ~1e4): {res.max().item():.2e}"
    idx = np.unique(np.concatenate([[0, 63, 64], np.linspace(65, B - 66, num=60, dtype=np.int64), [B - 65, B - 1]]))
    # (derivatives against differences along re-projected states: well-conditioned constraint Jacobians only, tests/models.py)
    kcond = O.spanning_state(blob, q[idx], qd[idx])[3]
    idx = idx[kcond < 100.0]
    assert idx.size >= 30
    m = parse_clusters(blob)
    c32 = lambda a: a.astype(np.float32).astype(np.float64)
    h = 1e-6
    worst = 0.0
    for i in idx:
        qs, qds, ts = c32(q[i]), c32(qd[i]), c32(tau[i])
        qs = O.project_positions(blob, qs[None])[0][0]   # (the fp32 rounding moved it off the manifold by ~1e-7)
        ref = np.empty((nv, nv))
        for k in range(nv):
            qp = _reference_plus_on_manifold(blob, m, qs, k, +h)[None]
            qm = _reference_plus_on_manifold(blob, m, qs, k, -h)[None]
            ref[:, k] = (O.forward_dynamics(blob, qp, qds[None], ts[None])[0] - O.forward_dynamics(blob, qm, qds[None], ts[None])[0]) / (2 * h)
        got = d["dq"][int(i)].double().cpu().numpy()
        worst = max(worst, np.abs(got - ref).max() / (1.0 + np.abs(ref).max()))
    assert worst < TOL32, f"dq vs oracle differences on the manifold: {worst:.2e}"


@pytest.mark.parametrize("name,implicit", [("parallel_chain_exp_d10_l16", False), ("parallel_chain_imp_d10_l17", True),
                                           ("three_loop_linkage", True)])  # (the last: 7 bodies, SIX constraint rows)
def test_big_cluster_models_at_size_round_trip(name, implicit, gpu):
    """The reference's own depth-10 parallel-chain files (a cluster of 16 bodies / 15 DoF, of 17 bodies / 15 DoF with a planar
    loop) through the spanning-tree route at 65 536 + 37 states: ID(FD(tau)) == tau in fp64 over the whole batch, fp32
    against fp64, and the oracle built for 48 bodies per cluster on a strided sample."""
    import torch
    from models import valid_states

    plan = G.Plan.from_urdf(os.path.join(ROBOT_MODELS, name + ".urdf"))
    assert plan.info().spanning_tree_route == 1
    blob = plan.blob
    B = 65536 + 37
    base = valid_states(blob, 4096, config_index=9, big=True, scale=0.5, max_cond=50 if implicit else None)
    rng = np.random.default_rng(5)
    pick = rng.integers(0, 4096, B)
    # distinct states: the drawn positions with fresh velocities and torques per state
    q = base[0][pick]
    qd, tau = rng.uniform(-1, 1, (B, plan.nv)), rng.uniform(-1, 1, (B, plan.nv))
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=gpu)
    q64, qd64, tau64 = t(q, torch.float64), t(qd, torch.float64), t(tau, torch.float64)
    ydd = plan.forward_dynamics(q64, qd64, tau64)
    back = plan.inverse_dynamics(q64, qd64, ydd)
    scale = 1.0 + float(ydd.abs().max())
    assert float((back - tau64).abs().max()) / scale < 1e-9
    ydd32 = plan.forward_dynamics(t(q, torch.float32), t(qd, torch.float32), t(tau, torch.float32)).double()
    assert float(((ydd32 - ydd).abs().amax(dim=1) / (1.0 + ydd.abs().amax(dim=1))).max()) < TOL32
    idx = sample_indices(B, 512)
    ref = O.forward_dynamics(blob, q[idx], qd[idx], tau[idx], big=True)
    assert rel_err(ydd.cpu().numpy()[idx], ref) < TOL64
