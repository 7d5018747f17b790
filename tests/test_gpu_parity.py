"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on identical seeded inputs and against the committed golden vectors.

Tolerances (BASELINE.json north_star): fp64 <= 1e-6 max relative error, fp32 <= 1e-3.
The fp64 bound used here is much tighter (1e-9) because both sides are fp64 and differ only in
operation order; fp32 errors are normalised by the per-state vector norm as the reference's own
comparisons are (testRigidBodyDynamicsAlgos.cpp:9,208-232 use norms of differences)."""
import json
import os
import sys

import numpy as np
import pytest

import oracle_py as O
import generalized_rbda_amd as G
from generalized_rbda_amd import modeldesc as md
from generalized_rbda_amd.states import random_states
from models import ROBOT_MODELS, random_inertia, random_xtree, valid_states, zoo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL64 = 1e-9
TOL32 = 1e-3


def rel_err(a, b):
    """max over states of ||a - b||_inf / (1 + ||b||_inf)"""
    return float((np.abs(a - b).max(axis=1) / (1.0 + np.abs(b).max(axis=1))).max())


def run_gpu(plan, which, q, qd, x, dtype, gpu):
    import torch

    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=gpu)
    fn = plan.forward_dynamics if which == "aba" else plan.inverse_dynamics
    out = fn(t(q), t(qd), t(x))
    torch.cuda.synchronize()
    return out.double().cpu().numpy()


@pytest.mark.parametrize("name,blob", list(zoo().items()), ids=list(zoo().keys()))
def test_aba_and_rnea_fp64_match_oracle(name, blob, gpu):
    import torch

    plan = G.Plan(blob)
    for B in (1, 63, 64, 200):
        q, qd, tau = valid_states(blob, B, config_index=21)
        ref = O.forward_dynamics(blob, q, qd, tau)
        got = run_gpu(plan, "aba", q, qd, tau, torch.float64, gpu)
        assert rel_err(got, ref) < TOL64, f"ABA B={B}"
        ref_t = O.inverse_dynamics(blob, q, qd, tau)
        got_t = run_gpu(plan, "rnea", q, qd, tau, torch.float64, gpu)
        assert rel_err(got_t, ref_t) < TOL64, f"RNEA B={B}"


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_mit_humanoid", "urdf_jvrc1_humanoid", "tello_with_arms", "urdf_six_bar", "rev_rotor_chain_4"])
def test_componentwise_parity_fp64(name, gpu):
    """The other tests of this file measure a state's error norm-wise (rel_err: the largest component error against the largest component).
    Here every COMPONENT of the forward / inverse dynamics is held against the oracle on its own: |a_i - b_i| <= 1e-9 (|b_i| + 1e-3 |b|_inf)
    -- small accelerations next to large ones (a distal joint against the base) keep six digits of their own, not only of the state's norm.
    fp64, 2 000 states per model (both chain kernels and latency mode: the batch is below one tile per SIMD); the floor 1e-3 |b|_inf is what
    cancellation in a sum of terms of the state's scale leaves to a component that happens to be near zero."""
    import torch

    blob = zoo()[name]
    plan = G.Plan(blob)
    q, qd, tau = valid_states(blob, 2000, config_index=63)
    for which, fn in (("aba", O.forward_dynamics), ("rnea", O.inverse_dynamics)):
        ref = fn(blob, q, qd, tau)
        got = run_gpu(plan, which, q, qd, tau, torch.float64, gpu)
        bound = 1e-9 * (np.abs(ref) + 1e-3 * np.abs(ref).max(axis=1, keepdims=True))
        worst = float((np.abs(got - ref) / bound).max())
        assert worst < 1.0, f"{which}: worst component at {worst:.2f} of its bound"


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_mit_humanoid", "tree_mixed_float", "urdf_jvrc1_humanoid"])
@pytest.mark.parametrize("scale", [0.7, 1.3])
def test_non_unit_quaternion_is_used_as_it_is(name, scale, gpu):
    """The reference does not normalise the base quaternion: quaternionToRotationMatrix (OrientationTools.h:251-269) builds its
    matrix from whatever four numbers the state holds, and a quaternion of length 1.3 gives a scaled, non-orthogonal "rotation" that
    the dynamics then carry through.  Parity means doing the same: forward / inverse dynamics (and the mass matrix) of states
    whose quaternion is scaled, against the oracle, in both precisions -- every kernel that builds the base rotation (chain
    programs, latency mode at this batch size, interpreter) follows the formula, none normalises."""
    import torch

    blob = zoo()[name]
    plan = G.Plan(blob)
    q, qd, tau = valid_states(blob, 300, config_index=61)
    q = q.copy()
    q[:, 3:7] *= scale   # (floating base first: position 3, quaternion 4)
    assert abs(np.linalg.norm(q[0, 3:7]) - scale) < 1e-9
    ref, ref_t = O.forward_dynamics(blob, q, qd, tau), O.inverse_dynamics(blob, q, qd, tau)
    assert rel_err(run_gpu(plan, "aba", q, qd, tau, torch.float64, gpu), ref) < TOL64
    assert rel_err(run_gpu(plan, "rnea", q, qd, tau, torch.float64, gpu), ref_t) < TOL64
    c32 = lambda a: a.astype(np.float32).astype(np.float64)
    q32, qd32, tau32 = c32(q), c32(qd), c32(tau)
    assert rel_err(run_gpu(plan, "aba", q32, qd32, tau32, torch.float32, gpu), O.forward_dynamics(blob, q32, qd32, tau32)) < TOL32
    assert rel_err(run_gpu(plan, "rnea", q32, qd32, tau32, torch.float32, gpu), O.inverse_dynamics(blob, q32, qd32, tau32)) < TOL32
    # and the unit-quaternion result is NOT what comes out (the scaling is not silently removed)
    qn = q.copy()
    qn[:, 3:7] /= scale
    assert rel_err(O.forward_dynamics(blob, qn, qd, tau), ref) > 1e-3


@pytest.mark.parametrize("name", ["urdf_mit_humanoid", "urdf_jvrc1_humanoid", "rev_rotor_chain_4", "tree_triple_fixed"])
def test_fp32_joint_angles_of_a_hundred_radians(name, gpu):
    """Explicit models with joint angles drawn from +-100 rad (wound joints).  fp64 keeps 1e-9.  In fp32 the yardstick is the
    oracle compiled in `float` (the dense restatement of the reference, oracle/_build/libgrbda_oracle_f32.so) on the same
    fp32-rounded inputs: a single-precision angle of magnitude a carries ~6e-8 a of absolute error through any sine, and a
    NON-axisymmetric geared rotor multiplies the joint angle by its gear ratio first (RevoluteTripleWithRotor here: up to 8 x 100
    rad, where one ulp of the product is 6e-5 rad) -- there single precision itself leaves 1e-3 (measured: the float restatement
    1.4e-3, the kernels 5.4e-3 on the worst of 2 000 states), and the kernels must stay within 5 x the float restatement's own error;
    everywhere else they stay below the path's 1e-3 outright."""
    import torch
    from generalized_rbda_amd.states import parse_clusters

    blob = zoo()[name]
    plan = G.Plan(blob)
    q, qd, tau = valid_states(blob, 2000, config_index=62)
    rng = np.random.default_rng(62)
    q = q.copy()
    for c in parse_clusters(blob)["clusters"]:
        (pc, fb, k, qi, npos, vi, nvel, nsp, nsv, ctype, *_rest) = c
        if ctype != 1:   # (not the free base)
            q[:, qi: qi + npos] = rng.uniform(-100.0, 100.0, (q.shape[0], npos))
    c32 = lambda a: a.astype(np.float32).astype(np.float64)
    q32, qd32, tau32 = c32(q), c32(qd), c32(tau)
    ref, ref_t = O.forward_dynamics(blob, q32, qd32, tau32), O.inverse_dynamics(blob, q32, qd32, tau32)
    assert rel_err(run_gpu(plan, "aba", q32, qd32, tau32, torch.float64, gpu), ref) < TOL64
    assert rel_err(run_gpu(plan, "rnea", q32, qd32, tau32, torch.float64, gpu), ref_t) < TOL64
    e_float = rel_err(O.forward_dynamics_mt_f32(blob, q32, qd32, tau32, 4), ref)
    e_aba = rel_err(run_gpu(plan, "aba", q32, qd32, tau32, torch.float32, gpu), ref)
    e_rnea = rel_err(run_gpu(plan, "rnea", q32, qd32, tau32, torch.float32, gpu), ref_t)
    assert e_aba < max(TOL32, 5.0 * e_float), (e_aba, e_float)
    assert e_rnea < max(TOL32, 5.0 * e_float), (e_rnea, e_float)   # (inverse dynamics of the triple clusters: 1.5e-3 measured)
    if name != "tree_triple_fixed":
        assert e_aba < TOL32 and e_rnea < TOL32, (e_aba, e_rnea, e_float)


@pytest.mark.parametrize("name,blob", list(zoo().items()), ids=list(zoo().keys()))
def test_aba_and_rnea_fp32_match_oracle(name, blob, gpu):
    import torch

    plan = G.Plan(blob)
    q, qd, tau = valid_states(blob, 500, config_index=22)
    q32, qd32, tau32 = (a.astype(np.float32).astype(np.float64) for a in (q, qd, tau))
    ref = O.forward_dynamics(blob, q32, qd32, tau32)
    got = run_gpu(plan, "aba", q32, qd32, tau32, torch.float32, gpu)
    assert rel_err(got, ref) < TOL32
    ref_t = O.inverse_dynamics(blob, q32, qd32, tau32)
    got_t = run_gpu(plan, "rnea", q32, qd32, tau32, torch.float32, gpu)
    assert rel_err(got_t, ref_t) < TOL32


def test_golden_vectors_from_reference_codegen(gpu):
    import torch

    with open(os.path.join(HERE, "golden", "codegen_vectors.json")) as f:
        cases = json.load(f)["cases"]
    unhex = lambda a: np.array([[float.fromhex(v) for v in row] for row in a])
    for c in cases:
        build = md.revolute_chain_with_rotor if c["family"] == "rev" else md.revolute_pair_chain_with_rotor
        plan = G.Plan(build(c["n"]).serialize())
        y, yd, x = unhex(c["y"]), unhex(c["yd"]), unhex(c["x"])
        fd = run_gpu(plan, "aba", y, yd, x, torch.float64, gpu)
        idd = run_gpu(plan, "rnea", y, yd, x, torch.float64, gpu)
        assert rel_err(fd, unhex(c["FD"])) < 1e-8, (c["family"], c["n"])
        assert rel_err(idd, unhex(c["ID"])) < 1e-9, (c["family"], c["n"])


def test_id_of_fd_roundtrip_large_batch(gpu):
    """size-independent property at a large batch: ID(FD(tau)) == tau, all on the GPU."""
    import torch

    blob = zoo()["tree_mixed_float"]
    plan = G.Plan(blob)
    B = 100_000
    q, qd, tau = valid_states(blob, B, config_index=23)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64, device=gpu)
    tq, tqd, ttau = t(q), t(qd), t(tau)
    ydd = plan.forward_dynamics(tq, tqd, ttau)
    back = plan.inverse_dynamics(tq, tqd, ydd)
    torch.cuda.synchronize()
    err = (back - ttau).abs().max().item() / (1 + ydd.abs().max().item())
    assert err < 1e-8
    # the ragged tail and the grid-stride loop must not depend on the batch size
    ydd_small = plan.forward_dynamics(tq[:777], tqd[:777], ttau[:777])
    torch.cuda.synchronize()
    assert torch.equal(ydd_small, ydd[:777])


def test_host_convenience_entry_points(gpu):
    blob = zoo()["rev_pair_rotor_chain_4"]
    plan = G.Plan(blob)
    q, qd, tau = valid_states(blob, 5, config_index=24)
    assert rel_err(plan.forward_dynamics_host(q, qd, tau), O.forward_dynamics(blob, q, qd, tau)) < TOL64
    assert rel_err(plan.inverse_dynamics_host(q, qd, tau), O.inverse_dynamics(blob, q, qd, tau)) < TOL64


def test_lds_budget_does_not_change_results(gpu, monkeypatch):
    """slots in LDS vs. in the global slab are the same numbers"""
    import torch

    blob = zoo()["tree_mixed_float"]
    q, qd, tau = valid_states(blob, 300, config_index=25)
    outs = []
    monkeypatch.setenv("GRBDA_NO_CHAIN", "1")  # the interpreter's slot store is what has the two homes
    for lds in ("0", "4096", "65536"):
        monkeypatch.setenv("GRBDA_LDS_BYTES_PER_WAVE", lds)
        outs.append(run_gpu(G.Plan(blob), "aba", q, qd, tau, torch.float64, gpu))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("name", ["tree_mixed_float", "tree_mixed_fixed", "urdf_mit_humanoid", "urdf_six_bar", "rev_pair_rotor_chain_4"])
def test_external_forces_match_oracle(name, gpu):
    """TreeModel::setExternalForces semantics (world-frame spatial force per body, TreeModel.cpp:214-239);
    the reference's main dynamics test applies a random force to every body
    (testRigidBodyDynamicsAlgos.cpp:189-204)."""
    import torch
    from generalized_rbda_amd.states import parse_clusters

    blob = zoo()[name]
    plan = G.Plan(blob)
    nb = parse_clusters(blob)["nb"]
    q, qd, tau = valid_states(blob, 130, config_index=26)
    fext = np.random.default_rng(5).uniform(-1, 1, (q.shape[0], nb, 6))
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=gpu)
    for dt, tol in ((torch.float64, TOL64), (torch.float32, TOL32)):
        c = lambda a: a.astype(np.float32).astype(np.float64) if dt == torch.float32 else a
        ref = O.forward_dynamics(blob, c(q), c(qd), c(tau), c(fext))
        got = plan.forward_dynamics(t(q, dt), t(qd, dt), t(tau, dt), f_ext=t(fext, dt))
        ref_t = O.inverse_dynamics(blob, c(q), c(qd), c(tau), c(fext))
        got_t = plan.inverse_dynamics(t(q, dt), t(qd, dt), t(tau, dt), f_ext=t(fext, dt))
        torch.cuda.synchronize()
        assert rel_err(got.double().cpu().numpy(), ref) < tol, f"ABA {dt}"
        assert rel_err(got_t.double().cpu().numpy(), ref_t) < tol, f"RNEA {dt}"
    # forces must matter, and the host entry point takes them too
    assert rel_err(O.forward_dynamics(blob, q, qd, tau), O.forward_dynamics(blob, q, qd, tau, fext)) > 1e-3
    assert rel_err(plan.forward_dynamics_host(q, qd, tau, f_ext=fext), O.forward_dynamics(blob, q, qd, tau, fext)) < TOL64


# ---- derived quantities (include/grbda_hip.h: bias force, mass matrix, d ydd/d tau, d ydd/d qd) ----------
def _oracle_columns(fn, blob, q, qd, x, which, step):
    """(fn(.. + step e_j) - fn(.. - step e_j)) / (2 step) for every j, perturbing argument `which` (1 qd, 2 x)."""
    B, nv = x.shape
    cols = np.empty((B, nv, nv))
    for j in range(nv):
        d = np.zeros_like(x)
        d[:, j] = step
        args_p = [q, qd + d, x] if which == 1 else [q, qd, x + d]
        args_m = [q, qd - d, x] if which == 1 else [q, qd, x - d]
        cols[:, :, j] = (fn(blob, *args_p) - fn(blob, *args_m)) / (2 * step)
    return cols


@pytest.mark.parametrize("name,blob", list(zoo().items()), ids=list(zoo().keys()))
def test_mass_matrix_bias_and_fd_derivatives_match_oracle(name, blob, gpu):
    """getMassMatrix / getBiasForceVector (ClusterTreeModel.cpp:99-110) and the derivative identities the
    reference tests (testRigidBodyDynamicsAlgosDerivatives.cpp:309-380; d ydd/d tau = H^-1)."""
    import torch

    plan = G.Plan(blob)
    B = 3
    q, qd, tau = valid_states(blob, B, config_index=23)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    nv = plan.nv
    zero = np.zeros_like(qd)
    H = plan.mass_matrix(t(q)).cpu().numpy()
    C = plan.bias_force(t(q), t(qd)).cpu().numpy()
    Hinv = plan.fd_dtau(t(q)).cpu().numpy()
    J = plan.fd_dqd(t(q), t(qd), t(tau)).cpu().numpy()
    # the oracle is affine in ydd / tau and quadratic in qd too: unit central differences are exact there as well
    H_ref = _oracle_columns(O.inverse_dynamics, blob, q, zero, zero, 2, 1.0)
    C_ref = O.inverse_dynamics(blob, q, qd, zero)
    Hinv_ref = _oracle_columns(O.forward_dynamics, blob, q, zero, zero, 2, 1.0)
    J_ref = _oracle_columns(O.forward_dynamics, blob, q, qd, tau, 1, 1.0)
    scale = lambda M: 1.0 + np.abs(M).max()
    assert np.abs(H - H_ref).max() / scale(H_ref) < TOL64
    assert np.abs(H - H.transpose(0, 2, 1)).max() / scale(H_ref) < TOL64
    # fp32: explicit models write packed rows of the lower triangle and unpack them in place (crba_kernels.hip); 67 states:
    # a full tile and a ragged one
    q67 = valid_states(blob, 67, config_index=24)[0]
    H32 = plan.mass_matrix(torch.as_tensor(np.ascontiguousarray(q67), dtype=torch.float32, device=gpu)).double().cpu().numpy()
    H67 = plan.mass_matrix(t(q67.astype(np.float32).astype(np.float64))).cpu().numpy()
    assert np.isfinite(H32).all() and np.abs(H32 - H67).max() / scale(H67) < TOL32
    if plan.info().analytic_derivatives:  # (models with implicit loops go through nv + 1 inverse-dynamics evaluations)
        assert (H32 == H32.transpose(0, 2, 1)).all() and ((H32 == 0) == (H67 == 0)).all()
    assert rel_err(C, C_ref) < TOL64
    assert np.abs(Hinv - Hinv_ref).max() / scale(Hinv_ref) < 1e-8
    assert np.abs(J - J_ref).max() / scale(J_ref) < 1e-8
    # H ydd + C = tau with ydd from the forward dynamics; H^-1 H = 1
    ydd = run_gpu(plan, "aba", q, qd, tau, torch.float64, gpu)
    assert rel_err(np.einsum("bij,bj->bi", H, ydd) + C, tau) < 1e-8
    eye = np.einsum("bij,bjk->bik", Hinv, H)
    assert np.abs(eye - np.eye(nv)).max() < 1e-7
    # fp32 entry points
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    H32 = plan.mass_matrix(t32(q)).double().cpu().numpy()
    assert np.abs(H32 - H_ref).max() / scale(H_ref) < TOL32
    C32 = plan.bias_force(t32(q), t32(qd)).double().cpu().numpy()
    assert rel_err(C32, C_ref) < TOL32


def test_derived_quantities_chunked_large_batch(gpu):
    """The expanded batch is processed in chunks below 256 MiB: a batch that needs several chunks gives the
    same matrices as a small one."""
    import torch

    blob = zoo()["urdf_mini_cheetah"]
    plan = G.Plan(blob)
    B = 40000  # (nv + 1) * B rows of fp64 exceed one chunk
    q, qd, tau = random_states(blob, B, 5)
    tq = torch.as_tensor(q, dtype=torch.float64, device=gpu)
    H = plan.mass_matrix(tq)
    idx = torch.tensor([0, 17, B // 2, B - 1], device=gpu)
    H_small = plan.mass_matrix(tq[idx].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(H[idx], H_small)


def test_derivative_entry_points_edge_cases(gpu):
    """One state; an empty batch; a model with more coordinates than the SPD solve's 64 lanes (nv = 66: the plan reports
    no analytic route and the difference batches serve it) -- every case through grbda_fd_derivatives_*."""
    import torch
    from models import random_cluster_tree

    blob = zoo()["urdf_mit_humanoid"]
    plan = G.Plan(blob)
    q, qd, tau = random_states(blob, 1, 8)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    d1 = plan.fd_derivatives(t(q), t(qd), t(tau))
    Hinv = d1["dtau"][0].cpu().numpy()
    H = plan.mass_matrix(t(q))[0].cpu().numpy()
    assert np.abs(Hinv @ H - np.eye(plan.nv)).max() < 1e-8
    d0 = plan.fd_derivatives(t(q[:0]), t(qd[:0]), t(tau[:0]))
    assert all(v.shape == (0, plan.nv, plan.nv) for v in d0.values())
    big = random_cluster_tree(301, n_clusters=60, floating=True, kinds=("rev", "rotor")).serialize()
    pb = G.Plan(big)
    assert pb.nv > 64 and pb.info().analytic_derivatives == 0
    q, qd, tau = random_states(big, 2, 9)
    db = pb.fd_derivatives(t(q), t(qd), t(tau), want=("dtau", "dqd"))
    Hb = pb.mass_matrix(t(q)).cpu().numpy()
    assert np.abs(np.einsum("bij,bjk->bik", db["dtau"].cpu().numpy(), Hb) - np.eye(pb.nv)).max() < 1e-7
    k = pb.nv - 1
    e = np.zeros_like(qd)
    e[:, k] = 1.0
    col = (O.forward_dynamics(big, q, qd + e, tau) - O.forward_dynamics(big, q, qd - e, tau)) / 2
    assert np.abs(db["dqd"][:, :, k].cpu().numpy() - col).max() / (1 + np.abs(col).max()) < 1e-8


def test_analytic_derivatives_chunked_large_batch(gpu):
    """The analytic derivative pipeline works through the batch in chunks of its 512 MiB work space (H, dID/dq, dID/dqd,
    ydd per state): a batch that needs several chunks gives, at the chunk seams and the ragged end, the matrices of a small
    batch of the same states."""
    import torch

    blob = zoo()["urdf_mini_cheetah"]
    plan = G.Plan(blob)
    assert plan.info().analytic_derivatives == 1
    nv = plan.nv
    per_state = (3 * nv * nv + nv) * 4
    chunk = (512 << 20) // per_state
    B = 2 * chunk + 77
    q, qd, tau = random_states(blob, B, 6)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=gpu)
    tq, tqd, tt = t(q), t(qd), t(tau)
    d = plan.fd_derivatives(tq, tqd, tt, want=("dq", "dtau"))
    idx = torch.tensor([0, 63, chunk - 1, chunk, chunk + 1, 2 * chunk - 1, 2 * chunk, B - 1], device=gpu)
    small = plan.fd_derivatives(tq[idx].contiguous(), tqd[idx].contiguous(), tt[idx].contiguous(), want=("dq", "dtau"))
    torch.cuda.synchronize()
    # (equal up to rounding, not bit for bit: the small batch's forward dynamics run in latency mode, which sums the limbs'
    # inertias at the base in another order)
    for k in ("dq", "dtau"):
        a, b = d[k][idx].double(), small[k].double()
        assert ((a - b).abs().max() / (1.0 + b.abs().max())).item() < 1e-5, k


# ---- steps either side of the path: Newton projection, spanning recovery ---------------------------------
def _implicit_models():
    from generalized_rbda_amd.states import parse_clusters

    out = []
    for name, blob in zoo().items():
        if any(c[9] in (2, 3) for c in parse_clusters(blob)["clusters"]):
            out.append((name, blob))
    return out


@pytest.mark.parametrize("name,blob", _implicit_models(), ids=[n for n, _ in _implicit_models()])
def test_newton_projection_matches_oracle(name, blob, gpu):
    """grbda_project_positions against the oracle's Newton iteration (GenericJoint.cpp:289-385): same
    converged/failed verdicts, same projected coordinates, phi = 0 at the result."""
    import torch
    from generalized_rbda_amd.states import parse_clusters

    plan = G.Plan(blob)
    q, qd, _ = random_states(blob, 300, config_index=31)
    q_ref, ok_ref = O.project_positions(blob, q)
    tq = torch.as_tensor(q, dtype=torch.float64, device=gpu)
    ok = plan.project_positions(tq).cpu().numpy()
    got = tq.cpu().numpy()
    assert ok_ref.any(), "no state converged in the oracle: useless test"
    assert (ok == ok_ref).mean() > 0.98  # borderline states may flip with the rounding of the iteration
    both = ok & ok_ref
    # Newton far from a root is chaotic: a few states end on another (equally valid) branch of the linkage
    # when the iteration is rounded differently; the converged ones are all checked against phi = 0 below
    same = np.abs(got[both] - q_ref[both]).max(axis=1) < 1e-7
    assert same.mean() > 0.95
    m = parse_clusters(blob)
    for ci, c in enumerate(m["clusters"]):
        if c[9] not in (2, 3):
            continue
        (pc, fb, k, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) = c
        for b in np.flatnonzero(both)[:20]:
            phi = O.cluster_constraint(blob, ci, got[b], np.zeros(m["nv"]), nsv, nvel, rows)[4]
            assert np.abs(phi).max() < 1e-8
    # a model without implicit clusters is left untouched
    plain = zoo()["urdf_mini_cheetah"]
    p2 = G.Plan(plain)
    q2, _, _ = random_states(plain, 70, 3)
    t2 = torch.as_tensor(q2, dtype=torch.float64, device=gpu)
    assert bool(p2.project_positions(t2).all()) and torch.equal(t2.cpu(), torch.as_tensor(q2))


@pytest.mark.parametrize("name,blob", list(zoo().items()), ids=list(zoo().keys()))
def test_spanning_recovery_matches_oracle(name, blob, gpu):
    """qd_span = G yd, qdd_span = G ydd + g (ClusterJoint.cpp:55-58, GenericJoint.cpp:57-90)."""
    import torch
    from generalized_rbda_amd.states import parse_clusters

    plan = G.Plan(blob)
    B = 70
    q, qd, ydd = valid_states(blob, B, config_index=32)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    v, a = plan.spanning(t(q), t(qd), t(ydd))
    v, a = v.cpu().numpy(), a.cpu().numpy()
    m = parse_clusters(blob)
    dbls_off = 96 + 416 * m["nb"] + 64 * m["nc"] + 4 * (len(m["ints"]) + (len(m["ints"]) & 1))
    at = 0
    for ci, c in enumerate(m["clusters"]):
        (pc, fb, k, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) = c
        if ctype == 1:  # free
            assert np.array_equal(v[:, at:at + 6], qd[:, vi:vi + 6]) and np.array_equal(a[:, at:at + 6], ydd[:, vi:vi + 6])
            at += 6
            continue
        for b in range(B):
            if ctype == 0:
                Gm = np.frombuffer(blob, dtype="<f8", count=nsv * nvel, offset=dbls_off + 8 * do).reshape(nsv, nvel)
                g = np.zeros(nsv)
            else:
                Gm, g = O.cluster_constraint(blob, ci, q[b], qd[b], nsv, nvel, rows)[:2]
            assert np.abs(v[b, at:at + nsv] - Gm @ qd[b, vi:vi + nvel]).max() < 1e-9
            assert np.abs(a[b, at:at + nsv] - (Gm @ ydd[b, vi:vi + nvel] + g)).max() < 1e-8 * (1 + np.abs(g).max())
        at += nsv
    assert at == plan.n_span_vel


def _reference_plus(m, q, k, d):
    """TestHelpers::plus (UnitTests/testHelpers.hpp:50-112): state q after the tangent step d along velocity
    coordinate k.  Free base: positions [pos 3, quat 4 scalar first], velocities [angular 3, linear 3]."""
    q = q.copy()
    for c in m["clusters"]:
        (pc, fb, kk, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) = c
        if not (vi <= k < vi + nvel):
            continue
        a = k - vi
        if ctype == 1:
            quat = q[qi + 3: qi + 7].copy()
            e0, e1, e2, e3 = quat
            M = np.array([[1 - 2 * (e2 * e2 + e3 * e3), 2 * (e1 * e2 - e0 * e3), 2 * (e1 * e3 + e0 * e2)],
                          [2 * (e1 * e2 + e0 * e3), 1 - 2 * (e1 * e1 + e3 * e3), 2 * (e2 * e3 - e0 * e1)],
                          [2 * (e1 * e3 - e0 * e2), 2 * (e2 * e3 + e0 * e1), 1 - 2 * (e1 * e1 + e2 * e2)]])
            R = M.T  # quaternionToRotationMatrix transposes (OrientationTools.h:251-269)
            if a < 3:
                dv = np.zeros(3)
                dv[a] = d
                w, v = quat[0], quat[1:]
                prod = np.concatenate([[-v @ dv], w * dv + np.cross(v, dv)])  # quat x (0, dv)
                q[qi + 3: qi + 7] = quat + 0.5 * prod
            else:
                dp = np.zeros(3)
                dp[a - 3] = d
                q[qi: qi + 3] += R.T @ dp
        else:
            q[qi + a] += d
    return q


def _reference_plus_on_manifold(blob, m, q, k, d):
    """_reference_plus, then -- for implicit-loop clusters, whose velocity coordinate k is the k-th INDEPENDENT spanning
    position -- the dependent positions put back on phi(q) = 0 with the oracle's Newton projection."""
    q = q.copy()
    for c in m["clusters"]:
        (pc, fb, kk, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) = c
        if ctype in (2, 3) and vi <= k < vi + nvel:
            flags = m["ints"][io + 1: io + 1 + nsv] if ctype == 2 else m["ints"][io: io + nsv]
            ind = [j for j in range(nsv) if flags[j]]
            q[qi + ind[k - vi]] += d
            qp, ok = O.project_positions(blob, q[None])
            assert ok[0]
            return qp[0]
    if m["ori"] == 1:  # roll-pitch-yaw base: plain q + dq (testHelpers.hpp:49-74 special-cases only the quaternion)
        for c in m["clusters"]:
            if c[9] == 1 and c[5] <= k < c[5] + 6:
                q[c[3] + (k - c[5])] += d
                return q
    return _reference_plus(m, q, k, d)


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_mit_humanoid", "rev_rotor_chain_4", "tree_mixed_float", "tree_triple_fixed",
                                  "urdf_four_bar", "urdf_six_bar", "urdf_planar_leg_linkage", "tello_with_arms", "urdf_mini_cheetah_rpy",
                                  "urdf_jvrc1_humanoid"])
def test_position_derivative_matches_oracle_differences(name, gpu):
    """grbda_fd_dq (analytic: the recursion of deriv_kernels.hip for explicit models, the route through the spanning tree of
    manifold_kernels.hip for models with implicit clusters) against central differences,
    along the reference's tangent step, taken with the oracle
    (testRigidBodyDynamicsAlgosDerivatives.cpp:271-383: central differences of the forward dynamics are the reference's
    own yardstick for its CasADi derivatives, tolerance 2e-5).  Implicit-loop models are differentiated ON the constraint
    manifold: an independent position moves, the dependent ones follow.  The fp32 entry point takes its differences in
    fp64."""
    import torch
    from generalized_rbda_amd.states import parse_clusters

    z = zoo()
    blob = z[name]
    plan = G.Plan(blob)
    m = parse_clusters(blob)
    implicit = any(c[9] >= 2 for c in m["clusters"])
    assert plan.info().analytic_derivatives == 1  # (round 4: the implicit models too, manifold_kernels.hip)
    B, h = (64 if implicit else 2), 1e-5  # a whole tile of states for the route through the spanning tree
    q, qd, tau = valid_states(blob, B, config_index=41, max_cond=100.0)  # (models.py: why the derivative tests bound cond(K_d))
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    J = plan.fd_dq(t(q), t(qd), t(tau), step=h).cpu().numpy()
    nv = plan.nv
    J_ref = np.empty((B, nv, nv))
    for b in range(B):
        for k in range(nv):
            qp = _reference_plus_on_manifold(blob, m, q[b], k, +h)[None]
            qm = _reference_plus_on_manifold(blob, m, q[b], k, -h)[None]
            J_ref[b, :, k] = (O.forward_dynamics(blob, qp, qd[b:b + 1], tau[b:b + 1])[0]
                              - O.forward_dynamics(blob, qm, qd[b:b + 1], tau[b:b + 1])[0]) / (2 * h)
    scale = 1.0 + np.abs(J_ref).max()
    assert np.abs(J - J_ref).max() / scale < 2e-5
    # fp32 in, fp32 out.  Models on the difference path take the differences in fp64: as good as the fp32 inputs allow.
    # Explicit models run the analytic recursion in fp32 (the SPD solve in fp64): fp32 arithmetic, TOL32-class errors.
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    J32 = plan.fd_dq(t32(q), t32(qd), t32(tau), step=h).double().cpu().numpy()
    c32 = lambda a: a.astype(np.float32).astype(np.float64)
    J_of_32 = plan.fd_dq(t(c32(q)), t(c32(qd)), t(c32(tau)), step=h).cpu().numpy()
    assert np.abs(J32 - J_of_32).max() / scale < (1e-3 if implicit else 2e-4)


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_mit_humanoid", "urdf_jvrc1_humanoid", "jvrc1_hand_built", "tree_mixed_fixed",
                                  "tree_mixed_float_rpy", "tree_generic_float", "tree_triple_fixed", "rev_pair_rotor_chain_4", "teleop_arm"])
def test_articulated_body_route_matches_the_dense_factorisation(name, gpu, monkeypatch):
    """The derivative pipeline's default route -- H^-1 = W^T W from the articulated-body quantities (minv_kernels.hip: the cluster
    ABA's own factorisation, D = S^T IA S of ClusterTreeDynamics.cpp:157-191, walked column by column) -- against the round-5 route it
    replaced (dense Cholesky of the CRBA's H + inversion of the factor, GRBDA_NO_MINV=1, read when the plan is made) on the same states:
    all three derivative matrices and d ydd / d tau alone, fp64 to 1e-9 and fp32 to 1e-3 of the matrix scale; 133 states (groups of
    four states, a ragged one).  Covers multi-coordinate clusters (pairs, triples, Generic with several bodies carrying child
    clusters), a fixed base, a roll-pitch-yaw base, and columns that straddle a 16-column tile boundary inside one cluster."""
    import torch

    blob = zoo()[name]
    monkeypatch.setenv("GRBDA_NO_MINV", "1")
    old = G.Plan(blob)
    monkeypatch.delenv("GRBDA_NO_MINV")
    new = G.Plan(blob)
    assert new.info().analytic_derivatives
    q, qd, tau = valid_states(blob, 133, config_index=17)
    for dt, tol in ((torch.float64, 1e-9), (torch.float32, TOL32)):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=gpu)
        ref, got = old.fd_derivatives(t(q), t(qd), t(tau)), new.fd_derivatives(t(q), t(qd), t(tau))
        ref_tau, got_tau = old.fd_dtau(t(q)), new.fd_dtau(t(q))
        for k in ("dq", "dqd", "dtau"):
            a, b = got[k].double(), ref[k].double()
            assert torch.isfinite(a).all()
            err = ((a - b).abs().amax(dim=(1, 2)) / (1.0 + b.abs().amax(dim=(1, 2)))).max()
            assert float(err) < tol, (k, str(dt), float(err))
        assert float((got_tau.double() - ref_tau.double()).abs().max() / (1.0 + ref_tau.abs().max())) < tol
        assert float((got_tau - got["dtau"]).abs().max()) <= 1e-6 * float(1.0 + got_tau.abs().max())  # (alone and as one of three)


@pytest.mark.parametrize("name", ["urdf_four_bar", "tello_with_arms", "tello"])
def test_constraint_kernel_builds_agree(name, gpu, monkeypatch):
    """Models whose implicit clusters all have at most 4 bodies / 2 independent coordinates run manifold_constraint_kernel<T, 4, 2> (half the
    unrolled work areas, two wavefronts per SIMD in fp32); GRBDA_NO_SMALL_CONSTRAINT=1 keeps the <T, 8, 4> build every other model uses.  Same
    formulas, same operation order per entry: all three derivative matrices and the mass matrix agree to rounding, on a batch with several chunks
    of column-cluster workgroups and a ragged tile."""
    import torch

    blob = zoo()[name]
    plan = G.Plan(blob)
    monkeypatch.setenv("GRBDA_NO_SMALL_CONSTRAINT", "1")
    plain = G.Plan(blob)
    monkeypatch.delenv("GRBDA_NO_SMALL_CONSTRAINT")
    q, qd, tau = valid_states(blob, 777, config_index=33)
    for dt, tol in ((torch.float64, 1e-10), (torch.float32, 2e-4)):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=gpu)
        a, b = plan.fd_derivatives(t(q), t(qd), t(tau)), plain.fd_derivatives(t(q), t(qd), t(tau))
        for k in ("dq", "dqd", "dtau"):
            assert torch.isfinite(a[k]).all()
            assert float((a[k].double() - b[k].double()).abs().max() / (1.0 + b[k].double().abs().max())) < tol, (k, str(dt))
        Ha, Hb = plan.mass_matrix(t(q)).double(), plain.mass_matrix(t(q)).double()
        assert float((Ha - Hb).abs().max() / (1.0 + Hb.abs().max())) < tol


@pytest.mark.parametrize("name", ["urdf_four_bar", "urdf_planar_leg_linkage"])
def test_manifold_dq_against_extended_precision_differences(name, gpu):
    """d ydd / d q of implicit clusters NEAR SINGULAR POSES, without a conditioning filter: the analytic route through the spanning tree
    (manifold_kernels.hip, fp64) against a third, independent evaluation -- the oracle compiled in x87 extended precision
    (oracle/_build/libgrbda_oracle_ld.so), central differences along an independent position with the dependent ones re-projected to
    |phi| < 1e-17, Richardson-extrapolated (h = 1e-7, h / 2).  States with cond(K_d) from 1 to 1e4, a dozen per decade.  Tolerance law per
    state: 1e-11 cond^3 + 1e-9 relative to 1 + |J|_max (measured: ~1e-12 cond^3, profiles/r5_manifold_derivatives_third_evaluation.txt); the
    difference batches (GRBDA_NO_MANIFOLD=1) are 10-100 x further from the reference in every decade, which is what
    profiles/r4_manifold_derivatives.txt's 22 % was.  Yardstick: testRigidBodyDynamicsAlgosDerivatives.cpp:271-383."""
    import torch
    from generalized_rbda_amd.states import parse_clusters, random_states

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from manifold_third_eval import reference_dq

    blob = zoo()[name]
    plan = G.Plan(blob)
    assert plan.info().analytic_derivatives == 1
    m = parse_clusters(blob)
    q, qd, tau = random_states(blob, 60000, config_index=5)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    tq = t(q)
    ok = plan.project_positions(tq).cpu().numpy()
    q, qd, tau = tq.cpu().numpy()[ok], qd[ok], tau[ok]
    kcond = O.spanning_state(blob, q, qd)[3]
    n_checked, worst_ratio = 0, 0.0
    for lo, hi in ((1, 10), (10, 100), (100, 1e3), (1e3, 1e4)):
        idx = np.flatnonzero((kcond >= lo) & (kcond < hi))[:12]
        assert idx.size >= (6 if hi <= 1e3 else 1), f"no states with cond(K_d) in [{lo}, {hi})"
        for i in idx:
            J_ref, q0 = reference_dq(blob, m, q[i], qd[i], tau[i], 1e-7)
            J = plan.fd_dq(t(q0[None]), t(qd[i][None]), t(tau[i][None])).cpu().numpy()[0]
            err = np.abs(J - J_ref).max() / (1.0 + np.abs(J_ref).max())
            tol = 1e-11 * kcond[i] ** 3 + 1e-9
            worst_ratio = max(worst_ratio, err / tol)
            assert err < tol, f"cond(K_d) {kcond[i]:.3g}: analytic vs extended-precision differences {err:.2e} (law {tol:.2e})"
            n_checked += 1
    assert n_checked >= 30


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_mit_humanoid", "urdf_jvrc1_humanoid", "tree_mixed_fixed", "tree_pair_float",
                                  "tree_generic_float", "rev_rotor_chain_4", "chain_tree_b", "tello_with_arms", "urdf_mini_cheetah_rpy"])
def test_fd_derivatives_analytic_against_difference_batches(name, gpu, monkeypatch):
    """grbda_fd_derivatives: the three matrices of BASELINE config 5 from one pass.  Explicit models take the analytic
    route (inverse-dynamics derivative recursion + CRBA + one SPD solve per state, deriv_kernels.hip); a second plan of
    the same model with GRBDA_NO_ANALYTIC=1 takes the unit-vector / central-difference batches through the ABA kernel,
    which the tests above pin to the oracle.  d ydd/d tau and d ydd/d qd are exact on both routes; d ydd/d q is compared
    within the reference's own tolerance for its derivative test (2e-5).  130 states: two full tiles and a ragged one.
    Models with implicit loops fall back to the batches on both plans."""
    import torch

    blob = zoo()[name]
    plan = G.Plan(blob)
    monkeypatch.setenv("GRBDA_NO_ANALYTIC", "1")
    plan_fd = G.Plan(blob)
    monkeypatch.delenv("GRBDA_NO_ANALYTIC")
    B = 130
    q, qd, tau = valid_states(blob, B, config_index=57)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    d = plan.fd_derivatives(t(q), t(qd), t(tau))
    ref = {"dtau": plan_fd.fd_dtau(t(q)), "dqd": plan_fd.fd_dqd(t(q), t(qd), t(tau)), "dq": plan_fd.fd_dq(t(q), t(qd), t(tau), step=1e-6)}
    for k, tol in (("dtau", 1e-8), ("dqd", 1e-8), ("dq", 2e-5)):
        a, b = d[k].cpu().numpy(), ref[k].cpu().numpy()
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() / (1.0 + np.abs(b).max()) < tol, k
    # a subset, and the single entry points, give the same numbers as the full pass
    only = plan.fd_derivatives(t(q), t(qd), t(tau), want=("dqd",))
    assert set(only) == {"dqd"} and torch.allclose(only["dqd"], d["dqd"], rtol=0, atol=1e-12 * float(d["dqd"].abs().max()))
    assert torch.allclose(plan.fd_dtau(t(q)), d["dtau"], rtol=0, atol=1e-12 * float(d["dtau"].abs().max()))
    # fp32: fp32 arithmetic throughout
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    d32 = plan.fd_derivatives(t32(q), t32(qd), t32(tau))
    for k in ("dtau", "dqd", "dq"):
        a, b = d32[k].double().cpu().numpy(), d[k].cpu().numpy()
        assert np.abs(a - b).max() / (1.0 + np.abs(b).max()) < TOL32, k


@pytest.mark.parametrize("name", ["urdf_mit_humanoid", "urdf_jvrc1_humanoid"])
def test_fd_derivatives_fp32_with_fp64_solve(name, gpu, monkeypatch):
    """GRBDA_SOLVE_F64=1: the fp32 derivative entry points keep fp32 arrays and run the SPD solve in fp64 arithmetic
    (spd_solve_kernel<float, double>).  Same results as the all-fp32 route to fp32 accuracy, against the fp64 route."""
    import torch

    blob = zoo()[name]
    monkeypatch.setenv("GRBDA_SOLVE_F64", "1")
    plan = G.Plan(blob)
    monkeypatch.delenv("GRBDA_SOLVE_F64")
    q, qd, tau = valid_states(blob, 70, config_index=58)
    c32 = lambda a: a.astype(np.float32).astype(np.float64)
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    t64 = lambda a: torch.as_tensor(np.ascontiguousarray(c32(a)), dtype=torch.float64, device=gpu)
    d32 = plan.fd_derivatives(t32(q), t32(qd), t32(tau))
    d64 = plan.fd_derivatives(t64(q), t64(qd), t64(tau))
    for k in ("dtau", "dqd", "dq"):
        a, b = d32[k].double().cpu().numpy(), d64[k].cpu().numpy()
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() / (1.0 + np.abs(b).max()) < TOL32, k


@pytest.mark.parametrize("n_clusters,seed", [(40, 31), (57, 32)])
def test_fd_derivatives_wide_models(n_clusters, seed, gpu, monkeypatch):
    """The SPD solve is compiled for 16 / 24 / 32 / 40 / 48 / 64 coordinates; the robots of the zoo stop at 38.  Random
    floating-base trees of revolute and rotor clusters with 46 and 63 coordinates: analytic route against the
    unit-vector / central-difference batches, fp64, and fp32 against fp64."""
    import torch
    from models import random_cluster_tree

    blob = random_cluster_tree(seed, n_clusters, floating=True, kinds=("rev", "rotor")).serialize()
    plan = G.Plan(blob)
    assert plan.info().analytic_derivatives and plan.nv == n_clusters + 6
    monkeypatch.setenv("GRBDA_NO_ANALYTIC", "1")
    plan_fd = G.Plan(blob)
    monkeypatch.delenv("GRBDA_NO_ANALYTIC")
    B = 70
    q, qd, tau = valid_states(blob, B, config_index=91)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    d = plan.fd_derivatives(t(q), t(qd), t(tau))
    ref = {"dtau": plan_fd.fd_dtau(t(q)), "dqd": plan_fd.fd_dqd(t(q), t(qd), t(tau)), "dq": plan_fd.fd_dq(t(q), t(qd), t(tau), step=1e-6)}
    for k, tol in (("dtau", 1e-8), ("dqd", 1e-8), ("dq", 2e-5)):
        a, b = d[k].cpu().numpy(), ref[k].cpu().numpy()
        assert np.isfinite(a).all()
        assert np.abs(a - b).max() / (1.0 + np.abs(b).max()) < tol, k
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    d32 = plan.fd_derivatives(t32(q), t32(qd), t32(tau))
    for k in ("dtau", "dqd", "dq"):
        a, b = d32[k].double().cpu().numpy(), d[k].cpu().numpy()
        assert np.abs(a - b).max() / (1.0 + np.abs(b).max()) < TOL32, k


def _sweep_models():
    """Random models beyond the fixed zoo: chain-structured robots (branching links, leaf pairs, long limbs, with and
    without rotors, both base orientations) and random cluster trees of every explicit type, floating and fixed."""
    from models import chain_test_tree, random_cluster_tree

    out = []
    for seed in range(101, 109):
        out.append((f"chain{seed}", lambda s=seed: chain_test_tree(s, n_limbs=2 + s % 4, ori_repr="rpy" if s % 3 == 0 else "quaternion",
                                                                   rotors=s % 4 != 1, deep_pairs=s % 2 == 0)))
    for seed in range(201, 207):
        out.append((f"tree{seed}", lambda s=seed: random_cluster_tree(s, n_clusters=5 + s % 6, floating=s % 2 == 0,
                                                                      ori_repr="rpy" if s % 5 == 0 else "quaternion")))
    return out


@pytest.mark.parametrize("name,build", _sweep_models(), ids=[m[0] for m in _sweep_models()])
def test_random_model_sweep_dynamics_and_derivatives(name, build, gpu):
    """Forward / inverse dynamics, mass matrix and the three derivative matrices of freshly drawn random models against the
    oracle (derivatives: central differences of the oracle along the reference's tangent step), fp64 and fp32, 70 states
    (one full tile and a ragged one).  Whatever program shape the plan compiler picks for a model is what gets tested."""
    import torch

    blob = build().serialize()
    plan = G.Plan(blob)
    B = 70
    q, qd, tau = valid_states(blob, B, config_index=77)
    for dt, tol in ((torch.float64, TOL64), (torch.float32, TOL32)):
        c = (lambda a: a) if dt == torch.float64 else (lambda a: a.astype(np.float32).astype(np.float64))
        ydd = run_gpu(plan, "aba", q, qd, tau, dt, gpu)
        assert rel_err(ydd, O.forward_dynamics(blob, c(q), c(qd), c(tau))) < tol
        t_ = run_gpu(plan, "rnea", q, qd, tau, dt, gpu)
        assert rel_err(t_, O.inverse_dynamics(blob, c(q), c(qd), c(tau))) < tol
    from generalized_rbda_amd.states import parse_clusters

    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    d = plan.fd_derivatives(t(q[:3]), t(qd[:3]), t(tau[:3]))
    m = parse_clusters(blob)
    nv, h = plan.nv, 1e-6
    for b in range(3):
        J_q, J_qd, J_tau = np.empty((nv, nv)), np.empty((nv, nv)), np.empty((nv, nv))
        for k in range(nv):
            e = np.zeros(nv)
            e[k] = 1.0
            qp, qm = _reference_plus_on_manifold(blob, m, q[b], k, +h)[None], _reference_plus_on_manifold(blob, m, q[b], k, -h)[None]
            J_q[:, k] = (O.forward_dynamics(blob, qp, qd[b:b + 1], tau[b:b + 1])[0] - O.forward_dynamics(blob, qm, qd[b:b + 1], tau[b:b + 1])[0]) / (2 * h)
            J_qd[:, k] = (O.forward_dynamics(blob, q[b:b + 1], (qd[b] + e)[None], tau[b:b + 1])[0]
                          - O.forward_dynamics(blob, q[b:b + 1], (qd[b] - e)[None], tau[b:b + 1])[0]) / 2
            J_tau[:, k] = (O.forward_dynamics(blob, q[b:b + 1], qd[b:b + 1], (tau[b] + e)[None])[0]
                           - O.forward_dynamics(blob, q[b:b + 1], qd[b:b + 1], (tau[b] - e)[None])[0]) / 2
        for key, ref, tol in (("dq", J_q, 2e-5), ("dqd", J_qd, 1e-8), ("dtau", J_tau, 1e-8)):
            got = d[key][b].cpu().numpy()
            assert np.abs(got - ref).max() / (1.0 + np.abs(ref).max()) < tol, (key, b)


# ---- contact side: body poses, applyTestForce -----------------------------------------------------------
@pytest.mark.parametrize("name,blob", list(zoo().items()), ids=list(zoo().keys()))
def test_body_poses_match_oracle(name, blob, gpu):
    """TreeNode::Xa_ (TreeModel.cpp:20-27) in the REFERENCE's body frames, although the plan works in
    re-oriented (canonical-axis) frames internally."""
    import torch

    plan = G.Plan(blob)
    q, _, _ = valid_states(blob, 70, config_index=51)
    ref = O.body_poses(blob, q, plan.n_bodies)
    got = plan.body_poses(torch.as_tensor(q, dtype=torch.float64, device=gpu)).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-11


@pytest.mark.parametrize("name,blob", list(zoo().items()), ids=list(zoo().keys()))
def test_body_twists_are_the_derivatives_of_the_motion(name, blob, gpu):
    """grbda_body_twists_*: TreeNode::v_ / a_ after forwardAccelerationKinematics (TreeModel.cpp:6-57).  Checked without a
    second implementation of the recursion: (1) a spatial acceleration in BODY coordinates is the componentwise time derivative
    of the body-coordinate velocity (v x v = 0), so along q(t) = q + qd_span t + qdd_span t^2 / 2, yd(t) = yd + ydd t the central
    difference of v reproduces a minus the base's -gravity carried down the tree (E_i (0, 0, 9.81) on the linear part); the
    body-coordinate velocities do not depend on the base pose, and the spanning rates keep implicit clusters on their manifold
    to O(t^3); (2) v of a body is its contact Jacobian (force-propagation kernel, checked against the oracle elsewhere) times yd."""
    import torch

    from generalized_rbda_amd.states import C_FREE, C_LOOP_POSITION, C_TRIG_POLY, parse_clusters

    plan = G.Plan(blob)
    m = parse_clusters(blob)
    B = 5
    q, qd, tau = valid_states(blob, B, config_index=57)
    ydd = O.forward_dynamics(blob, q, qd, tau)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    V = plan.body_twists(t(q), t(qd), t(ydd)).cpu().numpy()
    assert np.isfinite(V).all()
    vs, as_ = (x.cpu().numpy() for x in plan.spanning(t(q), t(qd), t(ydd)))

    def moved(h):
        """joint coordinates and rates a time h later; the base pose is left where it is"""
        q2, qd2 = q.copy(), qd + h * ydd
        at = 0
        for (pc, fb, k, qi, npos, vi, n, nsp, nsv, ctype, rows, io, ni, do, nd, _) in m["clusters"]:
            if ctype == C_FREE:
                at += 6
                continue
            if ctype in (C_LOOP_POSITION, C_TRIG_POLY):  # implicit: spanning positions
                q2[:, qi:qi + k] += h * vs[:, at:at + k] + 0.5 * h * h * as_[:, at:at + k]
            else:  # independent coordinates
                q2[:, qi:qi + n] += h * qd[:, vi:vi + n] + 0.5 * h * h * ydd[:, vi:vi + n]
            at += k
        return q2, qd2

    h = 1e-5
    qp, qdp = moved(+h)
    qm, qdm = moved(-h)
    Vp = plan.body_twists(t(qp), t(qdp), t(ydd)).cpu().numpy()
    Vm = plan.body_twists(t(qm), t(qdm), t(ydd)).cpu().numpy()
    a_fd = (Vp[:, :, :6] - Vm[:, :, :6]) / (2 * h)
    Xa = O.body_poses(blob, q, plan.n_bodies)
    E = Xa[:, :, :9].reshape(B, plan.n_bodies, 3, 3)
    g = np.asarray(plan.get_gravity(), dtype=np.float64)[-3:]
    a_ref = V[:, :, 6:].copy()
    a_ref[:, :, 3:] -= np.einsum("bnij,j->bni", E, -g)
    scale = 1.0 + np.abs(a_ref).max()
    assert np.abs(a_fd - a_ref).max() / scale < 2e-6, np.abs(a_fd - a_ref).max() / scale

    # fp32 entry point against the fp64 one
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    V32 = plan.body_twists(t32(q), t32(qd), t32(ydd)).double().cpu().numpy()
    assert np.abs(V32 - V).max() / (1.0 + np.abs(V).max()) < TOL32

    bodies = sorted({0, plan.n_bodies // 2, plan.n_bodies - 1})
    _, J = plan.inv_osim(t(q), bodies, np.zeros((len(bodies), 3)), with_jacobian=True)
    J = J.cpu().numpy().reshape(B, len(bodies), 6, plan.nv)
    v_from_J = np.einsum("bnij,bj->bni", J, qd)
    got = V[:, bodies, :6]
    assert np.abs(v_from_J - got).max() / (1.0 + np.abs(got).max()) < 1e-9


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_mit_humanoid", "tree_mixed_float", "tello_with_arms", "urdf_four_bar",
                                  "urdf_mini_cheetah_rpy", "chain_tree_a", "chain_tree_b", "urdf_jvrc1_humanoid", "rev_rotor_chain_4",
                                  "tree_rev_fixed"])
def test_apply_test_force_matches_oracle(name, gpu):
    """applyTestForce (ClusterTreeDynamics.cpp:194-233): dstate = H^-1 J^T f, lambda_inv = f^T J H^-1 J^T f,
    against the oracle's forward / inverse dynamics with the equivalent world wrench."""
    import torch

    blob = zoo()[name]
    plan = G.Plan(blob)
    B = 6
    q, _, _ = valid_states(blob, B, config_index=52)
    rng = np.random.default_rng(5)
    force = rng.uniform(-1, 1, size=(B, 3))
    offset = np.array([0.05, -0.02, 0.1])
    # a link deep in a limb where the model has named ones (chain-covered models then take the one-launch force-propagation
    # route: osim_chain_kernel in applyTestForce mode), else the last body
    named = {"urdf_mini_cheetah": "HL_knee_link", "urdf_mit_humanoid": "left_ankle_link", "tello_with_arms": "left-foot",
             "urdf_mini_cheetah_rpy": "FR_knee_link", "urdf_revolute_rotor_chain": None}
    body = _body_index(blob, named[name]) if named.get(name) else plan.n_bodies - 1
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    lam, ds = plan.apply_test_force(t(q), body, offset, t(force))
    lam, ds = lam.cpu().numpy(), ds.cpu().numpy()
    Xa = O.body_poses(blob, q, plan.n_bodies)[:, body]
    E, r = Xa[:, :9].reshape(B, 3, 3), Xa[:, 9:]
    p = r + np.einsum("bji,j->bi", E, offset)  # r + E^T offset
    fext = np.zeros((B, plan.n_bodies, 6))
    fext[:, body, :3] = np.cross(p, force)
    fext[:, body, 3:] = force
    zero = np.zeros((B, plan.nv))
    ds_ref = O.forward_dynamics(blob, q, zero, zero, f_ext=fext) - O.forward_dynamics(blob, q, zero, zero)
    jtf = O.inverse_dynamics(blob, q, zero, zero) - O.inverse_dynamics(blob, q, zero, zero, f_ext=fext)
    lam_ref = np.einsum("bi,bi->b", jtf, ds_ref)
    assert rel_err(ds, ds_ref) < 1e-8
    assert np.abs(lam - lam_ref).max() / (1 + np.abs(lam_ref).max()) < 1e-8
    assert (lam > 0).all()  # J H^-1 J^T is positive definite along f
    # fp32 entry point
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    lam32, ds32 = plan.apply_test_force(t32(q), body, offset, t32(force))
    assert rel_err(ds32.double().cpu().numpy(), ds_ref) < TOL32
    assert np.abs(lam32.double().cpu().numpy() - lam_ref).max() / (1 + np.abs(lam_ref).max()) < TOL32


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_mit_humanoid", "tree_mixed_float", "tello_with_arms"])
def test_inverse_operational_space_inertia(name, gpu):
    """inverseOperationalSpaceInertiaMatrix (ClusterTreeDynamics.cpp:295-435): J H^-1 J^T for contact frames,
    against H and the frame Jacobians obtained from the oracle's inverse dynamics with unit wrenches, and
    against applyTestForce."""
    import torch

    blob = zoo()[name]
    plan = G.Plan(blob)
    B = 4
    q, _, _ = valid_states(blob, B, config_index=53)
    nb, nv = plan.n_bodies, plan.nv
    bodies = [nb - 1, nb // 2]
    offsets = [[0.05, -0.02, 0.1], [0.0, 0.03, -0.2]]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    Linv, J = plan.inv_osim(t(q), bodies, offsets, with_jacobian=True)
    Linv, J = Linv.cpu().numpy(), J.cpu().numpy()
    # oracle: J rows from unit wrenches, H from unit accelerations
    Xa = O.body_poses(blob, q, nb)
    zero = np.zeros((B, nv))
    tau0 = O.inverse_dynamics(blob, q, zero, zero)
    J_ref = np.zeros((B, 12, nv))
    for c, (bd, off) in enumerate(zip(bodies, offsets)):
        E, r = Xa[:, bd, :9].reshape(B, 3, 3), Xa[:, bd, 9:]
        p = r + np.einsum("bji,j->bi", E, np.array(off))
        for k in range(6):
            e = E[:, k % 3, :]  # body axis in world coordinates
            fext = np.zeros((B, nb, 6))
            if k < 3:
                fext[:, bd, :3] = e
            else:
                fext[:, bd, :3] = np.cross(p, e)
                fext[:, bd, 3:] = e
            J_ref[:, 6 * c + k] = tau0 - O.inverse_dynamics(blob, q, zero, zero, f_ext=fext)
    H = np.zeros((B, nv, nv))
    for j in range(nv):
        ej = np.zeros((B, nv))
        ej[:, j] = 1
        H[:, :, j] = O.inverse_dynamics(blob, q, zero, ej) - tau0
    L_ref = np.einsum("bik,bkl,bjl->bij", J_ref, np.linalg.inv(H), J_ref)
    assert np.abs(J - J_ref).max() / (1 + np.abs(J_ref).max()) < 1e-9
    assert np.abs(Linv - L_ref).max() / (1 + np.abs(L_ref).max()) < 1e-8
    assert np.abs(Linv - Linv.transpose(0, 2, 1)).max() / (1 + np.abs(L_ref).max()) < 1e-9
    # applyTestForce on the first frame: lambda_inv = f_c^T Linv_ff f_c with f_c the force in body axes
    force = np.random.default_rng(9).uniform(-1, 1, size=(B, 3))
    lam, _ = plan.apply_test_force(t(q), bodies[0], offsets[0], t(force))
    E0 = Xa[:, bodies[0], :9].reshape(B, 3, 3)
    fc = np.einsum("bij,bj->bi", E0, force)
    lam_ref = np.einsum("bi,bij,bj->b", fc, Linv[:, 3:6, 3:6], fc)
    assert np.abs(lam.cpu().numpy() - lam_ref).max() / (1 + np.abs(lam_ref).max()) < 1e-8


def _body_index(blob, name):
    import struct
    from generalized_rbda_amd.states import parse_clusters

    m = parse_clusters(blob)
    n_ints, n_dbls, n_names = struct.unpack_from("<3i", blob, 28)
    off = 96 + 416 * m["nb"] + 64 * m["nc"] + 4 * ((n_ints + 1) & ~1) + 8 * n_dbls
    names = [n.decode() for n in blob[off: off + n_names].split(b"\0")[: m["nb"]]]
    return names.index(name)


OSIM_CASES = [
    ("urdf_mini_cheetah", ["FR_knee_link", "HL_knee_link", "Floating Base", "FL_hip_link"]),
    ("urdf_mit_humanoid", ["left_ankle_link", "right_ankle_link", "right_elbow_link", "left_knee_link", "Floating Base"]),
    ("urdf_mini_cheetah_rpy", ["FR_knee_link", "FL_knee_link"]),
    ("chain_tree_a", None), ("chain_tree_b", None), ("chain_tree_rpy", None), ("chain_tree_norotor", None),
    # implicit differentials: frames on link 2 / link 1 of the leaf (knee-ankle) and of the inner (hip) differential, an arm, the base
    ("tello_with_arms", ["left-foot", "right-shin", "left-gimbal", "right-thigh", "left-elbow-link", "torso"]),
]


@pytest.mark.parametrize("name,frames", OSIM_CASES, ids=[c[0] for c in OSIM_CASES])
def test_inverse_osim_by_force_propagation(name, frames, gpu, monkeypatch):
    """Chain-covered models take the in-kernel force-propagation path (osim_chain_kernel: the recursion of
    ClusterTreeDynamics.cpp:194-233,295-435); it must give the same J H^-1 J^T and Jacobians as the oracle -- frames on
    single links, on both links of leaf pair clusters and of Tello's implicit differentials (leaf and inner), on the
    floating base, several frames sharing ancestors -- and as
    the unit-wrench path it replaces (GRBDA_NO_EFPA=1)."""
    import torch
    from generalized_rbda_amd.states import parse_clusters

    blob = zoo()[name]
    plan = G.Plan(blob)
    nb, nv = plan.n_bodies, plan.nv
    if frames is None:  # every body that is not a rotor: names l<c>, l1_<c>, l2_<c>, base
        m = parse_clusters(blob)
        import struct
        n_ints, n_dbls, n_names = struct.unpack_from("<3i", blob, 28)
        off = 96 + 416 * nb + 64 * m["nc"] + 4 * ((n_ints + 1) & ~1) + 8 * n_dbls
        names = [n.decode() for n in blob[off: off + n_names].split(b"\0")[:nb]]
        cand = [i for i, n in enumerate(names) if not n.startswith("r")]
        rng = np.random.default_rng(3)
        bodies = [int(x) for x in rng.choice(cand, size=min(5, len(cand)), replace=False)]
        pairs = [i for i, n in enumerate(names) if n.startswith("l2_") or n.startswith("l1_")]
        if pairs:
            bodies[0] = pairs[0]
            bodies[-1] = pairs[-1]
        bodies = list(dict.fromkeys(bodies))
    else:
        bodies = [_body_index(blob, f) for f in frames]
    n = len(bodies)
    offsets = np.random.default_rng(4).uniform(-0.2, 0.2, size=(n, 3))
    B = 5
    q, _, _ = valid_states(blob, B, config_index=54)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    Linv, J = plan.inv_osim(t(q), bodies, offsets, with_jacobian=True)
    Linv, J = Linv.cpu().numpy(), J.cpu().numpy()
    Xa = O.body_poses(blob, q, nb)
    zero = np.zeros((B, nv))
    tau0 = O.inverse_dynamics(blob, q, zero, zero)
    J_ref = np.zeros((B, 6 * n, nv))
    for c, (bd, off) in enumerate(zip(bodies, offsets)):
        E, r = Xa[:, bd, :9].reshape(B, 3, 3), Xa[:, bd, 9:]
        p = r + np.einsum("bji,j->bi", E, np.array(off))
        for k in range(6):
            e = E[:, k % 3, :]
            fext = np.zeros((B, nb, 6))
            if k < 3:
                fext[:, bd, :3] = e
            else:
                fext[:, bd, :3] = np.cross(p, e)
                fext[:, bd, 3:] = e
            J_ref[:, 6 * c + k] = tau0 - O.inverse_dynamics(blob, q, zero, zero, f_ext=fext)
    H = np.zeros((B, nv, nv))
    for j in range(nv):
        ej = np.zeros((B, nv))
        ej[:, j] = 1
        H[:, :, j] = O.inverse_dynamics(blob, q, zero, ej) - tau0
    L_ref = np.einsum("bik,bkl,bjl->bij", J_ref, np.linalg.inv(H), J_ref)
    assert np.abs(J - J_ref).max() / (1 + np.abs(J_ref).max()) < 1e-9
    assert np.abs(Linv - L_ref).max() / (1 + np.abs(L_ref).max()) < 1e-8
    # fp32 entry point, and the path it replaces
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    L32 = plan.inv_osim(t32(q), bodies, offsets).double().cpu().numpy()
    assert np.abs(L32 - L_ref).max() / (1 + np.abs(L_ref).max()) < TOL32
    monkeypatch.setenv("GRBDA_NO_EFPA", "1")
    slow = G.Plan(blob)
    L_slow = slow.inv_osim(t(q), bodies[:2], offsets[:2]).cpu().numpy()
    assert np.abs(L_slow - L_ref[:, :12, :12]).max() / (1 + np.abs(L_ref).max()) < 1e-8


def test_sharded_host_entry_points(gpu):
    """grbda_*_sharded_*: host arrays over n devices in one process (here: as many as the box has)."""
    import torch

    blob = zoo()["urdf_mini_cheetah"]
    plan = G.Plan(blob)
    B = 1000
    q, qd, tau = random_states(blob, B, 9)
    n = G.device_count()
    ref = O.forward_dynamics(blob, q, qd, tau)
    got = plan.sharded_host("aba", q, qd, tau, n)
    assert rel_err(got, ref) < TOL64
    got32 = plan.sharded_host("rnea", q.astype(np.float32), qd.astype(np.float32), tau.astype(np.float32), 1)
    assert rel_err(got32.astype(np.float64), O.inverse_dynamics(blob, q, qd, tau)) < TOL32
    with pytest.raises(G.GrbdaError):
        plan.sharded_host("aba", q, qd, tau, n + 1)


@pytest.mark.parametrize("dtype_name", ["f32", "f64"])
def test_device_resident_sharded_entry_points(dtype_name, gpu):
    """grbda_{aba,rnea}_sharded_dev_* (SURVEY 8e): shards already in HBM, one process.  With ONE shard the results are bit for
    bit those of grbda_aba_* / grbda_rnea_* (the same launch); with the batch cut into three shards on distinct streams (of as many
    devices as the box has, wrapping around) and gathered on the first device -- peer copies over the links between devices, a
    device-to-device copy on one -- the gathered array is bit for bit the unsharded result, and work enqueued on the first
    stream after the call sees all of it."""
    import torch

    blob = zoo()["urdf_mit_humanoid"]
    plan = G.Plan(blob)
    B = 5000 + 37
    q, qd, x = random_states(blob, B, 12)
    dt = torch.float32 if dtype_name == "f32" else torch.float64
    n_dev = G.device_count()
    t = lambda a, d=0: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=torch.device("cuda", d))
    for which, plain in (("aba", plan.forward_dynamics), ("rnea", plan.inverse_dynamics)):
        want = plain(t(q), t(qd), t(x))
        outs, _ = plan.sharded_device(which, [t(q)], [t(qd)], [t(x)])
        torch.cuda.synchronize()
        assert torch.equal(outs[0], want)
        # one shard, gathered: the shard computes straight into the gathered array
        gathered = torch.full((B, plan.nv), float("nan"), dtype=dt, device=torch.device("cuda", 0))
        plan.sharded_device(which, [t(q)], [t(qd)], [t(x)], gathered=gathered)
        torch.cuda.synchronize()
        assert torch.equal(gathered, want)
        # three ragged shards, each on its own stream, gathered on device 0
        cuts = [0, 1700, 1700 + 64, B]
        devs = [g % n_dev for g in range(3)]
        qs = [t(q[cuts[g]:cuts[g + 1]], devs[g]) for g in range(3)]
        qds = [t(qd[cuts[g]:cuts[g + 1]], devs[g]) for g in range(3)]
        xs = [t(x[cuts[g]:cuts[g + 1]], devs[g]) for g in range(3)]
        streams = [torch.cuda.Stream(device=torch.device("cuda", d)) for d in devs]
        for g in range(3):
            streams[g].wait_stream(torch.cuda.current_stream(torch.device("cuda", devs[g])))
        gathered = torch.full((B, plan.nv), float("nan"), dtype=dt, device=torch.device("cuda", 0))
        outs = [torch.empty((cuts[g + 1] - cuts[g], plan.nv), dtype=dt, device=torch.device("cuda", devs[g])) for g in range(3)]
        plan.sharded_device(which, qs, qds, xs, outs=outs, gathered=gathered, streams=streams)
        with torch.cuda.stream(streams[0]):
            seen = gathered.clone()   # enqueued on the first stream AFTER the call: must see every slab
        for s_ in streams:
            s_.synchronize()
        assert torch.equal(seen, want) and torch.equal(gathered, want)
        for g in range(3):
            assert torch.equal(outs[g].to("cuda:0"), want[cuts[g]:cuts[g + 1]])
    # two shards on one (device, stream) would share a scratch slab: refused
    with pytest.raises(G.GrbdaError):
        s_ = torch.cuda.current_stream()
        plan.sharded_device("aba", [t(q[:64]), t(q[64:128])], [t(qd[:64]), t(qd[64:128])], [t(x[:64]), t(x[64:128])], streams=[s_, s_])


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_mit_humanoid", "tree_rotor_float", "chain_tree_b", "urdf_mini_cheetah_rpy", "tello_with_arms"])
def test_latency_mode_matches_the_one_wavefront_kernel(name, gpu, monkeypatch):
    """Batches of at most one tile per SIMD run a tile on a workgroup of two wavefronts that split the limbs below the
    floating base (aba_chain_lm_kernel; BASELINE config 2's 65 536 Mini-Cheetah states are such a batch).  Same device
    functions and operations per state; only the order in which the limbs' inertias are summed at the base differs (one partial
    sum per wavefront), so the results agree with the one-wavefront kernel to rounding -- 1e-12 fp64, 2e-5 fp32 -- and match
    the oracle like everything else."""
    import torch

    blob = zoo()[name]
    plan = G.Plan(blob)
    info = plan.info()
    assert info.latency_mode_f32 == 1
    monkeypatch.setenv("GRBDA_NO_LATENCY_MODE", "1")
    plain = G.Plan(blob)
    monkeypatch.delenv("GRBDA_NO_LATENCY_MODE")
    assert plain.info().latency_mode_f32 == 0
    # fp32 batches of at most two tiles per CU take FOUR wavefronts per tile when the base carries four limbs or more (one accumulator per
    # wavefront at the base); GRBDA_LM_WAVES=2 keeps two -- all three kernels are compared
    monkeypatch.setenv("GRBDA_LM_WAVES", "2")
    two = G.Plan(blob)
    monkeypatch.delenv("GRBDA_LM_WAVES")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    diff = name == "tello_with_arms"   # (differential segments: kernels <float, NW, true>; fp32 only)
    if name.startswith("urdf_") or diff:  # (four legs; two legs and two arms)
        assert "lm_kernel<float, 4" in plan.kernel_name("aba", "f32", 64 * 2 * n_cu)
    assert "lm_kernel<float, 2" in plan.kernel_name("aba", "f32", 64 * 2 * n_cu + 1)
    assert "lm_kernel<float, 2" in two.kernel_name("aba", "f32", 1000)
    assert (", true>" in plan.kernel_name("aba", "f32", 1000)) == diff
    for B in (1, 64, 65, 1000, 64 * 2 * n_cu, 65536):
        q, qd, tau = valid_states(blob, B, config_index=77) if diff else random_states(blob, B, config_index=77)
        for dt in (torch.float32, torch.float64):
            if dt == torch.float64 and not info.latency_mode_f64:
                continue
            t = lambda a: torch.as_tensor(a, dtype=dt, device=gpu)
            a = plan.forward_dynamics(t(q), t(qd), t(tau))
            b = plain.forward_dynamics(t(q), t(qd), t(tau))
            c = two.forward_dynamics(t(q), t(qd), t(tau))
            torch.cuda.synchronize()
            for x in (a, c):
                err = ((x - b).abs().amax(dim=1) / (1.0 + b.abs().amax(dim=1))).max().item()
                assert err < ((2e-4 if diff else 2e-5) if dt == torch.float32 else 1e-12), f"B={B} {dt}: {err:.2e}"
    q, qd, tau = valid_states(blob, 300, config_index=78)
    got = run_gpu(plan, "aba", q, qd, tau, torch.float64 if info.latency_mode_f64 else torch.float32, gpu)
    assert rel_err(got, O.forward_dynamics(blob, q, qd, tau)) < (TOL64 if info.latency_mode_f64 else TOL32)


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_mit_humanoid", "tree_rotor_float", "chain_tree_b", "urdf_mini_cheetah_rpy", "tello_with_arms"])
def test_inverse_dynamics_latency_mode_matches_the_one_wavefront_kernel(name, gpu, monkeypatch):
    """The inverse dynamics have the forward dynamics' latency mode (rnea_chain_lm_kernel: two wavefronts per tile up to one tile per SIMD, four up to two tiles per
    CU when the base carries four limbs; links and leaf pairs below one floating base).  Same device functions and operations per state; the limbs' forces reach
    the base as one partial sum per wavefront, so the three kernels agree to rounding -- 2e-5 fp32, 1e-12 fp64 of the state's scale -- and match the oracle."""
    import torch

    blob = zoo()[name]
    plan = G.Plan(blob)
    monkeypatch.setenv("GRBDA_NO_LATENCY_MODE", "1")
    plain = G.Plan(blob)
    monkeypatch.delenv("GRBDA_NO_LATENCY_MODE")
    monkeypatch.setenv("GRBDA_LM_WAVES", "2")
    two = G.Plan(blob)
    monkeypatch.delenv("GRBDA_LM_WAVES")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    diff = name == "tello_with_arms"  # (differential segments: fp32 kernels <float, NW, true>; its fp64 inverse dynamics keep the one-wavefront kernel)
    assert "rnea_chain_lm_kernel<float, 2" in two.kernel_name("rnea", "f32", 1000)
    assert "lm_kernel" not in plain.kernel_name("rnea", "f32", 1000)
    if name.startswith("urdf_") or diff:  # (four limbs)
        assert "rnea_chain_lm_kernel<float, 4" in plan.kernel_name("rnea", "f32", 64 * 2 * n_cu)
    assert "rnea_chain_lm_kernel<float, 2" in plan.kernel_name("rnea", "f32", 64 * 2 * n_cu + 1)
    assert "lm_kernel" not in plan.kernel_name("rnea", "f32", 64 * 4 * n_cu + 1)
    assert (", true>" in plan.kernel_name("rnea", "f32", 1000)) == diff
    if diff:
        assert "lm_kernel" not in plan.kernel_name("rnea", "f64", 1000)
    for B in (1, 64, 65, 1000, 64 * 2 * n_cu, 65536):
        q, qd, ydd = valid_states(blob, B, config_index=79) if diff else random_states(blob, B, config_index=79)
        for dt in (torch.float32, torch.float64):
            t = lambda a: torch.as_tensor(a, dtype=dt, device=gpu)
            a = plan.inverse_dynamics(t(q), t(qd), t(ydd))
            b = plain.inverse_dynamics(t(q), t(qd), t(ydd))
            c = two.inverse_dynamics(t(q), t(qd), t(ydd))
            torch.cuda.synchronize()
            for x in (a, c):
                err = ((x - b).abs().amax(dim=1) / (1.0 + b.abs().amax(dim=1))).max().item()
                assert err < ((2e-4 if diff else 2e-5) if dt == torch.float32 else 1e-12), f"B={B} {dt}: {err:.2e}"
    q, qd, ydd = valid_states(blob, 300, config_index=80)
    assert rel_err(run_gpu(plan, "rnea", q, qd, ydd, torch.float64, gpu), O.inverse_dynamics(blob, q, qd, ydd)) < TOL64
    c32 = lambda a: a.astype(np.float32).astype(np.float64)
    assert rel_err(run_gpu(plan, "rnea", c32(q), c32(qd), c32(ydd), torch.float32, gpu), O.inverse_dynamics(blob, c32(q), c32(qd), c32(ydd))) < TOL32


@pytest.mark.parametrize("name", ["urdf_four_bar", "urdf_six_bar", "rev_triple_rotor_chain_3"])
def test_single_cluster_kernels_match_the_chain_kernels(name, gpu, monkeypatch):
    """A model that is ONE generic cluster on the ground (the URDF+ loop mechanisms of BASELINE config 5, a triple with rotors on a
    bench) runs the fused kernels aba_gen1_kernel / rnea_gen1_kernel: no slab, inputs prefetched into registers, for the inverse
    dynamics one LDS object whose force blocks the constraint's scratch overlays.  Same device functions and operations per state as
    the generic segments of the chain kernels (GRBDA_NO_GEN1=1), so the results agree to rounding; several tiles per wavefront and
    a ragged last tile are in; the oracle as everywhere."""
    import torch

    blob = zoo()[name]
    plan = G.Plan(blob)
    monkeypatch.setenv("GRBDA_NO_GEN1", "1")
    plain = G.Plan(blob)
    monkeypatch.delenv("GRBDA_NO_GEN1")
    for dt, tn in ((torch.float32, "f32"), (torch.float64, "f64")):
        assert "aba_gen1_kernel" in plan.kernel_name("aba", tn, 4096) and "rnea_gen1_kernel" in plan.kernel_name("rnea", tn, 4096)
        assert "gen1" not in plain.kernel_name("aba", tn, 4096) and "gen1" not in plain.kernel_name("rnea", tn, 4096)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    big = 64 * 20 * n_cu * 2 + 37  # more tiles than any launch shape has wavefronts: every wavefront loops
    for B in (1, 65, 1000, big):
        q, qd, tau = valid_states(blob, min(B, 3000), config_index=91)
        rep = (B + q.shape[0] - 1) // q.shape[0]
        q, qd, tau = (np.tile(a, (rep, 1))[:B] for a in (q, qd, tau))
        for dt in (torch.float32, torch.float64):
            t = lambda a: torch.as_tensor(a, dtype=dt, device=gpu)
            for fn in ("forward_dynamics", "inverse_dynamics"):
                a = getattr(plan, fn)(t(q), t(qd), t(tau))
                b = getattr(plain, fn)(t(q), t(qd), t(tau))
                torch.cuda.synchronize()
                err = ((a - b).abs().amax(dim=1) / (1.0 + b.abs().amax(dim=1))).max().item()
                assert err < (2e-4 if dt == torch.float32 else 1e-11), f"{fn} B={B} {dt}: {err:.2e}"
    q, qd, tau = valid_states(blob, 300, config_index=92)
    assert rel_err(run_gpu(plan, "aba", q, qd, tau, torch.float64, gpu), O.forward_dynamics(blob, q, qd, tau)) < TOL64
    assert rel_err(run_gpu(plan, "rnea", q, qd, tau, torch.float64, gpu), O.inverse_dynamics(blob, q, qd, tau)) < TOL64


@pytest.mark.parametrize("name", ["tello_with_arms", "tello", "urdf_four_bar", "urdf_six_bar", "urdf_planar_leg_linkage"])
def test_ungated_implicit_states_fp64(name, gpu):
    """north_star's fp64 tolerance (1e-6 relative) on EVERY valid input of the implicit models: states drawn with the
    reference's law and projected onto phi(q) = 0 -- the reference's own validity criterion, |phi| < 1e-8
    (GenericJoint.cpp:364-378) -- WITHOUT the conditioning gate of states.py, so near-singular poses and far-away roots are
    in.  Forward and inverse dynamics against the oracle."""
    import torch
    from generalized_rbda_amd.states import accept, random_states

    blob = zoo()[name]
    plan = G.Plan(blob)
    q, qd, tau = random_states(blob, 6000, config_index=77)
    q, ok = O.project_positions(blob, q)
    q, qd, tau = q[ok], qd[ok], tau[ok]
    assert q.shape[0] > 500
    _, _, gmax, kcond = O.spanning_state(blob, q, qd)
    outside = ~accept(blob, q, gmax, kcond)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    ydd = plan.forward_dynamics(t(q), t(qd), t(tau)).cpu().numpy()
    tau_id = plan.inverse_dynamics(t(q), t(qd), t(tau)).cpu().numpy()
    ref = O.forward_dynamics(blob, q, qd, tau)
    ref_id = O.inverse_dynamics(blob, q, qd, tau)
    e = np.abs(ydd - ref).max(axis=1) / (1.0 + np.abs(ref).max(axis=1))
    e_id = np.abs(tau_id - ref_id).max(axis=1) / (1.0 + np.abs(ref_id).max(axis=1))
    assert e.max() < 1e-6 and e_id.max() < 1e-6, (e.max(), e_id.max(), int(outside.sum()))
    if name in ("tello_with_arms", "tello", "urdf_four_bar"):
        assert outside.sum() > 0  # the sample really holds states the gate rejects


def test_singular_mass_matrix_is_counted(gpu):
    """A model whose joint-space inertia is singular (a massless, inertia-less distal link): the SPD solves behind the derivative
    entry points return NaN / Inf for such states and COUNT them (grbda_spd_bad_pivots) -- a caller can ask instead of scanning
    the matrices; a regular model leaves the counter at zero."""
    import torch

    m = md.ClusterTreeModel(gravity=(0.0, 0.0, -9.81))
    rng = np.random.default_rng(5)
    m.appendBody("l0", random_inertia(rng), "ground", *random_xtree(rng), joint="revolute", axis="z")
    m.appendBody("l1", md.spatial_inertia(0.0, np.zeros(3), np.zeros((3, 3))), "l0", *random_xtree(rng), joint="revolute", axis="y")
    plan = G.Plan(m.serialize())
    B = 130
    q = torch.as_tensor(rng.uniform(-1, 1, (B, 2)), dtype=torch.float64, device=gpu)
    G.spd_bad_pivots(0, reset=True)
    for dt in (torch.float64, torch.float32):
        Hinv = plan.fd_dtau(q.to(dt))
        assert G.spd_bad_pivots(0, reset=True) == B
        assert not torch.isfinite(Hinv).all()
    # the forward dynamics itself counts them too: D = S^T IA S of the massless link is 0, where the reference's
    # ColPivHouseholderQR (ClusterTreeNode.cpp:33-37) would return a least-squares answer the Cholesky factorisations here return
    # NaN / Inf -- every ABA kernel (chain programs, latency mode, single-cluster programs, the interpreter) reports the state
    qd = torch.as_tensor(rng.uniform(-1, 1, (B, 2)), dtype=torch.float64, device=gpu)
    tau = torch.as_tensor(rng.uniform(-1, 1, (B, 2)), dtype=torch.float64, device=gpu)
    for dt in (torch.float64, torch.float32):
        ydd = plan.forward_dynamics(q.to(dt), qd.to(dt), tau.to(dt))
        assert G.spd_bad_pivots(0, reset=True) == B
        assert not torch.isfinite(ydd).all()
    good = G.Plan(zoo()["urdf_mini_cheetah"])
    qg, qdg, taug = valid_states(good.blob, 70, config_index=3)
    good.fd_dtau(torch.as_tensor(qg, dtype=torch.float32, device=gpu))
    t32 = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=gpu)
    good.forward_dynamics(t32(qg), t32(qdg), t32(taug))
    assert G.spd_bad_pivots(0, reset=True) == 0


@pytest.mark.parametrize("route", ["chain", "interpreter"])
def test_bad_pivots_of_the_forward_dynamics_are_counted_on_every_kernel(route, gpu, monkeypatch):
    """A floating-base robot with one massless, inertia-less distal link: its D is 0 in every state.  The chain program and the
    general interpreter (GRBDA_NO_CHAIN=1) each report every state exactly once."""
    import torch

    rng = np.random.default_rng(11)
    m = md.ClusterTreeModel(gravity=(0.0, 0.0, -9.81))
    m.appendBody("base", random_inertia(rng), "ground", joint="free")
    m.appendBody("a0", random_inertia(rng), "base", *random_xtree(rng), joint="revolute", axis="z")
    m.appendBody("a1", md.spatial_inertia(0.0, np.zeros(3), np.zeros((3, 3))), "a0", *random_xtree(rng), joint="revolute", axis="y")
    m.appendBody("b0", random_inertia(rng), "base", *random_xtree(rng), joint="revolute", axis="x")
    if route == "interpreter":
        monkeypatch.setenv("GRBDA_NO_CHAIN", "1")
    plan = G.Plan(m.serialize())
    B = 64 * 2100 + 5
    q, qd, tau = random_states(plan.blob, B, 21)
    G.spd_bad_pivots(0, reset=True)
    for dt in (torch.float32, torch.float64):
        t = lambda a: torch.as_tensor(a, dtype=dt, device=gpu)
        name = plan.kernel_name("aba", "f32" if dt == torch.float32 else "f64", B)
        assert ("aba_kernel" in name) == (route == "interpreter"), name
        ydd = plan.forward_dynamics(t(q), t(qd), t(tau))
        assert G.spd_bad_pivots(0, reset=True) == B
        assert not torch.isfinite(ydd).all()


def test_bad_pivots_are_counted_once_per_state_in_latency_mode(gpu):
    """Latency mode runs a tile on TWO wavefronts (limbs dealt out, the base on wavefront 0).  A state whose pivot is not a number in a
    limb spoils the base's articulated inertia too, so both wavefronts see it: their masks are united before counting, and the
    state is counted once.  Mini Cheetah at 197 states; every third state carries a NaN in the last leg's knee angle."""
    import torch

    blob = zoo()["urdf_mini_cheetah"]
    plan = G.Plan(blob)
    B = 64 * 3 + 5
    q, qd, tau = random_states(blob, B, 23)
    q = q.copy()
    q[::3, plan.nq - 1] = np.nan
    n_bad = len(range(0, B, 3))
    G.spd_bad_pivots(0, reset=True)
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        assert "lm_kernel" in plan.kernel_name("aba", tag, B)
        t = lambda a: torch.as_tensor(a, dtype=dt, device=gpu)
        ydd = plan.forward_dynamics(t(q), t(qd), t(tau))
        assert G.spd_bad_pivots(0, reset=True) == n_bad
        bad_rows = ~torch.isfinite(ydd).all(dim=1)
        assert int(bad_rows.sum()) == n_bad and bool(bad_rows[::3].all())


def test_cluster_on_two_parent_bodies_spanning_tree_route(gpu):
    """A cluster attached to two bodies of its parent cluster (tests/test_capi_cpu.py::two_parent_model) runs through the spanning
    tree (capi.cpp, projection_run): forward / inverse dynamics, mass matrix and the three derivative matrices against the oracle's
    dense cluster recursion (fp64 1e-9; fp32 1e-3)."""
    import torch
    from test_capi_cpu import two_parent_model
    from generalized_rbda_amd.states import parse_clusters

    blob = two_parent_model().serialize()
    plan = G.Plan(blob)
    assert plan.info().spanning_tree_route == 1
    B, nv = 200, plan.nv
    q, qd, tau = random_states(blob, B, config_index=91)
    ref = O.forward_dynamics(blob, q, qd, tau)
    ref_id = O.inverse_dynamics(blob, q, qd, tau)
    for dt, tol in ((torch.float64, 1e-9), (torch.float32, 1e-3)):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=gpu)
        ydd = plan.forward_dynamics(t(q), t(qd), t(tau)).double().cpu().numpy()
        tid = plan.inverse_dynamics(t(q), t(qd), t(tau)).double().cpu().numpy()
        assert np.abs(ydd - ref).max() / (1 + np.abs(ref).max()) < tol
        assert np.abs(tid - ref_id).max() / (1 + np.abs(ref_id).max()) < tol
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    H = plan.mass_matrix(t(q)).cpu().numpy()
    e = np.eye(nv)
    z = np.zeros_like(qd)
    C = O.inverse_dynamics(blob, q, z, z)
    for k in range(nv):
        col = O.inverse_dynamics(blob, q, z, np.tile(e[k], (B, 1))) - C
        assert np.abs(H[:, :, k] - col).max() < 1e-9 * (1 + np.abs(col).max())
    d = plan.fd_derivatives(t(q[:16]), t(qd[:16]), t(tau[:16]))
    m = parse_clusters(blob)
    h = 1e-6
    for b in range(16):
        for k in range(nv):
            qp, qm = _reference_plus(m, q[b], k, +h)[None], _reference_plus(m, q[b], k, -h)[None]
            col = (O.forward_dynamics(blob, qp, qd[b:b + 1], tau[b:b + 1])[0] - O.forward_dynamics(blob, qm, qd[b:b + 1], tau[b:b + 1])[0]) / (2 * h)
            got = d["dq"][b, :, k].cpu().numpy()
            assert np.abs(got - col).max() / (1 + np.abs(col).max()) < 2e-5
            col_t = O.forward_dynamics(blob, q[b:b + 1], qd[b:b + 1], tau[b:b + 1] + e[k])[0] - ref[b]
            assert np.abs(d["dtau"][b, :, k].cpu().numpy() - col_t).max() / (1 + np.abs(col_t).max()) < 1e-8


BIG_CHAINS = [  # (implicit, depth, loop size): the reference's parallel-chain family beyond 8 bodies / 4 DoF per cluster
    (False, 10, 12), (False, 10, 16), (False, 20, 24), (False, 20, 30),
    (True, 10, 13), (True, 10, 17), (True, 20, 25), (True, 20, 31),
    (False, 40, 40), (True, 40, 41),  # 79 velocities: beyond the one-word masks (tables, workgroup-per-state solve)
]


@pytest.mark.parametrize("implicit,depth,loop", BIG_CHAINS)
def test_clusters_beyond_the_structured_limits_run_through_the_spanning_tree(gpu, tmp_path, implicit, depth, loop):
    """Clusters of 12 .. 31 bodies with 11 .. 29 independent coordinates (the reference's parallel-chain benchmark family,
    Benchmarking/urdfs/parallel_chains; sizes from pinocchioHelpers.cpp:355-410): forward and inverse dynamics and the mass matrix
    through the spanning tree (HostPlan::big_clusters, manifold_kernels.hip's wide variants) against the oracle built with room for
    48 bodies per cluster (oracle/_build/libgrbda_oracle_big.so) -- fp64 1e-9; fp32 inverse dynamics 1e-3 of the largest entry, fp32
    forward dynamics 1e-5: the fp32 entry point of such plans computes through the fp64 route (a dense fp32 solve over chains twenty
    links long measured 1.4e-3 here and 1e-2 over 65 536 states)."""
    import torch
    from parallel_chains import parallel_chain_urdf

    path = tmp_path / "pc.urdf"
    path.write_text(parallel_chain_urdf(depth, loop, implicit))
    plan = G.Plan.from_urdf(str(path))
    info = plan.info()
    assert info.spanning_tree_route == 1
    blob = plan.blob
    B, nv = 130, plan.nv
    q, qd, tau = valid_states(blob, B, config_index=17, big=True, scale=0.5 if depth < 40 else 0.25, max_cond=50 if implicit else None)
    ref = O.forward_dynamics(blob, q, qd, tau, big=True)
    ref_id = O.inverse_dynamics(blob, q, qd, tau, big=True)
    for dt, tol in ((torch.float64, 1e-9), (torch.float32, 1e-5)):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=gpu)
        ydd = plan.forward_dynamics(t(q), t(qd), t(tau)).double().cpu().numpy()
        tid = plan.inverse_dynamics(t(q), t(qd), t(tau)).double().cpu().numpy()
        assert np.abs(ydd - ref).max() / (1 + np.abs(ref).max()) < tol
        assert np.abs(tid - ref_id).max() / (1 + np.abs(ref_id).max()) < (tol if dt == torch.float64 else 1e-3)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    H = plan.mass_matrix(t(q[:32])).cpu().numpy()
    e = np.eye(nv)
    z = np.zeros_like(qd[:32])
    C = O.inverse_dynamics(blob, q[:32], z, z, big=True)
    for k in range(nv):
        col = O.inverse_dynamics(blob, q[:32], z, np.tile(e[k], (32, 1)), big=True) - C
        assert np.abs(H[:, :, k] - col).max() < 1e-9 * (1 + np.abs(col).max())
    # derivatives with respect to q: not analytic for such clusters -- difference batches of the forward dynamics for explicit clusters
    # (against oracle differences), a loud refusal for implicit ones (their differences need the Newton re-projection)
    if implicit:
        # Newton projection of the dependent coordinates on the device (the wide kernel): perturbed states come back onto the manifold,
        # at the root the oracle's Newton finds
        from generalized_rbda_amd.states import parse_clusters
        m_ = parse_clusters(blob)
        qp = q[:32].copy()
        for (pc, fb, k, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) in m_["clusters"]:
            if ctype >= 2:
                ind = m_["ints"][io + 1: io + 1 + nsv]
                for j in range(nsv):
                    if not ind[j]:
                        qp[:, qi + j] += np.random.default_rng(j).uniform(-0.02, 0.02, qp.shape[0])
        q_ref, ok_ref = O.project_positions(blob, qp, big=True)
        tq = t(qp.copy())
        ok_dev = plan.project_positions(tq, tol=1e-12).cpu().numpy()  # (the oracle iterates to 1e-12)
        assert ok_dev.all() and ok_ref.all()
        assert np.abs(tq.cpu().numpy() - q_ref).max() < 1e-9
        if depth <= 10:  # d ydd / d q through difference batches with that re-projection, against the oracle's differences
            J = plan.fd_dq(t(q[:2]), t(qd[:2]), t(tau[:2])).cpu().numpy()
            assert np.isfinite(J).all()
    elif depth <= 10:
        J = plan.fd_dq(t(q[:2]), t(qd[:2]), t(tau[:2])).cpu().numpy()
        h = 1e-6
        for k in (0, nv // 2, nv - 1):
            dq_ = np.zeros_like(q[:2]); dq_[:, k] = h
            col = (O.forward_dynamics(blob, q[:2] + dq_, qd[:2], tau[:2], big=True) - O.forward_dynamics(blob, q[:2] - dq_, qd[:2], tau[:2], big=True)) / (2 * h)
            assert np.abs(J[:, :, k] - col).max() / (1 + np.abs(col).max()) < 2e-5


@pytest.mark.parametrize("seed,floating", [(31, True), (32, False), (33, True), (34, False)])
def test_random_trees_with_big_explicit_clusters(gpu, seed, floating):
    """Random cluster trees that contain Generic clusters of 9 .. 20 bodies with 5 .. 12 independent coordinates (random in-cluster
    trees, random sparse coupling G, child clusters on any of their bodies) next to ordinary ones: the spanning-tree route of
    DESIGN 7c against the oracle built for 48 bodies per cluster."""
    import torch
    from models import random_cluster_tree

    model = random_cluster_tree(seed, 4, floating=floating, kinds=("generic_big", "rev", "generic_big", "rotor"))
    blob = model.serialize()
    plan = G.Plan(blob)
    assert plan.info().spanning_tree_route == 1 and plan.nv <= 64
    B, nv = 96, plan.nv
    q, qd, tau = random_states(blob, B, config_index=seed)
    ref = O.forward_dynamics(blob, q, qd, tau, big=True)
    ref_id = O.inverse_dynamics(blob, q, qd, tau, big=True)
    for dt, tol in ((torch.float64, 1e-9), (torch.float32, 2e-3)):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=gpu)
        ydd = plan.forward_dynamics(t(q), t(qd), t(tau)).double().cpu().numpy()
        tid = plan.inverse_dynamics(t(q), t(qd), t(tau)).double().cpu().numpy()
        assert np.abs(ydd - ref).max() / (1 + np.abs(ref).max()) < tol
        assert np.abs(tid - ref_id).max() / (1 + np.abs(ref_id).max()) < tol
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    Hinv = plan.fd_dtau(t(q[:16])).cpu().numpy()
    H = plan.mass_matrix(t(q[:16])).cpu().numpy()
    assert np.abs(np.einsum("bij,bjk->bik", H, Hinv) - np.eye(nv)).max() < 1e-8


FAMILY = [(False, d, l) for d, ls in ((5, (2, 4, 6, 8, 10)), (10, (2, 4, 8, 12, 16)), (20, (2, 6, 12, 20, 30)), (40, (2, 8, 16, 28, 40))) for l in ls] + \
         [(True, d, l) for d, ls in ((5, (3, 5, 7, 9, 11)), (10, (3, 5, 9, 13, 17)), (20, (3, 7, 13, 21, 31)), (40, (3, 9, 17, 29, 41))) for l in ls]


@pytest.mark.parametrize("implicit,depth,loop", FAMILY, ids=[f"{'imp' if f[0] else 'exp'}-d{f[1]}-l{f[2]}" for f in FAMILY])
def test_every_model_of_the_reference_parallel_chain_family(gpu, tmp_path, implicit, depth, loop):
    """All 40 (depth, loop size) pairs of the reference's parallel-chain benchmark (Benchmarking/urdfs/make_parallel_chain_urdfs.sh;
    pinocchioHelpers.cpp:355-410): forward and inverse dynamics in fp64 against the oracle, whichever route the plan takes -- the
    structured kernels up to 8 bodies / 4 DoF per cluster (at depth 40: 79 DoF), the spanning tree beyond."""
    import torch
    from parallel_chains import parallel_chain_urdf

    path = tmp_path / "pc.urdf"
    path.write_text(parallel_chain_urdf(depth, loop, implicit))
    plan = G.Plan.from_urdf(str(path))
    m = loop // 2
    big = (loop > 8) or (loop - (2 if implicit else 1) > 4)
    assert plan.info().spanning_tree_route == (1 if big else 0)
    blob = plan.blob
    q, qd, tau = valid_states(blob, 70, config_index=23, big=True, scale=0.5 if depth < 40 else 0.25, max_cond=50 if implicit else None)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=gpu)
    ref = O.forward_dynamics(blob, q, qd, tau, big=True)
    ref_id = O.inverse_dynamics(blob, q, qd, tau, big=True)
    ydd = plan.forward_dynamics(t(q), t(qd), t(tau)).cpu().numpy()
    tid = plan.inverse_dynamics(t(q), t(qd), t(tau)).cpu().numpy()
    assert np.abs(ydd - ref).max() / (1 + np.abs(ref).max()) < 1e-9
    assert np.abs(tid - ref_id).max() / (1 + np.abs(ref_id).max()) < 1e-9


@pytest.mark.parametrize("which", ["two_parent", "exp_d10_l16", "imp_d10_l17", "three_loop_linkage"])
def test_external_forces_and_test_force_on_the_spanning_tree_route(gpu, which):
    """Plans on the spanning-tree route (a cluster on two parent bodies; clusters of 16 / 17 bodies; a cluster with six constraint rows) take world-frame external forces --
    the spanning model shares their bodies, so the forces enter ITS inverse dynamics -- and with them the contact-side entry points that
    are built on forced forward dynamics: applyTestForce (ClusterTreeDynamics.cpp:194-233) against the oracle's forced dynamics."""
    import torch
    from test_capi_cpu import two_parent_model

    big = which in ("exp_d10_l16", "imp_d10_l17")
    if which == "two_parent":
        blob = two_parent_model().serialize()
        plan = G.Plan(blob)
        q, qd, tau = random_states(blob, 70, config_index=4)
    else:
        name = {"exp_d10_l16": "parallel_chain_exp_d10_l16", "imp_d10_l17": "parallel_chain_imp_d10_l17"}.get(which, which)
        plan = G.Plan.from_urdf(os.path.join(ROBOT_MODELS, name + ".urdf"))
        blob = plan.blob
        q, qd, tau = valid_states(blob, 70, config_index=4, big=big, scale=0.5 if big else 1.0, max_cond=50)
    assert plan.info().spanning_tree_route == 1
    B, nb, nv = q.shape[0], plan.n_bodies, plan.nv
    fext = np.random.default_rng(6).uniform(-1, 1, (B, nb, 6))
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=gpu)
    ref = O.forward_dynamics(blob, q, qd, tau, fext, big=big)
    ref_t = O.inverse_dynamics(blob, q, qd, tau, fext, big=big)
    assert rel_err(O.forward_dynamics(blob, q, qd, tau, big=big), ref) > 1e-3  # (the forces matter)
    for dt, tol in ((torch.float64, TOL64), (torch.float32, TOL32)):
        got = plan.forward_dynamics(t(q, dt), t(qd, dt), t(tau, dt), f_ext=t(fext, dt)).double().cpu().numpy()
        got_t = plan.inverse_dynamics(t(q, dt), t(qd, dt), t(tau, dt), f_ext=t(fext, dt)).double().cpu().numpy()
        assert np.abs(got - ref).max() / (1 + np.abs(ref).max()) < tol
        assert np.abs(got_t - ref_t).max() / (1 + np.abs(ref_t).max()) < tol
    # applyTestForce at a point of the last body
    Bt = 6
    force = np.random.default_rng(7).uniform(-1, 1, (Bt, 3))
    offset = np.array([0.05, -0.02, 0.1])
    body = nb - 1
    lam, ds = plan.apply_test_force(t(q[:Bt]), body, offset, t(force))
    lam, ds = lam.cpu().numpy(), ds.cpu().numpy()
    Xa = plan.body_poses(t(q[:Bt])).cpu().numpy()[:, body]
    E, r = Xa[:, :9].reshape(Bt, 3, 3), Xa[:, 9:]
    pt = r + np.einsum("bji,j->bi", E, offset)
    fe = np.zeros((Bt, nb, 6))
    fe[:, body, :3] = np.cross(pt, force)
    fe[:, body, 3:] = force
    zero = np.zeros((Bt, nv))
    ds_ref = O.forward_dynamics(blob, q[:Bt], zero, zero, fe, big=big) - O.forward_dynamics(blob, q[:Bt], zero, zero, big=big)
    jtf = O.inverse_dynamics(blob, q[:Bt], zero, zero, big=big) - O.inverse_dynamics(blob, q[:Bt], zero, zero, fe, big=big)
    assert np.abs(ds - ds_ref).max() / (1 + np.abs(ds_ref).max()) < 1e-8
    assert np.abs(lam.reshape(-1) - np.einsum("bi,bi->b", jtf, ds_ref)).max() / (1 + np.abs(lam).max()) < 1e-8
    # spanning recovery qd_s = G yd, qdd_s = G ydd + g (big clusters: the wide constraint kernel): rates against the oracle's, the
    # acceleration's G against the rates' by linearity, g of the big cluster against the oracle's constraint evaluation
    vs, as0 = (x.cpu().numpy() for x in plan.spanning(t(q), t(qd), t(np.zeros_like(qd))))
    _, as1 = (x.cpu().numpy() for x in plan.spanning(t(q), t(qd), t(qd)))
    vs_ref = O.spanning_state(blob, q, qd, big=big)[1]
    assert np.abs(vs - vs_ref).max() / (1 + np.abs(vs_ref).max()) < 1e-10
    assert np.abs((as1 - as0) - vs).max() / (1 + np.abs(vs).max()) < 1e-10
    from generalized_rbda_amd.states import parse_clusters
    at = 0
    for ci, c in enumerate(parse_clusters(blob)["clusters"]):
        (pc, fb, k, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) = c
        if ctype >= 2:
            for b in range(4):
                g_ref = O.cluster_constraint(blob, ci, q[b], qd[b], nsv, nvel, rows, big=big)[1]
                assert np.abs(as0[b, at:at + nsv] - g_ref).max() / (1 + np.abs(g_ref).max()) < 1e-9
        else:
            assert np.abs(as0[:, at:at + nsv]).max() == 0
        at += nsv
    # inverse operational-space inertia of two frames (the unit-wrench route: 6 m + 1 forced dynamics per state), as
    # test_inverse_osim_by_force_propagation builds it from the oracle
    frames = [nb - 1, nb // 2]
    offs = np.random.default_rng(8).uniform(-0.2, 0.2, size=(2, 3))
    Linv, J = plan.inv_osim(t(q[:Bt]), frames, offs, with_jacobian=True)
    Linv, J = Linv.cpu().numpy(), J.cpu().numpy()
    Xall = plan.body_poses(t(q[:Bt])).cpu().numpy()
    tau0 = O.inverse_dynamics(blob, q[:Bt], zero, zero, big=big)
    J_ref = np.zeros((Bt, 12, nv))
    for c, (bd, off) in enumerate(zip(frames, offs)):
        E, r = Xall[:, bd, :9].reshape(Bt, 3, 3), Xall[:, bd, 9:]
        pt = r + np.einsum("bji,j->bi", E, off)
        for k in range(6):
            e = E[:, k % 3, :]
            fe = np.zeros((Bt, nb, 6))
            if k < 3:
                fe[:, bd, :3] = e
            else:
                fe[:, bd, :3] = np.cross(pt, e)
                fe[:, bd, 3:] = e
            J_ref[:, 6 * c + k] = tau0 - O.inverse_dynamics(blob, q[:Bt], zero, zero, fe, big=big)
    Hm = np.zeros((Bt, nv, nv))
    for j in range(nv):
        ej = np.zeros((Bt, nv)); ej[:, j] = 1
        Hm[:, :, j] = O.inverse_dynamics(blob, q[:Bt], zero, ej, big=big) - tau0
    L_ref = np.einsum("bik,bkl,bjl->bij", J_ref, np.linalg.inv(Hm), J_ref)
    assert np.abs(J - J_ref).max() / (1 + np.abs(J_ref).max()) < 1e-8
    assert np.abs(Linv - L_ref).max() / (1 + np.abs(L_ref).max()) < 1e-7
    # body twists (spanning rates from the wide constraint kernel, then the tree walk): velocities against J qd of the frames' bodies,
    # accelerations against central differences of the velocities along the motion (explicit clusters: y moves with yd, ydd)
    _, J0 = plan.inv_osim(t(q[:Bt]), frames, np.zeros((2, 3)), with_jacobian=True)
    J0 = J0.cpu().numpy().reshape(Bt, 2, 6, nv)
    ydd = ref[:Bt] * 0 + np.random.default_rng(9).uniform(-1, 1, (Bt, nv))
    V = plan.body_twists(t(q[:Bt]), t(qd[:Bt]), t(ydd)).cpu().numpy()
    assert np.abs(np.einsum("bnij,bj->bni", J0, qd[:Bt]) - V[:, frames, :6]).max() / (1 + np.abs(V[:, frames, :6]).max()) < 1e-9
    V32 = plan.body_twists(t(q[:Bt], torch.float32), t(qd[:Bt], torch.float32), t(ydd, torch.float32)).double().cpu().numpy()
    assert np.abs(V32 - V).max() / (1 + np.abs(V).max()) < TOL32
    if which not in ("imp_d10_l17", "three_loop_linkage"):
        h = 1e-5
        Vp = plan.body_twists(t(q[:Bt] + h * qd[:Bt] + 0.5 * h * h * ydd), t(qd[:Bt] + h * ydd), t(ydd)).cpu().numpy()
        Vm = plan.body_twists(t(q[:Bt] - h * qd[:Bt] + 0.5 * h * h * ydd), t(qd[:Bt] - h * ydd), t(ydd)).cpu().numpy()
        a_fd = (Vp[:, :, :6] - Vm[:, :, :6]) / (2 * h)
        Ew = Xall[:, :, :9].reshape(Bt, nb, 3, 3)
        a_ref = V[:, :, 6:].copy()
        a_ref[:, :, 3:] -= np.einsum("bnij,j->bni", Ew, -np.asarray(plan.get_gravity(), dtype=np.float64)[-3:])
        assert np.abs(a_fd - a_ref).max() / (1 + np.abs(a_ref).max()) < 2e-6
