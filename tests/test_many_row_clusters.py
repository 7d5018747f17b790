"""Implicit clusters with four to six constraint rows (several position loops that share links; tests/planar_linkages.py).

The structured kernels keep three constraint rows per cluster in registers; a cluster with more goes through the spanning-tree route
(the wide kernels of manifold_kernels.hip with room for kMaxConstraintRows = 6 rows, K_d inverted by Gauss-Jordan elimination
with partial pivoting), as GenericJoint.cpp:17-90 / 289-385 treats any number of rows through Eigen's dense solvers.  The GPU tests run
the bodies of the zoo's own parity tests on these models: forward / inverse dynamics, mass matrix and derivative identities, Newton
projection, spanning recovery, state input, body poses and twists, all against the oracle."""
import os

import numpy as np
import pytest

import generalized_rbda_amd as G
from generalized_rbda_amd.states import parse_clusters
from models import ROBOT_MODELS
from planar_linkages import linkage_urdf

MODELS = {"watt_six_bar": (1, 5, 4), "three_loop_linkage": (2, 7, 6)}  # name -> (dyads, bodies, constraint rows)


def _blob(name):
    return G.urdf_to_blob(os.path.join(ROBOT_MODELS, name + ".urdf"))


@pytest.mark.parametrize("name", sorted(MODELS))
def test_linkages_with_many_rows_compile_onto_the_spanning_tree_route(name, tmp_path):
    dyads, bodies, rows = MODELS[name]
    path = tmp_path / "l.urdf"
    path.write_text(linkage_urdf(dyads, name))
    blob = G.urdf_to_blob(str(path))
    assert blob == _blob(name)  # (the committed file is the generator's output)
    (cl,) = parse_clusters(blob)["clusters"]
    assert cl[2] == bodies and cl[6] == 1 and cl[9] == 2 and cl[10] == rows
    plan = G.Plan(blob)
    assert plan.nv == 1 and plan.n_bodies == bodies
    assert plan.info().spanning_tree_route == 1


def test_more_than_six_rows_are_refused_with_a_message(tmp_path, monkeypatch):
    path = tmp_path / "l.urdf"
    path.write_text(linkage_urdf(3))
    blob = G.urdf_to_blob(str(path))
    assert parse_clusters(blob)["clusters"][0][10] == 8
    with pytest.raises(G.GrbdaError, match="constraint rows"):
        G.Plan(blob)
    # and without the spanning-tree route the structured kernels' limit of three rows is what is reported
    monkeypatch.setenv("GRBDA_NO_PROJECTION", "1")
    with pytest.raises(G.GrbdaError, match="constraint rows"):
        G.Plan(_blob("watt_six_bar"))


# ---- GPU: the zoo's parity tests, on these models ------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODELS))
def test_dynamics_match_oracle(name, gpu):
    import test_gpu_parity as P

    blob = _blob(name)
    P.test_aba_and_rnea_fp64_match_oracle(name, blob, gpu)
    P.test_aba_and_rnea_fp32_match_oracle(name, blob, gpu)
    P.test_mass_matrix_bias_and_fd_derivatives_match_oracle(name, blob, gpu)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODELS))
def test_constraint_side_entry_points_match_oracle(name, gpu):
    import torch
    import oracle_py as O
    import test_gpu_parity as P
    from models import valid_states

    blob = _blob(name)
    P.test_newton_projection_matches_oracle(name, blob, gpu)
    P.test_spanning_recovery_matches_oracle(name, blob, gpu)
    P.test_body_poses_match_oracle(name, blob, gpu)
    P.test_body_twists_are_the_derivatives_of_the_motion(name, blob, gpu)
    plan = G.Plan(blob)
    q, qd, _ = valid_states(blob, 300, config_index=5)
    _, _, gm, kc = O.spanning_state(blob, q, qd)
    gmax, kcond, status = plan.constraint_gain(torch.as_tensor(q, dtype=torch.float64, device=gpu))
    assert int(status.abs().sum().item()) == 0
    assert np.abs(gmax.cpu().numpy() - gm).max() <= 1e-8 * (1 + gm.max())
    assert (np.abs(kcond.cpu().numpy() - kc) / kc).max() <= 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODELS))
def test_state_input_conventions(name, gpu, monkeypatch):
    import test_state_input_gpu as S

    monkeypatch.setitem(S.ZOO, name, _blob(name))
    S.test_spanning_and_independent_states_give_the_same_dynamics(name, gpu)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MODELS))
def test_derivatives_on_the_manifold_match_oracle_differences(name, gpu):
    """d ydd / d q, d qd, d tau of the one independent coordinate: central differences of the oracle's forward dynamics along the manifold
    (the dependent angles re-projected), as test_position_derivative_matches_oracle_differences takes them."""
    import torch
    import oracle_py as O
    from models import valid_states
    from test_gpu_parity import _reference_plus_on_manifold

    blob = _blob(name)
    plan = G.Plan(blob)
    m = parse_clusters(blob)
    B, h = 64, 1e-5
    q, qd, tau = valid_states(blob, B, config_index=41, max_cond=50.0)
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=gpu)
    d = plan.fd_derivatives(t(q), t(qd), t(tau))
    ref = {k: np.empty((B, 1, 1)) for k in ("dq", "dqd", "dtau")}
    for b in range(B):
        qp = _reference_plus_on_manifold(blob, m, q[b], 0, +h)[None]
        qm = _reference_plus_on_manifold(blob, m, q[b], 0, -h)[None]
        fd = lambda qq, v, x: O.forward_dynamics(blob, qq, v, x)[0]
        ref["dq"][b, :, 0] = (fd(qp, qd[b:b + 1], tau[b:b + 1]) - fd(qm, qd[b:b + 1], tau[b:b + 1])) / (2 * h)
        ref["dqd"][b, :, 0] = (fd(q[b:b + 1], qd[b:b + 1] + h, tau[b:b + 1]) - fd(q[b:b + 1], qd[b:b + 1] - h, tau[b:b + 1])) / (2 * h)
        ref["dtau"][b, :, 0] = (fd(q[b:b + 1], qd[b:b + 1], tau[b:b + 1] + 1.0) - fd(q[b:b + 1], qd[b:b + 1], tau[b:b + 1] - 1.0)) / 2.0
    for k in ref:
        got = d[k].cpu().numpy()
        assert np.abs(got - ref[k]).max() / (1.0 + np.abs(ref[k]).max()) < 2e-5, k
    d32 = plan.fd_derivatives(t(q, torch.float32), t(qd, torch.float32), t(tau, torch.float32))
    c32 = lambda a: a.astype(np.float32).astype(np.float64)
    d_of_32 = plan.fd_derivatives(t(c32(q)), t(c32(qd)), t(c32(tau)))
    for k in ref:
        a, b_ = d32[k].double().cpu().numpy(), d_of_32[k].cpu().numpy()
        assert np.abs(a - b_).max() / (1.0 + np.abs(b_).max()) < 1e-3, k
