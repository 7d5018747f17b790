"""CPU tests: the oracle (oracle/grbda_oracle.c) against
 (i)   golden vectors generated from the reference's own closed-form codegen
       (tests/golden/codegen_vectors.json, made by oracle/gen_golden.py) -- the reference's test
       UnitTests/testReflectedInertiaAlgos.cpp:144-222 uses tol 1e-5; fp64 gives ~1e-12;
 (ii)  the Projection identity on the spanning tree (testRigidBodyDynamicsAlgos.cpp:208-232, tol 5e-8);
 (iii) ID(FD(tau)) == tau (testRigidBodyDynamicsAlgos.cpp:221,235)."""
import json
import os

import numpy as np
import pytest

import oracle_py as O
from generalized_rbda_amd import modeldesc as md
from generalized_rbda_amd.states import random_states
from models import valid_states, zoo

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    with open(os.path.join(HERE, "golden", "codegen_vectors.json")) as f:
        return json.load(f)["cases"]


def _unhex(a):
    return np.array([[float.fromhex(v) for v in row] for row in a])


@pytest.mark.parametrize("case", _golden(), ids=lambda c: f"{c['family']}{c['n']}")
def test_oracle_matches_reference_codegen(case):
    build = md.revolute_chain_with_rotor if case["family"] == "rev" else md.revolute_pair_chain_with_rotor
    blob = build(case["n"]).serialize()
    y, yd, x = _unhex(case["y"]), _unhex(case["yd"]), _unhex(case["x"])
    fd = O.forward_dynamics(blob, y, yd, x)
    idd = O.inverse_dynamics(blob, y, yd, x)
    assert np.abs(fd - _unhex(case["FD"])).max() < 1e-9 * (1 + np.abs(_unhex(case["FD"])).max())
    assert np.abs(idd - _unhex(case["ID"])).max() < 1e-10 * (1 + np.abs(_unhex(case["ID"])).max())


@pytest.mark.parametrize("name,blob", list(zoo().items()), ids=list(zoo().keys()))
def test_oracle_cross_algorithm_identities(name, blob):
    q, qd, tau = valid_states(blob, 20, config_index=11)
    fd = O.forward_dynamics(blob, q, qd, tau)
    pj = O.forward_dynamics_projection(blob, q, qd, tau)
    scale = 1 + np.abs(fd).max()
    assert np.abs(fd - pj).max() < 5e-8 * scale, "cluster ABA vs Projection"
    back = O.inverse_dynamics(blob, q, qd, fd)
    assert np.abs(back - tau).max() < 5e-8 * scale, "ID(FD(tau)) != tau"


def test_oracle_external_forces_consistent():
    blob = zoo()["tree_mixed_float"]
    q, qd, tau = valid_states(blob, 8, config_index=12)
    nb = len(blob) and __import__("generalized_rbda_amd.states", fromlist=["parse_clusters"]).parse_clusters(blob)["nb"]
    fext = np.random.default_rng(3).uniform(-1, 1, (8, nb, 6))
    fd = O.forward_dynamics(blob, q, qd, tau, fext)
    pj = O.forward_dynamics_projection(blob, q, qd, tau, fext)
    assert np.abs(fd - pj).max() < 5e-8 * (1 + np.abs(fd).max())
    back = O.inverse_dynamics(blob, q, qd, fd, fext)
    assert np.abs(back - tau).max() < 5e-8 * (1 + np.abs(fd).max())
    assert np.abs(fd - O.forward_dynamics(blob, q, qd, tau)).max() > 1e-3, "external forces had no effect"


def test_oracle_mt_matches_single_thread():
    blob = zoo()["tree_rotor_float"]
    q, qd, tau = valid_states(blob, 257, config_index=13)
    assert np.array_equal(O.forward_dynamics(blob, q, qd, tau), O.forward_dynamics_mt(blob, q, qd, tau, 4))


def test_oracle_single_precision_build_tracks_the_checker():
    """oracle/_build/libgrbda_oracle_f32.so is the same source compiled with float arithmetic: bench.py's fp32 CPU
    baseline, never a checker.  It has to stay a valid statement of the algorithm: fp32-class agreement with the fp64
    build on the headline model and on a model with rotors and pairs."""
    for name in ("urdf_mit_humanoid", "tree_pair_float"):
        blob = zoo()[name]
        q, qd, tau = valid_states(blob, 130, config_index=19)
        a = O.forward_dynamics_mt(blob, q, qd, tau, 2)
        b = O.forward_dynamics_mt_f32(blob, q, qd, tau, 2).astype(np.float64)
        assert np.isfinite(b).all()
        err = np.abs(a - b).max(axis=1) / (1.0 + np.abs(a).max(axis=1))
        assert np.quantile(err, 0.99) < 1e-4


def test_extended_precision_build_of_the_oracle_agrees_with_the_fp64_checker():
    """oracle/_build/libgrbda_oracle_ld.so is the same source in x87 extended precision (the third evaluation of d ydd / d q near singular
    poses, tests/test_gpu_parity.py): on the golden-vector models it reproduces the fp64 checker to fp64 rounding, and its Newton
    projection of an implicit cluster lands closer to the manifold than fp64 can."""
    import generalized_rbda_amd.modeldesc as md
    from generalized_rbda_amd.states import random_states
    from models import zoo

    blob = md.revolute_pair_chain_with_rotor(4).serialize()
    q, qd, tau = random_states(blob, 16, 0)
    a = O.forward_dynamics(blob, q, qd, tau)
    b = O.forward_dynamics_ld(blob, q, qd, tau)
    assert b.dtype == np.longdouble and np.abs(a - b.astype(np.float64)).max() / (1 + np.abs(a).max()) < 1e-12
    fb = zoo()["urdf_four_bar"]
    q, qd, tau = random_states(fb, 64, 5)
    q64, ok64 = O.project_positions(fb, q)
    qld, okld = O.project_positions_ld(fb, q64)
    both = ok64 & okld
    assert both.sum() >= 32
    assert np.abs(qld.astype(np.float64)[both] - q64[both]).max() < 1e-9   # the same branch of the manifold, refined
