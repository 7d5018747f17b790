"""The inverse-dynamics derivative recursion behind `deriv_kernels.hip`, stated in numpy (tests/deriv_recursion_numpy.py),
against central differences of the CPU oracle along the reference's tangent step (UnitTests/testHelpers.hpp:50-112;
the reference validates its own CasADi derivatives the same way, testRigidBodyDynamicsAlgosDerivatives.cpp:271-383).
Runs without a GPU: it pins the MATH of the kernel -- common-frame composites, the free base's body-twist columns, the
projection with the clusters' G -- independently of the HIP implementation, which tests/test_gpu_parity.py checks."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
sys.path.insert(0, HERE)

import oracle_py as O  # noqa: E402
import deriv_recursion_numpy as P  # noqa: E402
from models import valid_states, zoo  # noqa: E402


@pytest.mark.parametrize("name", ["urdf_mini_cheetah", "urdf_revolute_rotor_chain", "tree_pair_float", "tree_triple_fixed",
                                  "tree_generic_float", "rev_pair_rotor_chain_4", "urdf_mini_cheetah_rpy", "chain_tree_rpy"])
def test_recursion_matches_differences_of_the_oracle(name):
    blob = zoo()[name]
    m = P.parse(blob)
    q, qd, ydd = valid_states(blob, 1, 11)
    tau, dq, dqd = P.rnea_derivs(m, q[0], qd[0], ydd[0])
    ref = O.inverse_dynamics(blob, q, qd, ydd)[0]
    assert np.abs(tau - ref).max() / (1.0 + np.abs(ref).max()) < 1e-12
    h = 1e-6
    nv = m["nv"]
    fdq, fdqd = np.zeros((nv, nv)), np.zeros((nv, nv))
    for j in range(nv):
        e = np.zeros(nv)
        e[j] = h
        fdq[:, j] = (O.inverse_dynamics(blob, P.plus(m, q[0], e)[None], qd, ydd)[0]
                     - O.inverse_dynamics(blob, P.plus(m, q[0], -e)[None], qd, ydd)[0]) / (2 * h)
        fdqd[:, j] = (O.inverse_dynamics(blob, q, qd + e, ydd)[0] - O.inverse_dynamics(blob, q, qd - e, ydd)[0]) / (2 * h)
    assert np.abs(dq - fdq).max() / (1.0 + np.abs(fdq).max()) < 1e-7
    assert np.abs(dqd - fdqd).max() / (1.0 + np.abs(fdqd).max()) < 1e-7
