"""The Tello differentials' constraint functions, as the SOURCE EXPRESSIONS of the reference (src/Robots/Tello.cpp:139-155
hip_diff_phi, :237-252 knee_ankle_diff_phi), against the trig-polynomial TERM TABLES the model description carries
(generalized_rbda_amd/robots.py, _tello_hip_phi / _tello_knee_ankle_phi) as the oracle evaluates them -- so the tables
are checked against the reference's formulas, not only against themselves.  The expressions below are transcribed
operator for operator, including the two constants that are INTEGER divisions in C++ (3021 / 160000 and
163349 / 6250000 are int / int = 0) and the literal 3.1415 for pi; hand-computed values at q = 0 anchor both.
K = d phi / d q is checked by central differences of the same expressions."""
import math

import numpy as np

import oracle_py as O
from generalized_rbda_amd.robots import tello_with_arms
from generalized_rbda_amd.states import parse_clusters
from generalized_rbda_amd.modeldesc import C_TRIG_POLY

sin, cos = math.sin, math.cos


def hip_diff_phi(q):
    """Tello.cpp:139-155"""
    N = 6.0
    ql_1, ql_2, y_1, y_2 = q[0], q[1], q[2] / N, q[3] / N
    out0 = ((57 * sin(y_1)) / 2500 - (49 * cos(ql_1)) / 5000 - (399 * sin(ql_1)) / 20000 - (8 * cos(y_1) * cos(ql_2)) / 625
            - (57 * cos(ql_1) * sin(ql_2)) / 2500 - (7 * sin(y_1) * sin(ql_1)) / 625 + (7 * sin(ql_1) * sin(ql_2)) / 625
            - (8 * cos(ql_1) * sin(y_1) * sin(ql_2)) / 625 + 3021 // 160000)
    out1 = ((57 * sin(y_2)) / 2500 - (49 * cos(ql_1)) / 5000 + (399 * sin(ql_1)) / 20000 - (8 * cos(y_2) * cos(ql_2)) / 625
            - (57 * cos(ql_1) * sin(ql_2)) / 2500 + (7 * sin(y_2) * sin(ql_1)) / 625 - (7 * sin(ql_1) * sin(ql_2)) / 625
            - (8 * cos(ql_1) * sin(y_2) * sin(ql_2)) / 625 + 3021 // 160000)
    return np.array([out0, out1])


def knee_ankle_diff_phi(q):
    """Tello.cpp:237-252"""
    N = 6.0
    ql_1, ql_2, y_1, y_2 = q[0], q[1], q[2] / N, q[3] / N
    out0 = ((21 * cos(y_1 / 2 - y_2 / 2 + (1979 * 3.1415) / 4500)) / 6250 - (13 * cos(y_1 / 2 - y_2 / 2 + (493 * 3.1415) / 1500)) / 625
            - (273 * cos(3.1415 / 9)) / 12500 - (7 * sin(y_1 / 2 - y_2 / 2 + ql_2 + (231 * 3.1415) / 500)) / 2500
            + (91 * sin(ql_2 + (2 * 3.1415) / 15)) / 5000 - (147 * sin(ql_2 + 3.1415 / 45)) / 50000 + 163349 // 6250000)
    out1 = ql_1 - y_2 / 2 - y_1 / 2
    return np.array([out0, out1])


def test_hand_computed_anchor_values():
    # q = 0: hip rows -49/5000 - 8/625 = -0.0226 each (all sines vanish, the integer-division constant is 0)
    assert np.allclose(hip_diff_phi(np.zeros(4)), [-0.0226, -0.0226], atol=1e-15)
    p = 3.1415
    ka0 = (21 * cos(1979 * p / 4500) / 6250 - 13 * cos(493 * p / 1500) / 625 - 273 * cos(p / 9) / 12500
           - 7 * sin(231 * p / 500) / 2500 + 91 * sin(2 * p / 15) / 5000 - 147 * sin(p / 45) / 50000)
    assert np.allclose(knee_ankle_diff_phi(np.zeros(4)), [ka0, 0.0], atol=1e-15)
    # the constant the integer division drops is exactly what would put the zero pose on phi = 0 (to the 3.1415-vs-pi error):
    assert abs(ka0 + 163349 / 6250000) < 2e-6 and abs(ka0) > 0.026
    assert abs(hip_diff_phi(np.zeros(4))[0] + 3021 / 160000) < 4e-3   # likewise for the hip rows (-0.0226 + 0.0189)


def test_term_tables_reproduce_the_source_expressions():
    blob = tello_with_arms().serialize()
    m = parse_clusters(blob)
    rng = np.random.default_rng(7)
    implicit = [(ci, cl) for ci, cl in enumerate(m["clusters"]) if cl[9] == C_TRIG_POLY]
    assert len(implicit) == 4
    for n, (ci, cl) in enumerate(implicit):
        (pc, fb, k, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) = cl
        fn = hip_diff_phi if n % 2 == 0 else knee_ankle_diff_phi   # left hip, left knee-ankle, right hip, right knee-ankle
        for _ in range(50):
            q = rng.uniform(-1.5, 1.5, m["nq"])
            qd = np.zeros(m["nv"])
            _, _, K, _, phi = O.cluster_constraint(blob, ci, q, qd, nsv, nvel, rows)
            qs = q[qi: qi + 4]
            assert np.abs(phi - fn(qs)).max() < 1e-15, f"cluster {ci}: phi"
            h = 1e-6
            Kfd = np.stack([(fn(qs + h * e) - fn(qs - h * e)) / (2 * h) for e in np.eye(4)], axis=1)
            assert np.abs(K - Kfd).max() < 1e-9, f"cluster {ci}: K = d phi / d q"
