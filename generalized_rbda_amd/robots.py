"""Hand-built robots of the reference that have no URDF+ description with constraints.

``robot-models/tello_humanoid.urdf`` carries no ``<loop>`` / ``<coupling>`` element, so the
reference's Tello cluster model only exists as the C++ builders ``Tello`` / ``TelloWithArms``
(src/Robots/Tello.cpp:6-277, src/Robots/TelloWithArms.cpp:6-170; parameters
include/grbda/Robots/Tello.hpp:20-141, TelloWithArms.hpp:14-89).  This module restates that model
through the same construction API (``registerBody`` / ``appendRegisteredBodiesAsCluster``); the hip
and knee-ankle differentials, CasADi lambdas in the reference, become data: trig-polynomial
constraints (``GRBDA_CONSTRAINT_TRIG_POLY``).
"""
from __future__ import annotations

import numpy as np

from .modeldesc import ClusterTreeModel, coordinate_rotation, spatial_inertia


def _flip_y(mass, com, I3):
    """SpatialInertia::flipAlongAxis(Y) (include/grbda/Utils/SpatialInertia.h:233-245): mirror the body
    through the x-z plane -- COM y and the products of inertia involving y change sign."""
    com = np.array(com, dtype=np.float64)
    I3 = np.array(I3, dtype=np.float64)
    com[1] = -com[1]
    for i, j in ((0, 1), (1, 0), (1, 2), (2, 1)):
        I3[i, j] = -I3[i, j]
    return mass, com, I3


def _tello_hip_phi(N=6.0):
    """hip_diff_phi (Tello.cpp:139-155): q = [rotor1, rotor2, gimbal, thigh];
    ql_1 = q0, ql_2 = q1, y_1 = q2 / N, y_2 = q3 / N.  The trailing '3021 / 160000' is integer
    division in the reference (= 0) and is therefore absent."""
    ql1, ql2 = [1, 0, 0, 0], [0, 1, 0, 0]
    y1, y2 = [0, 0, 1 / N, 0], [0, 0, 0, 1 / N]
    S, C = "sin", "cos"

    def row(y, s_399, s_7a, s_7b):
        return [
            (57 / 2500, [(S, y, 0)]),
            (-49 / 5000, [(C, ql1, 0)]),
            (s_399 * 399 / 20000, [(S, ql1, 0)]),
            (-8 / 625, [(C, y, 0), (C, ql2, 0)]),
            (-57 / 2500, [(C, ql1, 0), (S, ql2, 0)]),
            (s_7a * 7 / 625, [(S, y, 0), (S, ql1, 0)]),
            (s_7b * 7 / 625, [(S, ql1, 0), (S, ql2, 0)]),
            (-8 / 625, [(C, ql1, 0), (S, y, 0), (S, ql2, 0)]),
        ]

    return [row(y1, -1, -1, +1), row(y2, +1, +1, -1)]


def _tello_knee_ankle_phi(N=6.0):
    """knee_ankle_diff_phi (Tello.cpp:237-252): q = [rotor1, rotor2, shin, foot]; the reference uses the
    literal 3.1415 for pi and an integer division '163349 / 6250000' (= 0)."""
    pi = 3.1415
    d = [0, 0, 0.5 / N, -0.5 / N]           # y_1 / 2 - y_2 / 2
    dq = [0, 1, 0.5 / N, -0.5 / N]          # ... + ql_2
    ql2 = [0, 1, 0, 0]
    S, C = "sin", "cos"
    row0 = [
        (21 / 6250, [(C, d, 1979 * pi / 4500)]),
        (-13 / 625, [(C, d, 493 * pi / 1500)]),
        (-273 * np.cos(pi / 9) / 12500, []),
        (-7 / 2500, [(S, dq, 231 * pi / 500)]),
        (91 / 5000, [(S, ql2, 2 * pi / 15)]),
        (-147 / 50000, [(S, ql2, pi / 45)]),
    ]
    row1 = [(1.0, [("lin", [1, 0, -0.5 / N, -0.5 / N], 0)])]  # ql_1 - y_2/2 - y_1/2
    return [row0, row1]


def tello_with_arms() -> ClusterTreeModel:
    """TelloWithArms<double>::buildClusterTreeModel (TelloWithArms.cpp:6-170 on top of Tello.cpp:6-277)."""
    m = ClusterTreeModel(gravity=(0.0, 0.0, -9.81))
    I3 = np.eye(3)
    R_down = np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1.0]])
    R_left = np.array([[-1, 0, 0], [0, 0, 1], [0, 1, 0.0]])
    R_right = np.array([[1, 0, 0], [0, 0, 1], [0, -1, 0.0]])
    sym = lambda a, b, c, d, e, f: np.array([[a, b, c], [b, d, e], [c, e, f]], dtype=np.float64)

    torso = spatial_inertia(2.3008, [0.0073, -0.0013, -0.0023], sym(0.0366, 0., -0.0006, 0.0142, -0.0002, 0.0291))
    hip_clamp = spatial_inertia(1.3289, [-0.0010, 0., -0.0069], sym(0.0032, 0., 0.0001, 0.0033, 0., 0.0027))
    gimbal = spatial_inertia(0.4433, [-0.0027, 0., 0.0258], sym(0.0018, 0., 0., 0.0017, 0., 0.0015))
    thigh = spatial_inertia(1.5424, [0.003, -0.0001, -0.0323], sym(0.0103, 0., -0.0005, 0.0097, 0., 0.0027))
    shin = spatial_inertia(0.3072, [0.0047, -0.0003, -0.1043], sym(0.0054, -0., -0.0002, 0.0054, 0., 0.0001))
    foot = spatial_inertia(0.1025, [0.0042, -0., -0.0251],
                           sym(0.094e-3, -0., -0.0038e-3, 0.1773e-3, 0., 0.0901e-3))
    rotor = spatial_inertia(0.07, [0, 0, 0], np.diag([2.5984e-5, 2.5984e-5, 5.1512e-5]))
    gear = 6.0

    m.appendBody("torso", torso, "ground", joint="free")
    for side, sy in (("left", 1.0), ("right", -1.0)):
        # hip clamp + rotor (Tello.cpp:36-75)
        m.registerBody(f"{side}-hip-clamp", hip_clamp, "torso", I3, [0., sy * 126e-3, -87e-3])
        m.registerBody(f"{side}-hip-clamp-rotor", rotor, "torso", R_down, [0., sy * 126e-3, -26e-3])
        m.appendRegisteredBodiesAsCluster(f"{side}-hip-clamp", "RevoluteWithRotor", joint_axis="z", rotor_axis="z",
                                          gear_ratio=gear)
        # hip differential (Tello.cpp:77-163): [rotor1, rotor2, gimbal, thigh], axes Z Z X Y
        m.registerBody(f"{side}-hip-rotor-1", rotor, f"{side}-hip-clamp", R_left, [0., 0.04, 0.])
        m.registerBody(f"{side}-hip-rotor-2", rotor, f"{side}-hip-clamp", R_right, [0., -0.04, 0.])
        m.registerBody(f"{side}-gimbal", gimbal, f"{side}-hip-clamp", I3, [0., 0., -142.5e-3])
        m.registerBody(f"{side}-thigh", thigh, f"{side}-gimbal", I3, [0., 0., 0.])
        m.appendTrigPolyCluster(f"{side}-hip-differential", "zzxy", [True, True, False, False], _tello_hip_phi())
        # knee-ankle differential (Tello.cpp:165-261): [rotor1, rotor2, shin, foot], axes Z Z Y Y
        m.registerBody(f"{side}-knee-ankle-rotor-1", rotor, f"{side}-thigh", R_right, [0., 26.55e-3, 0.])
        m.registerBody(f"{side}-knee-ankle-rotor-2", rotor, f"{side}-thigh", R_left, [0., -26.55e-3, 0.])
        m.registerBody(f"{side}-shin", shin, f"{side}-thigh", I3, [0., 0., -226.8e-3])
        m.registerBody(f"{side}-foot", foot, f"{side}-shin", I3, [0., 0., -260e-3])
        m.appendTrigPolyCluster(f"{side}-knee-ankle-differential", "zzyy", [True, True, False, False],
                                _tello_knee_ankle_phi())

    # arms (TelloWithArms.cpp:13-164, TelloWithArms.hpp:14-89); armID 0 = left, 1 = right (mirrored in y)
    rotor_z = np.diag([1.084e-4, 1.084e-4, 1.6841e-4])
    RY, RX = coordinate_rotation("y", np.pi / 2), coordinate_rotation("x", -np.pi / 2)
    rotor_x, rotor_y = RY.T @ rotor_z @ RY, RX.T @ rotor_z @ RX
    links = [
        ("shoulder-ry", "shoulder-ry-rotor", "y", 0.788506, [0.009265, 0.052623, -0.0001249],
         sym(0.0013678, 0.0000266, 0.0000021, 0.0007392, -0.0000012, 0.000884), rotor_y,
         [0.01346, 0.17608, 0.24657], [0.01346, 0.16, 0.24657], 6.0),
        ("shoulder-rx", "shoulder-rx-rotor", "x", 0.80125, [0.0006041, 0.0001221, -0.082361],
         sym(0.0011524, 0.0000007, 0.0000396, 0.0011921, 0.0000014, 0.0012386), rotor_x,
         [0.0, 0.0575, 0.0], [0, 0.0575, 0], 6.0),
        ("shoulder-rz-link", "shoulder-rz-rotor", "z", 0.905588, [0.0001703, -0.016797, -0.060],
         sym(0.0012713, 0.000001, -0.000008, 0.0017477, -0.0000225, 0.0008191), rotor_z,
         [0.0, 0.0, -0.10250], [0., 0., -0.1025], 6.0),
        ("elbow-link", "elbow-rotor", "y", 0.34839, [-0.0059578, 0.000111, -0.0426735],
         sym(0.001570, 0.0000002, 0.0000335, 0.0016167, 0.000003, 0.0000619), rotor_y,
         [0.0, 0.0, -0.1455], [0., -0.0325, -0.06], 9.0),
    ]
    for arm, (side, sy) in enumerate((("left", 1.0), ("right", -1.0))):
        parent = "torso"
        for link_name, rotor_name, axis, mass, com, Ic, Irot, loc, rloc, ratio in links:
            lm, lc, lI = (mass, np.array(com), Ic) if arm == 0 else _flip_y(mass, com, Ic)
            rm, rc, rI = (0.0, np.zeros(3), Irot) if arm == 0 else _flip_y(0.0, np.zeros(3), Irot)
            mirror = lambda v: [v[0], sy * v[1], v[2]]
            m.registerBody(f"{side}-{link_name}", spatial_inertia(lm, lc, lI), parent, I3, mirror(loc))
            m.registerBody(f"{side}-{rotor_name}", spatial_inertia(rm, rc, rI), parent, I3, mirror(rloc))
            m.appendRegisteredBodiesAsCluster(f"{side}-{link_name}", "RevoluteWithRotor", joint_axis=axis,
                                              rotor_axis=axis, gear_ratio=ratio)
            parent = f"{side}-{link_name}"
    return m
