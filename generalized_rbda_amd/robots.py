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


def tello() -> ClusterTreeModel:
    """Tello<double>::buildClusterTreeModel (src/Robots/Tello.cpp:6-277): floating torso and the two legs (hip clamp, hip
    differential, knee-ankle differential each)."""
    return tello_with_arms(arms=False)


def tello_with_arms(arms: bool = True) -> ClusterTreeModel:
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

    if not arms:
        return m
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


# -------------------------------------------------------------------------------------------------
# Hand-built robots that ALSO exist as URDF+ files.  The reference checks its URDF reader against these
# builders (UnitTests/testClusterTreeModel.cpp:100-114,146-229); restated here as parameter tables through
# the same construction API so that tests/test_urdf_vs_manual.py can do the same for the product's reader.
# -------------------------------------------------------------------------------------------------
def _sym3(rows):
    return np.array(rows, dtype=np.float64).reshape(3, 3)


def _inertia_lr(mass, com, I3, flip):
    """SpatialInertia(m, c, I) then withLeftRightSigns: the RIGHT side is the mirrored one
    (MIT_Humanoid.hpp:214-225, MiniCheetah.hpp:125-137)."""
    if flip:
        mass, com, I3 = _flip_y(mass, com, I3)
    return spatial_inertia(mass, np.asarray(com, dtype=np.float64), np.asarray(I3, dtype=np.float64))


def mini_cheetah(ori_repr: str = "quaternion") -> ClusterTreeModel:
    """MiniCheetah<double>::buildClusterTreeModel (src/Robots/MiniCheetah.cpp:6-139; parameters
    include/grbda/Robots/MiniCheetah.hpp:14-84)."""
    m = ClusterTreeModel(gravity=(0.0, 0.0, -9.81), ori_repr=ori_repr)
    I3 = np.eye(3)
    RY, RX = coordinate_rotation("y", np.pi / 2), coordinate_rotation("x", np.pi / 2)
    body_I = np.diag([11253, 36203, 42673]) * 1e-6
    abad_I = _sym3([381, 58, 0.45, 58, 560, 0.95, 0.45, 0.95, 444]) * 1e-6
    hip_I = _sym3([1983, 245, 13, 245, 2103, 1.5, 13, 1.5, 408]) * 1e-6
    knee_I_rot = np.diag([6, 248, 245]) * 1e-6
    rotor_z = np.diag([33, 33, 63]) * 1e-6
    rotor_x, rotor_y = RY @ rotor_z @ RY.T, RX @ rotor_z @ RX.T
    m_body, m_abad, m_hip, m_knee, m_rotor = 3.3, 0.54, 0.634, 0.064, 0.055
    com_abad, com_hip, com_knee = [0, 0.036, 0], [0, 0.016, -0.02], [0, 0, -0.061]
    abad_rotor_loc, abad_loc = np.array([0.125, 0.049, 0]), np.array([0.38, 0.098, 0]) * 0.5
    hip_loc, hip_rotor_loc = np.array([0, 0.062, 0]), np.array([0, 0.04, 0])
    knee_loc, knee_rotor_loc = np.array([0, 0, -0.209]), np.zeros(3)
    prefix = {0: "FR_", 1: "FL_", 2: "HR_", 3: "HL_"}
    sign = {0: (1, -1), 1: (1, 1), 2: (-1, -1), 3: (-1, 1)}  # withLegSigns (MiniCheetah.hpp:86-104)

    def leg_signs(v, leg):
        sx, sy = sign[leg]
        return np.array([sx * v[0], sy * v[1], v[2]])

    m.appendBody("Floating Base", spatial_inertia(m_body, [0, 0, 0], body_I), "ground", joint="free")
    side = -1
    RZ = coordinate_rotation("z", np.pi)
    for leg in (2, 3, 0, 1):
        p = prefix[leg]
        flip = side == -1
        m.registerBody(p + "abad_link", _inertia_lr(m_abad, com_abad, abad_I, flip), "Floating Base", I3, leg_signs(abad_loc, leg))
        m.registerBody(p + "abad_rotor", _inertia_lr(m_rotor, [0, 0, 0], rotor_x, flip), "Floating Base", I3, leg_signs(abad_rotor_loc, leg))
        m.appendRegisteredBodiesAsCluster(p + "abad", "RevoluteWithRotor", joint_axis="x", rotor_axis="x", gear_ratio=6.0)
        m.registerBody(p + "hip_link", _inertia_lr(m_hip, com_hip, hip_I, flip), p + "abad_link", RZ, leg_signs(hip_loc, leg))
        m.registerBody(p + "hip_rotor", _inertia_lr(m_rotor, [0, 0, 0], rotor_y, flip), p + "abad_link", RZ, leg_signs(hip_rotor_loc, leg))
        m.appendRegisteredBodiesAsCluster(p + "hip", "RevoluteWithRotor", joint_axis="y", rotor_axis="y", gear_ratio=6.0)
        m.registerBody(p + "knee_link", _inertia_lr(m_knee, com_knee, knee_I_rot, flip), p + "hip_link", I3, leg_signs(knee_loc, leg))
        m.registerBody(p + "knee_rotor", _inertia_lr(m_rotor, [0, 0, 0], rotor_y, flip), p + "hip_link", I3, leg_signs(knee_rotor_loc, leg))
        m.appendRegisteredBodiesAsCluster(p + "knee", "RevoluteWithRotor", joint_axis="y", rotor_axis="y", gear_ratio=9.33)
        side *= -1
    return m


# MIT humanoid parameters (include/grbda/Robots/MIT_Humanoid.hpp:16-66,71-150)
_MITH = dict(
    torso=(8.52, [0.009896, 0.004771, 0.100522],
           [0.172699, 0.001419, 0.004023, 0.001419, 0.105949, -0.001672, 0.004023, -0.001672, 0.091906]),
    hip_rz=(0.84563, [-0.064842, -0.000036, -0.063090],
            [0.0015373, 0.0000011, 0.0005578, 0.0000011, 0.0014252, 0.0000024, 0.0005578, 0.0000024, 0.0012028]),
    hip_rx=(1.20868, [0.067232, -0.013018, 0.0001831],
            [0.0017535, -0.0000063, -0.000080, -0.0000063, 0.003338, -0.000013, -0.000080, -0.000013, 0.0019927]),
    hip_ry=(2.64093, [0.0132054, 0.0269864, -0.096021],
            [0.0243761, 0.0000996, 0.0006548, 0.0000996, 0.0259015, 0.0026713, 0.0006548, 0.0026713, 0.0038929]),
    knee=(0.35435, [0.00528, 0.0014762, -0.13201],
          [0.003051, 0.000000, 0.0000873, 0.000000, 0.003033, 0.0000393, 0.0000873, 0.0000393, 0.0002529]),
    ankle=(0.280951, [0.022623, 0.0, -0.012826],
           [0.0000842, 0.000000, -0.0000488, 0.000000, 0.0007959, -0.000000, -0.0000488, -0.000000, 0.0007681]),
    shoulder_ry=(0.788506, [0.009265, 0.052623, -0.0001249],
                 [0.0013678, 0.0000266, 0.0000021, 0.0000266, 0.0007392, -0.0000012, 0.0000021, -0.0000012, 0.000884]),
    shoulder_rx=(0.80125, [0.0006041, 0.0001221, -0.082361],
                 [0.0011524, 0.0000007, 0.0000396, 0.0000007, 0.0011921, 0.0000014, 0.0000396, 0.0000014, 0.0012386]),
    shoulder_rz=(0.905588, [0.0001703, -0.016797, -0.060],
                 [0.0012713, 0.000001, -0.000008, 0.000001, 0.0017477, -0.0000225, -0.000008, -0.0000225, 0.0008191]),
    elbow=(0.34839, [-0.0059578, 0.000111, -0.0426735],
           [0.001570, 0.0000002, 0.0000335, 0.0000002, 0.0016167, 0.000003, 0.0000335, 0.000003, 0.0000619]),
)
_MITH_LOC = dict(
    hip_rz=([-0.00565, -0.082, -0.05735], [-0.00842837, -0.082, -0.041593]),
    hip_rx=([-0.06435, 0.0, -.07499], [-0.0827, 0.0, -0.066436]),
    hip_ry=([0.071, 0.0018375, 0.0], [0.071, 0.024, 0.0]),
    knee=([0.0, 0.0, -0.267], [0.013, -0.0497, -0.0178]),
    ankle=([0.0, 0.0, -0.2785], [.01563, -.0454, -.13354]),
    shoulder_ry=([0.01346, -0.17608, 0.24657], [0.01346, -0.16, 0.24657]),
    shoulder_rx=([0.0, -0.0575, 0.0], [0, -0.0575, 0]),
    shoulder_rz=([0.0, 0.0, -0.10250], [0., 0., -0.1025]),
    elbow=([0.0, 0.0, -0.1455], [0., 0.0325, -0.06]),
)
_MITH_PITCH = dict(hip_rz=-0.174533, hip_rx=0.436332, hip_ry=-(0.436332 + -0.174533))


def _mith_rotors():
    large_z = np.diag([3.443e-4, 3.443e-4, 5.548e-4])
    small_z = np.diag([1.084e-4, 1.084e-4, 1.6841e-4])
    RY, RX = coordinate_rotation("y", np.pi / 2), coordinate_rotation("x", -np.pi / 2)
    return dict(small_x=RY.T @ small_z @ RY, small_y=RX.T @ small_z @ RX, small_z=small_z,
                large_y=RX.T @ large_z @ RX, large_z=large_z)


def mit_humanoid(ori_repr: str = "quaternion") -> ClusterTreeModel:
    """MIT_Humanoid<double>::buildClusterTreeModel (src/Robots/MIT_Humanoid.cpp:6-358): floating torso, then
    right arm, right leg, left arm, left leg; legs end in a RevolutePairWithRotor knee/ankle cluster registered
    [ankle_rotor, knee_link, knee_rotor, ankle_link] (:172-179)."""
    m = ClusterTreeModel(gravity=(0.0, 0.0, -9.81), ori_repr=ori_repr)
    I3 = np.eye(3)
    rot = _mith_rotors()
    small_m, large_m = 0.05, 0.1
    torso = "Floating Base"
    m.appendBody(torso, spatial_inertia(_MITH["torso"][0], _MITH["torso"][1], _sym3(_MITH["torso"][2])), "ground", joint="free")

    def lr(v, side):  # withLeftRightSigns on a location: the LEFT side (1) mirrors y (MIT_Humanoid.hpp:186-198)
        return np.array([v[0], -v[1] if side == 1 else v[1], v[2]])

    def link(key, side):
        mass, com, I = _MITH[key]
        return _inertia_lr(mass, com, _sym3(I), side == 0)

    def rotor(mass, I, side):
        return _inertia_lr(mass, [0, 0, 0], I, side == 0)

    def rev_with_rotor(name, key, parent, side, axis, rotor_I, rotor_mass, ratio, pitch=None):
        pre = "right_" if side == 0 else "left_"
        E = I3 if pitch is None else coordinate_rotation("y", pitch)
        loc, rloc = _MITH_LOC[key]
        m.registerBody(pre + name + "_link", link(key, side), parent, E, lr(loc, side))
        m.registerBody(pre + name + "_rotor", rotor(rotor_mass, rotor_I, side), parent, E, lr(rloc, side))
        m.appendRegisteredBodiesAsCluster(pre + name, "RevoluteWithRotor", joint_axis=axis, rotor_axis=axis, gear_ratio=ratio)
        return pre + name + "_link"

    def leg(side):
        pre = "right_" if side == 0 else "left_"
        p = rev_with_rotor("hip_rz", "hip_rz", torso, side, "z", rot["small_z"], small_m, 6.0, _MITH_PITCH["hip_rz"])
        p = rev_with_rotor("hip_rx", "hip_rx", p, side, "x", rot["small_x"], small_m, 6.0, _MITH_PITCH["hip_rx"])
        p = rev_with_rotor("hip_ry", "hip_ry", p, side, "y", rot["large_y"], large_m, 6.0, _MITH_PITCH["hip_ry"])
        ar = m.registerBody(pre + "ankle_rotor", rotor(small_m, rot["small_y"], side), p, I3, lr(_MITH_LOC["ankle"][1], side))
        kl = m.registerBody(pre + "knee_link", link("knee", side), p, I3, lr(_MITH_LOC["knee"][0], side))
        kr = m.registerBody(pre + "knee_rotor", rotor(large_m, rot["large_y"], side), p, I3, lr(_MITH_LOC["knee"][1], side))
        al = m.registerBody(pre + "ankle_link", link("ankle", side), pre + "knee_link", I3, lr(_MITH_LOC["ankle"][0], side))
        m.appendRegisteredBodiesAsCluster(pre + "knee_and_ankle", "RevolutePairWithRotor", link1=kl, rotor1=kr, rotor2=ar,
                                          link2=al, joint_axes="yy", rotor_axes="yy", gear_ratios=[6.0, 6.0],
                                          belt_ratios_1=[2.0], belt_ratios_2=[2.0, 1.0])

    def arm(side):
        p = rev_with_rotor("shoulder_ry", "shoulder_ry", torso, side, "y", rot["small_y"], small_m, 6.0)
        p = rev_with_rotor("shoulder_rx", "shoulder_rx", p, side, "x", rot["small_x"], small_m, 6.0)
        p = rev_with_rotor("shoulder_rz", "shoulder_rz", p, side, "z", rot["small_z"], small_m, 6.0)
        rev_with_rotor("elbow", "elbow", p, side, "y", rot["small_y"], small_m, 9.0)

    arm(0)
    leg(0)
    arm(1)
    leg(1)
    return m


def mit_humanoid_no_rotors(ori_repr: str = "quaternion") -> ClusterTreeModel:
    """MIT_Humanoid_no_rotors<double>::buildClusterTreeModel (src/Robots/MIT_Humanoid_no_rotors.cpp:6-199): the MIT Humanoid's
    links with its parameters (the class derives from MIT_Humanoid), plain Revolute clusters, the knee and ankle links of a
    leg as one RevolutePair cluster (:112-113; G = 1), arm / leg / arm / leg as in the model with rotors (:192-195)."""
    m = ClusterTreeModel(gravity=(0.0, 0.0, -9.81), ori_repr=ori_repr)
    I3 = np.eye(3)
    torso = "Floating Base"
    m.appendBody(torso, spatial_inertia(_MITH["torso"][0], _MITH["torso"][1], _sym3(_MITH["torso"][2])), "ground", joint="free")

    def lr(v, side):
        return np.array([v[0], -v[1] if side == 1 else v[1], v[2]])

    def link(key, side):
        mass, com, I = _MITH[key]
        return _inertia_lr(mass, com, _sym3(I), side == 0)

    def rev(key, parent, side, axis, pitch=None):
        name = ("right_" if side == 0 else "left_") + key + "_link"
        E = I3 if pitch is None else coordinate_rotation("y", pitch)
        m.appendBody(name, link(key, side), parent, E, lr(_MITH_LOC[key][0], side), joint="revolute", axis=axis)
        return name

    def leg(side):
        pre = "right_" if side == 0 else "left_"
        p = rev("hip_rz", torso, side, "z", _MITH_PITCH["hip_rz"])
        p = rev("hip_rx", p, side, "x", _MITH_PITCH["hip_rx"])
        p = rev("hip_ry", p, side, "y", _MITH_PITCH["hip_ry"])
        m.registerBody(pre + "knee_link", link("knee", side), p, I3, lr(_MITH_LOC["knee"][0], side))
        m.registerBody(pre + "ankle_link", link("ankle", side), pre + "knee_link", I3, lr(_MITH_LOC["ankle"][0], side))
        m.appendRegisteredBodiesAsCluster(pre + "knee_ankle_cluster", "generic", axes="yy", G=np.eye(2), K=np.zeros((0, 2)))

    def arm(side):
        p = rev("shoulder_ry", torso, side, "y")
        p = rev("shoulder_rx", p, side, "x")
        p = rev("shoulder_rz", p, side, "z")
        rev("elbow", p, side, "y")

    arm(0)
    leg(0)
    arm(1)
    leg(1)
    return m


def mit_humanoid_leg() -> ClusterTreeModel:
    """MIT_Humanoid_Leg<double>::buildClusterTreeModel (src/Robots/MIT_Humanoid_Leg.cpp:6-163): the LEFT leg's
    parameters un-mirrored, fixed to the ground, massless rotors, knee/ankle cluster registered
    [knee_link, ankle_rotor, knee_rotor, ankle_link] (:131-138)."""
    m = ClusterTreeModel(gravity=(0.0, 0.0, -9.81))
    I3 = np.eye(3)
    rot = _mith_rotors()
    link = lambda key: spatial_inertia(_MITH[key][0], _MITH[key][1], _sym3(_MITH[key][2]))
    rotor = lambda I: spatial_inertia(0.0, [0, 0, 0], I)
    parent = "ground"
    for name, axis, rI in (("hip_rz", "z", rot["small_z"]), ("hip_rx", "x", rot["small_x"]), ("hip_ry", "y", rot["large_y"])):
        E = coordinate_rotation("y", _MITH_PITCH[name])
        m.registerBody(name + "_link", link(name), parent, E, _MITH_LOC[name][0])
        m.registerBody(name + "_rotor", rotor(rI), parent, E, _MITH_LOC[name][1])
        m.appendRegisteredBodiesAsCluster(name, "RevoluteWithRotor", joint_axis=axis, rotor_axis=axis, gear_ratio=6.0)
        parent = name + "_link"
    kl = m.registerBody("knee_link", link("knee"), parent, I3, _MITH_LOC["knee"][0])
    ar = m.registerBody("ankle_rotor", rotor(rot["small_y"]), parent, I3, _MITH_LOC["ankle"][1])
    kr = m.registerBody("knee_rotor", rotor(rot["large_y"]), parent, I3, _MITH_LOC["knee"][1])
    al = m.registerBody("ankle_link", link("ankle"), "knee_link", I3, _MITH_LOC["ankle"][0])
    m.appendRegisteredBodiesAsCluster("knee_and_ankle", "RevolutePairWithRotor", link1=kl, rotor1=kr, rotor2=ar, link2=al,
                                      joint_axes="yy", rotor_axes="yy", gear_ratios=[6.0, 6.0], belt_ratios_1=[2.0],
                                      belt_ratios_2=[2.0, 1.0])
    return m


def four_bar_rows(path1_link_lengths, path2_link_lengths, offset):
    """LoopConstraint::FourBar phi (src/Dynamics/ClusterJoints/FourBarJoint.cpp:22-49) as trig-polynomial rows:
    spanning coordinates (q0, q2) accumulate along path 1, (q1) along path 2;
    phi = sum_1 l_i [cos, sin](cumulative) - offset - sum_2 l_i [cos, sin](cumulative)."""
    if len(path1_link_lengths) + len(path2_link_lengths) != 3 or len(path1_link_lengths) != 2:
        raise RuntimeError("FourBar: Must contain 3 links (two on path 1, one on path 2)")
    rows = []
    for r, fn in enumerate(("cos", "sin")):
        terms = []
        w = [0.0, 0.0, 0.0]
        for i, coord in enumerate((0, 2)):
            w = list(w)
            w[coord] = 1.0
            terms.append((float(path1_link_lengths[i]), [(fn, w, 0.0)]))
        terms.append((-float(offset[r]), []))
        terms.append((-float(path2_link_lengths[0]), [(fn, [0.0, 1.0, 0.0], 0.0)]))
        rows.append(terms)
    return rows


def planar_leg_linkage() -> ClusterTreeModel:
    """PlanarLegLinkage<double>::buildClusterTreeModel (src/Robots/PlanarLegLinkage.cpp:5-93): a revolute thigh and
    a parallelogram four-bar lower leg [shank_driver, shank_support, foot] with LoopConstraint::FourBar."""
    m = ClusterTreeModel(gravity=(0.0, 0.0, -9.81))
    zz = lambda v: np.diag([0.0, 0.0, v])
    m.appendBody("thigh", spatial_inertia(0.23, [.0364, 0, 0], zz(45.389e-6)), "ground", np.eye(3), [0, 0, 0],
                 joint="revolute", axis="z")
    m.registerBody("shank_driver", spatial_inertia(.004, [.048, 0, 0], zz(3.257e-6)), "thigh", np.eye(3), [.011, 0, 0])
    m.registerBody("shank_support", spatial_inertia(0.225, [.04, 0, 0], zz(22.918e-6)), "thigh", np.eye(3), [.042, 0, 0])
    m.registerBody("foot", spatial_inertia(.017, [.0635, 0, 0], zz(22.176e-6)), "shank_driver", np.eye(3), [.096, 0, 0])
    p1 = [.096, .042 - .011]
    m.appendTrigPolyCluster("lower-leg-cluster", "zzz", [True, False, False], four_bar_rows(p1, [p1[0]], [p1[1], 0.0]))
    return m


# -------------------------------------------------------------------------------------------------
# JVRC-1 humanoid, hand-built (src/Robots/JVRC1_Humanoid.cpp:6-771; parameters include/grbda/Robots/JVRC1_Humanoid.hpp:16-275).
# Every joint of the hand-built robot is a RevoluteWithRotor cluster (gear ratio 6, one rotor design, rotor and link
# share their tree transform, all tree rotations are the identity); robot-models/jvrc1_humanoid.urdf -- the model
# BASELINE config 5 names -- describes the same links but gives no rotor to the neck (3 joints) and the wrists (4): 58
# bodies against 65.  `urdf_variant=True` builds the hand-built robot WITHOUT those seven rotors and with the URDF's link
# names, so that tests/test_urdf_vs_manual.py can pin the URDF reader's JVRC-1 (body data, coordinate order, dynamics)
# to the values the reference's header holds.
# -------------------------------------------------------------------------------------------------
_JVRC1_LINK = dict(   # design: (mass, CoM, diag of the rotational inertia at the CoM)      JVRC1_Humanoid.hpp:181-269
    pelvis=(10.0, [-0.01, 0.0, 0.034], [0.089583, 0.089583, 0.1125]),
    hip_p=(1.0, [0.0, 0.0, 0.0], [0.00196, 0.00196, 0.00196]),
    hip_r=(1.0, [0.0, 0.0, 0.0], [0.00196, 0.00196, 0.00196]),
    hip_y=(3.0, [0.01, 0.0, -0.22], [0.031925, 0.034525, 0.00865]),
    knee=(3.0, [0.04, 0.0, -0.16], [0.031925, 0.034525, 0.00865]),
    ankle_r=(1.0, [0.0, 0.0, 0.0], [0.00064, 0.00064, 0.00064]),
    ankle_p=(1.5, [0.03, 0.0, -0.07], [0.001417, 0.005617, 0.006217]),
    waist_y=(1.0, [0.0, 0.0, -0.07], [0.00173, 0.00173, 0.0032]),
    waist_p=(1.0, [0.0, 0.0, 0.0], [0.002425, 0.002425, 0.002425]),
    waist_r=(10.0, [0.02, 0.0, 0.24], [0.157083, 0.101083, 0.1367]),
    neck_y=(0.5, [0.0, 0.0, -0.05], [0.000729167, 0.000729167, 0.000625]),
    neck_r=(0.5, [0.0, 0.0, 0.0], [0.0005, 0.0005, 0.0005]),
    neck_p=(2.0, [0.01, 0.0, 0.11], [0.00968, 0.00968, 0.00968]),
    shoulder_p=(1.0, [0.0, 0.0, 0.0], [0.00196, 0.00196, 0.00196]),
    shoulder_r=(1.0, [0.0, 0.0, 0.0], [0.00196, 0.00196, 0.00196]),
    shoulder_y=(2.0, [-0.01, 0.0, -0.19], [0.01365, 0.0146, 0.00635]),
    elbow_p=(1.0, [-0.02, 0.0, -0.1], [0.010675, 0.010675, 0.0027]),
    elbow_y=(1.0, [0.0, 0.0, 0.0], [0.00064, 0.00064, 0.00064]),
    wrist_r=(0.5, [0.0, 0.0, 0.0], [0.00032, 0.00032, 0.00032]),
    left_wrist_y=(0.5, [0.0, -0.01, -0.06], [0.0004625, 0.0007625, 0.0004625]),
    right_wrist_y=(0.5, [0.0, 0.01, -0.06], [0.0004625, 0.0007625, 0.0004625]),
)
_JVRC1_ROTOR = (0.07, [0.0, 0.0, 0.0], [2e-5, 2e-5, 5e-5])   # JVRC1_Humanoid.hpp:176-179
_JVRC1_POS = dict(    # tree translations p_* (all R_* are the identity)                      JVRC1_Humanoid.hpp:19-172
    left_hip_p=[0.0, 0.096, 0.0], left_hip_r=[0.0, -2.77e-17, 0.0], left_hip_y=[0.0, -2.77e-17, 0.0],
    left_knee=[-0.02, -2.77e-17, -0.389], left_ankle_r=[0.04, 1.52e-16, -0.357], left_ankle_p=[-4.86e-17, 3.61e-16, -2.22e-16],
    right_hip_p=[0.0, -0.096, 0.0], right_hip_r=[0.0, 2.77e-17, 0.0], right_hip_y=[0.0, 2.77e-17, 0.0],
    right_knee=[-0.02, 2.77e-17, -0.389], right_ankle_r=[0.04, 1.11e-16, -0.357], right_ankle_p=[-1.734e-17, -4.44e-16, -1.11e-16],
    waist_y=[0.0, 0.0, 0.192], waist_p=[0.0, 0.0, -1.11e-16], waist_r=[0.0, 0.0, -1.11e-16],
    neck_y=[-0.003, 0.0, 0.453], neck_r=[8.24e-18, 0.0, 2.22e-16], neck_p=[8.24e-18, 0.0, 2.22e-16],
    left_shoulder_p=[0.0, 0.24, 0.33], left_shoulder_r=[0.0, 2.78e-17, -5.55e-16], left_shoulder_y=[0.0, 2.78e-17, -5.55e-16],
    left_elbow_p=[0.004, -2.78e-17, -0.305], left_elbow_y=[-0.004, 3.89e-16, -0.239], left_wrist_r=[0.0, -2.78e-16, -1.8e-16],
    left_wrist_y=[0.0, -2.78e-16, -1.8e-16],
    right_shoulder_p=[0.0, -0.24, 0.33], right_shoulder_r=[0.0, -2.78e-17, -7.77e-16], right_shoulder_y=[0.0, -2.78e-17, -7.77e-16],
    right_elbow_p=[0.004, 2.78e-17, -0.305], right_elbow_y=[-0.004, -5e-16, -0.239], right_wrist_r=[0.0, 2.78e-16, -1.8e-16],
    right_wrist_y=[0.0, 2.78e-16, -1.8e-16],
)
# (joint, parent, axis) in the order the reference appends the clusters (JVRC1_Humanoid.cpp:22-755); {s} = left / right
_JVRC1_TRUNK = [("waist_y", "pelvis", "z"), ("waist_p", "waist_y", "y"), ("waist_r", "waist_p", "x"),
                ("neck_y", "waist_r", "z"), ("neck_r", "neck_y", "x"), ("neck_p", "neck_r", "y")]
_JVRC1_LEG = [("hip_p", "pelvis", "y"), ("hip_r", "{s}_hip_p", "x"), ("hip_y", "{s}_hip_r", "z"), ("knee", "{s}_hip_y", "y"),
              ("ankle_r", "{s}_knee", "x"), ("ankle_p", "{s}_ankle_r", "y")]
_JVRC1_ARM = [("shoulder_p", "waist_r", "y"), ("shoulder_r", "{s}_shoulder_p", "x"), ("shoulder_y", "{s}_shoulder_r", "z"),
              ("elbow_p", "{s}_shoulder_y", "y"), ("elbow_y", "{s}_elbow_p", "z"), ("wrist_r", "{s}_elbow_y", "x")]
_JVRC1_NO_ROTOR_IN_URDF = ("neck_y", "neck_r", "neck_p", "wrist_r", "wrist_y")
# The ONE number on which the reference's two descriptions of the robot disagree: the URDF puts the right hip 87 mm below
# the pelvis frame (robot-models/jvrc1_humanoid.urdf:378,392: xyz="0.0 -0.096 -87e-03"), the header does not
# (JVRC1_Humanoid.hpp:52: p_right_hip_p = {0., -0.096, 0.}; the left hip is at z = 0 in both).  The URDF variant takes
# the file's value -- the file is what BASELINE config 5 names.
_JVRC1_URDF_POS = dict(right_hip_p=[0.0, -0.096, -87e-03])
_JVRC1_URDF_WORD = dict(p="pitch", r="roll", y="yaw")


def jvrc1_urdf_link_name(name: str) -> str:
    """'left_hip_p' -> 'left-hip-pitch', 'left_knee' -> 'left-knee' (the link names of robot-models/jvrc1_humanoid.urdf)."""
    parts = name.split("_")
    if parts[-1] in _JVRC1_URDF_WORD and len(parts) > 1:
        parts[-1] = _JVRC1_URDF_WORD[parts[-1]]
    return "-".join(parts)


def jvrc1_humanoid(urdf_variant: bool = False) -> ClusterTreeModel:
    """JVRC1_Humanoid::buildClusterTreeModel (src/Robots/JVRC1_Humanoid.cpp:6-771)."""
    m = ClusterTreeModel(gravity=(0.0, 0.0, -9.81))
    I3 = np.eye(3)
    nm = jvrc1_urdf_link_name if urdf_variant else (lambda s: s)
    si = lambda d: spatial_inertia(d[0], np.array(d[1], dtype=np.float64), np.diag(d[2]))
    m.appendBody(nm("pelvis"), si(_JVRC1_LINK["pelvis"]), "ground", joint="free")

    def joint(name, design, parent, axis):
        has_rotor = not (urdf_variant and any(name.endswith(t) for t in _JVRC1_NO_ROTOR_IN_URDF))
        pos = _JVRC1_URDF_POS.get(name, _JVRC1_POS[name]) if urdf_variant else _JVRC1_POS[name]
        if not has_rotor:
            m.appendBody(nm(name), si(_JVRC1_LINK[design]), nm(parent), I3, pos, joint="revolute", axis=axis)
            return
        m.registerBody(nm(name), si(_JVRC1_LINK[design]), nm(parent), I3, pos)
        m.registerBody(nm(name) + ("-rotor" if urdf_variant else "_rotor"), si(_JVRC1_ROTOR), nm(parent), I3, pos)
        m.appendRegisteredBodiesAsCluster(nm(name), "RevoluteWithRotor", joint_axis=axis, rotor_axis=axis, gear_ratio=6.0)

    for name, parent, axis in _JVRC1_TRUNK:
        joint(name, name, parent, axis)
    for table in (_JVRC1_LEG, _JVRC1_ARM):
        for s in ("left", "right"):
            for name, parent, axis in table:
                joint(f"{s}_{name}", name, parent.format(s=s), axis)
    for s in ("left", "right"):   # JVRC1_Humanoid.cpp:692-755: the wrist yaw joints come last
        joint(f"{s}_wrist_y", f"{s}_wrist_y", f"{s}_wrist_r", "z")
    return m


# -------------------------------------------------------------------------------------------------
# TeleopArm (src/Robots/TeleopArm.cpp, include/grbda/Robots/TeleopArm.hpp): fixed base, three RevoluteWithRotor clusters,
# one RevoluteTripleWithRotor cluster (upper link, wrist pitch, wrist roll and their three rotors on the shoulder link) and
# the gripper.  Values as the reference holds them (TeleopArm.cpp:167-266), its rotor arithmetic included: the "small" rotor
# inertia is 1e-3 times the ALREADY scaled large one (:173-177), and the X / Y variants are R I R^T with R = Ry(pi/2) / Rx(+pi/2).
# -------------------------------------------------------------------------------------------------
_TELEOP = {  # mass, com, rotational inertia rows (about the COM)
    "base": (0.996728196, [-1.32E-07, -0.001991858, 0.042209332],
             [[0.003411721, -2.78E-09, -1.45E-09], [-2.78E-09, 0.003047159, -0.000609866], [-1.45E-09, -0.000609866, 0.003412964]]),
    "shoulder-rx-link": (0.4796, [-4.96E-10, -0.004546817, 0.045690967],
                         [[0.001190777, 3.21E-12, 2.07E-11], [3.21E-12, 0.001028401, 0.00022441], [2.07E-11, 0.00022441, 0.001459421]]),
    "shoulder-ry-link": (1.670980082, [0.000482465, -0.000659857, 0.168192061],
                         [[0.016496934, 2.23E-07, 7.79E-05], [2.23E-07, 0.018757685, 0.000273553], [7.79E-05, 0.000273553, 0.003224214]]),
    "upper-link": (0.4617247, [3.99E-08, 0.003531916, 0.130185501],
                   [[0.006286029, -1.39E-10, -3.25E-09], [-1.39E-10, 0.006437035, 6.24E-06], [-3.25E-09, 6.24E-06, 0.00020455]]),
    "wrist-pitch-link": (0.063603259, [-1.70E-09, 0.004167466, 0.01683806],
                         [[1.73E-05, -2.13E-12, -3.11E-13], [-2.13E-12, 1.42E-05, -4.98E-07], [-3.11E-13, -4.98E-07, 2.03E-05]]),
    "wrist-roll-link": (0.167365855, [-0.001177177, 0.004697849, 0.040664076],
                        [[0.000182127, 2.15E-06, -1.57E-06], [2.15E-06, 8.28E-05, -9.44E-06], [-1.57E-06, -9.44E-06, 0.00011452]]),
    "gripper": (0.170943071, [0.000564355, -0.003238555, 0.105873754],
                [[6.20E-05, 5.26E-09, 1.04E-06], [5.26E-09, 6.64E-05, 8.69E-07], [1.04E-06, 8.69E-07, 1.81E-05]]),
}
_TELEOP_LOC = {"base": [0, 0, 0.051], "shoulder-rx-link": [0, 0, 0.106], "shoulder-ry-link": [0, 0, 0.071],
               "upper-link": [0, -0.0095, 0.3855], "wrist-pitch-link": [0, 0, 0.362], "wrist-roll-link": [0, 0.004, 0.03574],
               "gripper": [0.0004, 0.0375, 0.070995]}


def teleop_arm() -> ClusterTreeModel:
    """TeleopArm::buildClusterTreeModel (src/Robots/TeleopArm.cpp:6-165)."""
    m = ClusterTreeModel(gravity=(0.0, 0.0, -9.81))
    I3 = np.eye(3)
    large_z = 1e-3 * np.diag([1.052, 1.046, 1.811])
    small_z = 1e-3 * large_z
    RY, RX = coordinate_rotation("y", np.pi / 2), coordinate_rotation("x", np.pi / 2)
    rot = {("small", "x"): RY @ small_z @ RY.T, ("small", "y"): RX @ small_z @ RX.T, ("small", "z"): small_z,
           ("large", "x"): RY @ large_z @ RY.T, ("large", "y"): RX @ large_z @ RX.T, ("large", "z"): large_z}
    rotor = lambda size, axis: spatial_inertia(0.055 if size == "small" else 1.081, [0, 0, 0], rot[(size, axis)])
    link = lambda name: spatial_inertia(_TELEOP[name][0], _TELEOP[name][1], np.asarray(_TELEOP[name][2]))
    zero = [0.0, 0.0, 0.0]

    def rev_with_rotor(cluster, name, rotor_name, parent, axis, size, ratio):
        m.registerBody(name, link(name), parent, I3, _TELEOP_LOC[name])
        m.registerBody(rotor_name, rotor(size, axis), parent, I3, zero)
        m.appendRegisteredBodiesAsCluster(cluster, "RevoluteWithRotor", joint_axis=axis, rotor_axis=axis, gear_ratio=ratio)

    rev_with_rotor("base-cluster", "base", "base-rotor", "ground", "z", "large", 6.0)
    rev_with_rotor("shoulder-rx-cluster", "shoulder-rx-link", "shoulder-rx-rotor", "base", "x", "large", 6.0)
    rev_with_rotor("shoulder-ry-cluster", "shoulder-ry-link", "shoulder-ry-rotor", "shoulder-rx-link", "y", "large", 6.0)
    par = "shoulder-ry-link"
    for name in ("upper-link", "wrist-pitch-link", "wrist-roll-link"):
        m.registerBody(name, link(name), par, I3, _TELEOP_LOC[name])
        par = name
    for name, size, axis in (("elbow-rotor", "large", "y"), ("wrist-pitch-rotor", "small", "y"), ("wrist-roll-rotor", "small", "z")):
        m.registerBody(name, rotor(size, axis), "shoulder-ry-link", I3, zero)
    m.appendRegisteredBodiesAsCluster("upper-arm-cluster", "RevoluteTripleWithRotor", joint_axes="yyz", rotor_axes="yyz",
                                      gear_ratios=[6.0, 6.0, 6.0], belt_ratios_1=[1.0], belt_ratios_2=[1.0, 1.0],
                                      belt_ratios_3=[-1.0, 1.0, -1.0])
    rev_with_rotor("gripper-cluster", "gripper", "gripper-rotor", "wrist-roll-link", "x", "small", 2.0)
    return m

