"""The product's URDF+ reader (csrc/urdf.cpp, behind grbda_plan_from_urdf) against an INDEPENDENT reading of the same files: plain
xml.etree, thirty lines, no code shared with the reader or with the hand-built robots of tests/test_urdf_vs_manual.py.  GPU parity compares
product and oracle on the SAME model description, so an error of the reader would be invisible there (review of round 4, soft spot 1a); this
test pins every body of every robot file under tests/golden/robot-models/ -- parent, tree transform, spatial inertia, joint axis -- to what the
file says under the reference's conventions:
  * Xtree = (E, r) with E = quaternionToRotationMatrix(pose.rotation) = R^T for the URDF rotation R = Rz(yaw) Ry(pitch) Rx(roll), r = <origin xyz>
    (src/Utils/SpatialTransforms.cpp:17-23, include/grbda/Utils/OrientationTools.h:251-269);
  * spatial inertia from (mass, <inertial><origin xyz> = COM, the 3 x 3 inertia as written -- the rotation of the inertial frame is ignored)
    (include/grbda/Utils/SpatialInertia.h:74-82,105-116);
  * joint axis: the coordinate axis the <axis> vector names, its sign dropped (include/grbda/Utils/OrientationTools.h:70-93)."""
import glob
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

import generalized_rbda_amd as G
from models import ROBOT_MODELS
from test_urdf_vs_manual import unpack


def _floats(s, n=3):
    v = [float(x) for x in (s or "0 0 0").split()]
    assert len(v) == n
    return np.array(v)


def _rot(rpy):
    r, p, y = rpy
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def read_urdf(path):
    """{link name: dict(parent, E, r, I 6 x 6, axis index or None, type)} for every link that hangs off a joint"""
    root = ET.parse(path).getroot()
    inertial = {}
    for link in root.findall("link"):
        ine = link.find("inertial")
        if ine is None:
            inertial[link.get("name")] = np.zeros((6, 6))
            continue
        m = float(ine.find("mass").get("value"))
        org = ine.find("origin")
        c = _floats(org.get("xyz")) if org is not None else np.zeros(3)
        t = ine.find("inertia")
        I3 = np.array([[float(t.get("ixx")), float(t.get("ixy")), float(t.get("ixz"))],
                       [float(t.get("ixy")), float(t.get("iyy")), float(t.get("iyz"))],
                       [float(t.get("ixz")), float(t.get("iyz")), float(t.get("izz"))]])
        S = np.zeros((6, 6))
        S[:3, :3] = I3 + m * _skew(c) @ _skew(c).T
        S[:3, 3:] = m * _skew(c)
        S[3:, :3] = m * _skew(c).T
        S[3:, 3:] = m * np.eye(3)
        inertial[link.get("name")] = S
    out = {}
    for j in root.findall("joint"):
        child, parent = j.find("child").get("link"), j.find("parent").get("link")
        org = j.find("origin")
        xyz = _floats(org.get("xyz")) if org is not None else np.zeros(3)
        rpy = _floats(org.get("rpy")) if org is not None and org.get("rpy") else np.zeros(3)
        ax = j.find("axis")
        axis = None
        if ax is not None:
            a = _floats(ax.get("xyz"))
            axis = int(np.argmax(np.abs(a)))
            assert abs(abs(a[axis]) - 1.0) < 1e-12 and np.abs(np.delete(a, axis)).max() < 1e-12, f"{path}: axis {a} is not a coordinate axis"
        out[child] = dict(parent=parent, E=_rot(rpy).T, r=xyz, I=inertial[child], axis=axis, type=j.get("type"))
    return out


URDFS = sorted(glob.glob(os.path.join(ROBOT_MODELS, "*.urdf")))


@pytest.mark.parametrize("path", URDFS, ids=[os.path.basename(p)[:-5] for p in URDFS])
def test_reader_agrees_with_an_independent_reading_of_the_file(path):
    plan = G.Plan.from_urdf(path)   # host-side: the reader and the plan compiler need no GPU
    m = unpack(plan.blob)
    ref = read_urdf(path)
    bodies = {b["name"]: b for b in m["bodies"]}
    assert len(bodies) == len(m["bodies"]), "body names are unique"
    # every body of the model is a link of the file that hangs off a joint, and every such link is a body
    assert set(bodies) == set(ref), (sorted(set(bodies) ^ set(ref)))
    names = [b["name"] for b in m["bodies"]]
    for name, b in bodies.items():
        r = ref[name]
        parent = names[b["parent"]] if b["parent"] >= 0 else None
        # the root of the file ("world" / "base" without inertia) is the ground of the model
        assert parent == (r["parent"] if r["parent"] in bodies else None), name
        assert np.abs(b["E"] - r["E"]).max() < 1e-12, f"{name}: tree rotation"
        assert np.abs(b["r"] - r["r"]).max() < 1e-12, f"{name}: tree translation"
        assert np.abs(b["I"] - r["I"]).max() < 1e-12 * (1.0 + np.abs(r["I"]).max()), f"{name}: spatial inertia"
        if r["type"] in ("revolute", "continuous"):
            assert b["axis"] == r["axis"], f"{name}: joint axis"
