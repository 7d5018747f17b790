"""Synthetic random states with the reference's sampling law.

* revolute coordinates, velocities, torques: i.i.d. U(-1, 1)
  (ClusterJoints::Base::randomJointState, src/Dynamics/ClusterJoints/ClusterJoint.cpp:73-80);
* floating base: position U(-1,1)^3, orientation = quaternion of RPY ~ U(-1,1)^3
  (FreeJoint.cpp:48-60, OrientationRepresentation.h:24-28), velocity U(-1,1)^6;
* implicit-loop clusters: independent positions U(-1,1), dependent guess U(-0.1,0.1); the caller
  projects them onto phi(q) = 0 (GenericJoint.cpp:289-348).
Counter-based RNG (Philox), seed = 0x67726264 + config_index.
"""
from __future__ import annotations

import struct

import numpy as np

from .modeldesc import C_FREE, C_LOOP_POSITION, C_STATIC, C_TRIG_POLY, ORI_QUATERNION, rotmat_to_quat, rpy_to_rotmat

SEED_BASE = 0x67726264


def rpy_to_quat_batch(rpy: np.ndarray) -> np.ndarray:
    """Vectorised ori::rpyToQuat = rotationMatrixToQuaternion(rpyToRotMat(rpy))
    (OrientationTools.h:121-130,160-200,292-300) for an [B,3] array."""
    B = rpy.shape[0]
    sx, cx = np.sin(rpy[:, 0]), np.cos(rpy[:, 0])
    sy, cy = np.sin(rpy[:, 1]), np.cos(rpy[:, 1])
    sz, cz = np.sin(rpy[:, 2]), np.cos(rpy[:, 2])
    one, zero = np.ones(B), np.zeros(B)
    Rx = np.stack([one, zero, zero, zero, cx, sx, zero, -sx, cx], axis=1).reshape(B, 3, 3)
    Ry = np.stack([cy, zero, -sy, zero, one, zero, sy, zero, cy], axis=1).reshape(B, 3, 3)
    Rz = np.stack([cz, sz, zero, -sz, cz, zero, zero, zero, one], axis=1).reshape(B, 3, 3)
    r = np.transpose(Rx @ Ry @ Rz, (0, 2, 1))  # rotationMatrixToQuaternion works on the transpose
    tr = r[:, 0, 0] + r[:, 1, 1] + r[:, 2, 2]
    q = np.empty((B, 4))
    c0 = tr > 0
    c1 = ~c0 & (r[:, 0, 0] > r[:, 1, 1]) & (r[:, 0, 0] > r[:, 2, 2])
    c2 = ~c0 & ~c1 & (r[:, 1, 1] > r[:, 2, 2])
    c3 = ~c0 & ~c1 & ~c2
    with np.errstate(invalid="ignore"):
        S = np.sqrt(np.maximum(tr + 1.0, 0)) * 2.0
        q[c0] = np.stack([0.25 * S, (r[:, 2, 1] - r[:, 1, 2]) / S, (r[:, 0, 2] - r[:, 2, 0]) / S,
                          (r[:, 1, 0] - r[:, 0, 1]) / S], axis=1)[c0]
        S = np.sqrt(np.maximum(1.0 + r[:, 0, 0] - r[:, 1, 1] - r[:, 2, 2], 0)) * 2.0
        q[c1] = np.stack([(r[:, 2, 1] - r[:, 1, 2]) / S, 0.25 * S, (r[:, 0, 1] + r[:, 1, 0]) / S,
                          (r[:, 0, 2] + r[:, 2, 0]) / S], axis=1)[c1]
        S = np.sqrt(np.maximum(1.0 + r[:, 1, 1] - r[:, 0, 0] - r[:, 2, 2], 0)) * 2.0
        q[c2] = np.stack([(r[:, 0, 2] - r[:, 2, 0]) / S, (r[:, 0, 1] + r[:, 1, 0]) / S, 0.25 * S,
                          (r[:, 1, 2] + r[:, 2, 1]) / S], axis=1)[c2]
        S = np.sqrt(np.maximum(1.0 + r[:, 2, 2] - r[:, 0, 0] - r[:, 1, 1], 0)) * 2.0
        q[c3] = np.stack([(r[:, 1, 0] - r[:, 0, 1]) / S, (r[:, 0, 2] + r[:, 2, 0]) / S,
                          (r[:, 1, 2] + r[:, 2, 1]) / S, 0.25 * S], axis=1)[c3]
    return q


def parse_clusters(blob: bytes):
    """Minimal reader of the cluster table of a model-description blob."""
    magic, version, nb, nc, nq, nv, ori, n_ints, n_dbls, n_names = struct.unpack_from("<II8i", blob, 0)
    off = 96 + 416 * nb
    clusters = [struct.unpack_from("<16i", blob, off + 64 * c) for c in range(nc)]
    off += 64 * nc
    ints = np.frombuffer(blob, dtype="<i4", count=n_ints, offset=off)
    return dict(nb=nb, nc=nc, nq=nq, nv=nv, ori=ori, clusters=clusters, ints=ints)


def random_states(blob: bytes, B: int, config_index: int = 0, dtype=np.float64):
    """Returns q[B,nq], qd[B,nv], tau[B,nv] (tau doubles as ydd for inverse dynamics)."""
    m = parse_clusters(blob)
    rng = np.random.Generator(np.random.Philox(SEED_BASE + config_index))
    q = rng.uniform(-1.0, 1.0, size=(B, m["nq"]))
    qd = rng.uniform(-1.0, 1.0, size=(B, m["nv"]))
    tau = rng.uniform(-1.0, 1.0, size=(B, m["nv"]))
    for c in m["clusters"]:
        (pc, fb, k, qi, npos, vi, nvel, nsp, nsv, ctype, rows, io, ni, do, nd, _) = c
        if ctype == C_FREE and m["ori"] == ORI_QUATERNION:
            q[:, qi + 3: qi + 7] = rpy_to_quat_batch(q[:, qi + 3: qi + 6].copy())
        elif ctype in (C_LOOP_POSITION, C_TRIG_POLY):
            ind = m["ints"][io + 1: io + 1 + nsv] if ctype == C_LOOP_POSITION else m["ints"][io: io + nsv]
            for j in range(nsv):
                if not ind[j]:
                    q[:, qi + j] *= 0.1
    return q.astype(dtype), qd.astype(dtype), tau.astype(dtype)
