"""Test infrastructure: numpy statement (one state) of the analytic first-order derivatives of cluster inverse dynamics that
`deriv_kernels.hip` implements, checked against central differences of the CPU oracle.  Explicit (constant G) models.

All spatial quantities are expressed in ONE inertial frame F that coincides with the floating base at this instant, so
composite quantities add without transforms.  With S_j the joint axis, Sd_j = v_j x S_j, Pd_j = v_parent x S_j,
Pdd_j = a_parent x S_j + v_parent x Pd_j and, per body, B_i = (v x*) I - I (v x) + (I v) xbar*:
    j <= k (j ancestor or equal):  dtau_k/dq_j  = Pd_j . (Bc_k^T S_k) + Pdd_j . (Ic_k S_k)
                                   dtau_k/dqd_j = S_j . (Bc_k^T S_k) + (Sd_j + Pd_j) . (Ic_k S_k)
    k <  j:                        dtau_k/dq_j  = S_k . (S_j x* Fc_j + Bc_j Pd_j + Ic_j Pdd_j)
                                   dtau_k/dqd_j = S_k . (Bc_j S_j + Ic_j (Sd_j + Pd_j))
(Ic, Bc, Fc: sums over the subtree), then  d tau_y / d y = G^T (.) G.
usage: python tests/deriv_recursion_numpy.py"""
import os, struct, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from generalized_rbda_amd.modeldesc import coordinate_rotation, quat_to_rotmat, rpy_to_rotmat, skew


def parse(blob):
    magic, version, nb, nc, nq, nv, ori, n_ints, n_dbls, n_names = struct.unpack_from("<II8i", blob, 0)
    grav = np.array(struct.unpack_from("<6d", blob, 48))
    bodies = []
    off = 96
    for b in range(nb):
        parent, cluster, sub, jt, axis = struct.unpack_from("<5i", blob, off)
        d = np.frombuffer(blob, dtype="<f8", count=48, offset=off + 32)
        bodies.append(dict(parent=parent, cluster=cluster, sub=sub, jtype=jt, axis=axis, E=d[:9].reshape(3, 3), r=d[9:12],
                           I=d[12:].reshape(6, 6)))
        off += 416
    clusters = [struct.unpack_from("<16i", blob, off + 64 * c) for c in range(nc)]
    off += 64 * nc
    ints = np.frombuffer(blob, dtype="<i4", count=n_ints, offset=off)
    off += 4 * (n_ints + (n_ints & 1))
    dbls = np.frombuffer(blob, dtype="<f8", count=n_dbls, offset=off)
    return dict(nb=nb, nc=nc, nq=nq, nv=nv, ori=ori, grav=grav, bodies=bodies, clusters=clusters, ints=ints, dbls=dbls)


def crm(v):
    return np.block([[skew(v[:3]), np.zeros((3, 3))], [skew(v[3:]), skew(v[:3])]])


def crf(v):
    return -crm(v).T


def xbar(f):
    """(f xbar*) m = m x* f"""
    return np.block([[-skew(f[:3]), -skew(f[3:])], [-skew(f[3:]), np.zeros((3, 3))]])


def Xmot(E, r):
    return np.block([[E, np.zeros((3, 3))], [-E @ skew(r), E]])


def span_maps(m):
    """G (n_span x nv) block diagonal, position map q_span = Gq y for explicit clusters; base handled apart."""
    return None


def rnea_derivs(m, q, qd, ydd):
    nb, nv = m["nb"], m["nv"]
    bodies, clusters = m["bodies"], m["clusters"]
    # spanning coordinates
    qs, qds, qdds, Grow = [None] * nb, [None] * nb, [None] * nb, [None] * nb
    base = None
    for c, cl in enumerate(clusters):
        pc, fb, k, qi, npos, vi, n, nsp, nsv, ctype = cl[:10]
        if ctype == 1:
            base = (fb, qi, vi)
            continue
        assert ctype == 0, "explicit clusters only"
        G = m["dbls"][cl[13]:cl[13] + k * n].reshape(k, n)
        for i in range(k):
            Grow[fb + i] = (vi, G[i])
            qs[fb + i] = G[i] @ q[qi:qi + n]
            qds[fb + i] = G[i] @ qd[vi:vi + n]
            qdds[fb + i] = G[i] @ ydd[vi:vi + n]
    # frame F = base frame (or the world for fixed-base models)
    XF = [None] * nb      # motion transform F -> body
    v, a, S, Sd, Pd, Pdd = ([None] * nb for _ in range(6))
    a0 = -m["grav"]
    if base is not None:
        fb, qi, vi = base
        pos = q[qi:qi + 3]
        R = quat_to_rotmat(q[qi + 3:qi + 7]) if m["ori"] == 0 else rpy_to_rotmat(q[qi + 3:qi + 6])
        a0 = Xmot(R, pos) @ a0
        XF[fb] = np.eye(6)
        v[fb] = qd[vi:vi + 6].copy()
        a[fb] = a0 + ydd[vi:vi + 6]
    for b in range(nb):
        bd = bodies[b]
        if bd["jtype"] == 1:
            continue
        E = coordinate_rotation(bd["axis"], qs[b]) @ bd["E"]
        Xup = Xmot(E, bd["r"])
        p = bd["parent"]
        Xp = XF[p] if p >= 0 else np.eye(6)
        vp = v[p] if p >= 0 else np.zeros(6)
        ap = a[p] if p >= 0 else a0
        XF[b] = Xup @ Xp
        s = np.zeros(6); s[bd["axis"]] = 1.0
        S[b] = np.linalg.solve(XF[b], s)          # axis in F coordinates
        v[b] = vp + S[b] * qds[b]
        Sd[b] = crm(v[b]) @ S[b]
        Pd[b] = crm(vp) @ S[b]
        Pdd[b] = crm(ap) @ S[b] + crm(vp) @ Pd[b]
        a[b] = ap + S[b] * qdds[b] + Sd[b] * qds[b]
    # body quantities in F, composites
    Ic, Bc, Fc = [None] * nb, [None] * nb, [None] * nb
    for b in range(nb):
        IF = XF[b].T @ bodies[b]["I"] @ XF[b]
        h = IF @ v[b]
        Ic[b] = IF.copy()
        Bc[b] = crf(v[b]) @ IF - IF @ crm(v[b]) + xbar(h)
        Fc[b] = IF @ a[b] + crf(v[b]) @ h
    for b in range(nb - 1, -1, -1):
        p = bodies[b]["parent"]
        if p >= 0:
            Ic[p] += Ic[b]; Bc[p] += Bc[b]; Fc[p] += Fc[b]
    tau = np.zeros(nv)
    dq = np.zeros((nv, nv)); dqd = np.zeros((nv, nv))
    def cols(b):
        """(coordinate indices, weights, S, Sd, Pd, Pdd) of the joint columns of body b"""
        if bodies[b]["jtype"] == 1:
            vi = base[2]
            out = []
            for k6 in range(6):
                e = np.zeros(6); e[k6] = 1.0
                out.append((np.array([vi + k6]), np.array([1.0]), e, crm(v[b]) @ e, np.zeros(6), crm(a0) @ e))
            return out
        vi, g = Grow[b]
        return [(vi + np.arange(len(g)), g, S[b], Sd[b], Pd[b], Pdd[b])]
    for k in range(nb):
        for (ik, gk, Sk, Sdk, Pdk, Pddk) in cols(k):
            tau[ik] += gk * (Sk @ Fc[k])
            t1 = Bc[k].T @ Sk
            t2 = Ic[k] @ Sk
            t3 = Bc[k] @ Sk + Ic[k] @ (Sdk + Pdk)
            t4 = crf(Sk) @ Fc[k] + Bc[k] @ Pdk + Ic[k] @ Pddk
            # j <= k: k itself (every column of its joint) and the strict ancestors
            j = k
            while j >= 0:
                for (ij, gj, Sj, Sdj, Pdj, Pddj) in cols(j):
                    dq[np.ix_(ik, ij)] += np.outer(gk, gj) * (Pdj @ t1 + Pddj @ t2)
                    dqd[np.ix_(ik, ij)] += np.outer(gk, gj) * (Sj @ t1 + (Sdj + Pdj) @ t2)
                    if j != k:
                        dq[np.ix_(ij, ik)] += np.outer(gj, gk) * (Sj @ t4)
                        dqd[np.ix_(ij, ik)] += np.outer(gj, gk) * (Sj @ t3)
                j = bodies[j]["parent"]
    if base is not None and m["ori"] == 1:
        # roll-pitch-yaw base: the reference's tangent step is plain q + dq there (position in world coordinates, then the
        # three angles), so the base's body-twist columns [rotation; translation] are mapped through
        # d(twist) / d(pos, rpy) = [[0, T], [R, 0]], T = body angular velocity per unit rate of (roll, pitch, yaw)
        fb, qi, vi = base
        r, p_ = q[qi + 3], q[qi + 4]
        sx, cx, sy, cy = np.sin(r), np.cos(r), np.sin(p_), np.cos(p_)
        T = np.array([[1, 0, -sy], [0, cx, sx * cy], [0, -sx, cx * cy]])
        R = rpy_to_rotmat(q[qi + 3:qi + 6])
        J6 = np.block([[np.zeros((3, 3)), T], [R, np.zeros((3, 3))]])
        dq[:, vi:vi + 6] = dq[:, vi:vi + 6] @ J6
    return tau, dq, dqd


def plus(m, q, dq_tan):
    """tangent step of the reference's derivative test (UnitTests/testHelpers.hpp:50-112)"""
    out = q.copy()
    for cl in m["clusters"]:
        pc, fb, k, qi, npos, vi, n = cl[:7]
        if cl[9] == 1 and m["ori"] == 0:
            R = quat_to_rotmat(q[qi + 3:qi + 7])
            out[qi:qi + 3] += R.T @ dq_tan[vi + 3:vi + 6]
            e = q[qi + 3:qi + 7]
            w = dq_tan[vi:vi + 3]
            dquat = 0.5 * np.array([-e[1] * w[0] - e[2] * w[1] - e[3] * w[2], e[0] * w[0] + e[2] * w[2] - e[3] * w[1],
                                    e[0] * w[1] - e[1] * w[2] + e[3] * w[0], e[0] * w[2] + e[1] * w[1] - e[2] * w[0]])
            out[qi + 3:qi + 7] += dquat
        else:
            out[qi:qi + n] += dq_tan[vi:vi + n]
    return out


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    from models import zoo, valid_states
    z = zoo()
    names = sys.argv[1:] or ["urdf_mini_cheetah", "urdf_mit_humanoid", "tree_rotor_float", "chain_tree_a", "urdf_revolute_rotor_chain"]
    for name in names:
        blob = z[name] if isinstance(z[name], (bytes, bytearray)) else z[name].serialize()
        m = parse(blob)
        q, qd, x = valid_states(blob, 2, 3)
        for s in range(2):
            tau, dq, dqd = rnea_derivs(m, q[s], qd[s], x[s])
            ref = O.inverse_dynamics(blob, q[s:s + 1], qd[s:s + 1], x[s:s + 1])[0]
            h = 1e-6
            fdq = np.zeros_like(dq); fdqd = np.zeros_like(dqd)
            for j in range(m["nv"]):
                e = np.zeros(m["nv"]); e[j] = h
                tp = O.inverse_dynamics(blob, plus(m, q[s], e)[None], qd[s:s + 1], x[s:s + 1])[0]
                tm = O.inverse_dynamics(blob, plus(m, q[s], -e)[None], qd[s:s + 1], x[s:s + 1])[0]
                fdq[:, j] = (tp - tm) / (2 * h)
                tp = O.inverse_dynamics(blob, q[s:s + 1], (qd[s] + e)[None], x[s:s + 1])[0]
                tm = O.inverse_dynamics(blob, q[s:s + 1], (qd[s] - e)[None], x[s:s + 1])[0]
                fdqd[:, j] = (tp - tm) / (2 * h)
            sc = lambda A: np.abs(A).max() + 1e-300
            print(f"{name:28s} tau err {np.abs(tau - ref).max() / sc(ref):.2e}  dq err {np.abs(dq - fdq).max() / sc(fdq):.2e}"
                  f"  dqd err {np.abs(dqd - fdqd).max() / sc(fdqd):.2e}")
