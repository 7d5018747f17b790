"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/_build/libgrbda_oracle.so and
oracle/_ref/libgrbda_codegen_ref.so.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_int, c_longlong, c_size_t, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "_build", "libgrbda_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libgrbda_codegen_ref.so")

BIG_LIB = os.path.join(_HERE, "_build", "libgrbda_oracle_big.so")  # the same source with MAXK = MAXN = 48 (oracle/Makefile)

_lib = None
_lib_big = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)


def lib(big=False):
    """big: the build with room for clusters of up to 48 bodies / 48 independent coordinates (tests of the spanning-tree
    route for clusters beyond the HIP kernels' structured limits)."""
    global _lib, _lib_big
    if big:
        if _lib_big is None:
            if not os.path.exists(BIG_LIB):
                build()
            L = ctypes.CDLL(BIG_LIB)
            for name in ("grbda_oracle_forward_dynamics", "grbda_oracle_inverse_dynamics",
                         "grbda_oracle_forward_dynamics_projection"):
                getattr(L, name).argtypes = [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]
            L.grbda_oracle_cluster_constraint.argtypes = [c_void_p, c_size_t, c_int] + [c_void_p] * 7
            L.grbda_oracle_project_positions.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_void_p]
            _lib_big = L
        return _lib_big
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            build()
        L = ctypes.CDLL(ORACLE_LIB)
        for name in ("grbda_oracle_forward_dynamics", "grbda_oracle_inverse_dynamics",
                     "grbda_oracle_forward_dynamics_projection"):
            getattr(L, name).argtypes = [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]
        L.grbda_oracle_forward_dynamics_mt.argtypes = [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                                       c_size_t, c_int]
        L.grbda_oracle_cluster_constraint.argtypes = [c_void_p, c_size_t, c_int] + [c_void_p] * 7
        L.grbda_oracle_project_positions.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_void_p]
        L.grbda_oracle_body_poses.argtypes = [c_void_p, c_size_t, c_void_p, c_void_p, c_size_t]
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _run(fn, blob, q, qd, x, f_ext=None):
    q, qd, x = _f64(q), _f64(qd), _f64(x)
    out = np.empty_like(x)
    fe = None if f_ext is None else _f64(f_ext)
    rc = fn(blob, len(blob), q.ctypes.data, qd.ctypes.data, x.ctypes.data, None if fe is None else fe.ctypes.data,
            out.ctypes.data, q.shape[0])
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return out


def forward_dynamics(blob, q, qd, tau, f_ext=None, big=False):
    return _run(lib(big).grbda_oracle_forward_dynamics, blob, q, qd, tau, f_ext)


def inverse_dynamics(blob, q, qd, ydd, f_ext=None, big=False):
    return _run(lib(big).grbda_oracle_inverse_dynamics, blob, q, qd, ydd, f_ext)


def forward_dynamics_projection(blob, q, qd, tau, f_ext=None, big=False):
    return _run(lib(big).grbda_oracle_forward_dynamics_projection, blob, q, qd, tau, f_ext)


def forward_dynamics_mt(blob, q, qd, tau, n_threads):
    q, qd, tau = _f64(q), _f64(qd), _f64(tau)
    out = np.empty_like(tau)
    rc = lib().grbda_oracle_forward_dynamics_mt(blob, len(blob), q.ctypes.data, qd.ctypes.data, tau.ctypes.data,
                                                out.ctypes.data, q.shape[0], n_threads)
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return out


_lib32 = None


def forward_dynamics_mt_f32(blob, q, qd, tau, n_threads):
    """The same restatement compiled in single precision (_build/libgrbda_oracle_f32.so): the fp32 CPU baseline of
    bench.py.  Not a parity checker."""
    global _lib32
    if _lib32 is None:
        path = os.path.join(_HERE, "_build", "libgrbda_oracle_f32.so")
        if not os.path.exists(path):
            build()
        _lib32 = ctypes.CDLL(path)
        _lib32.grbda_oracle_forward_dynamics_mt.argtypes = [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                                            c_size_t, c_int]
    q, qd, tau = (np.ascontiguousarray(a, dtype=np.float32) for a in (q, qd, tau))
    out = np.empty_like(tau)
    rc = _lib32.grbda_oracle_forward_dynamics_mt(blob, len(blob), q.ctypes.data, qd.ctypes.data, tau.ctypes.data,
                                                 out.ctypes.data, q.shape[0], n_threads)
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return out


def cluster_constraint(blob, cluster, q, qd, nsv, n, rows, big=False):
    q, qd = _f64(q), _f64(qd)
    G, g = np.zeros((nsv, n)), np.zeros(nsv)
    K, k, phi = np.zeros((rows, nsv)), np.zeros(rows), np.zeros(rows)
    rc = lib(big).grbda_oracle_cluster_constraint(blob, len(blob), cluster, q.ctypes.data, qd.ctypes.data, G.ctypes.data,
                                               g.ctypes.data, K.ctypes.data, k.ctypes.data, phi.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return G, g, K, k, phi


def body_poses(blob, q, n_bodies):
    q = _f64(q)
    out = np.zeros((q.shape[0], n_bodies, 12))
    rc = lib().grbda_oracle_body_poses(blob, len(blob), q.ctypes.data, out.ctypes.data, q.shape[0])
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return out


def spanning_state(blob, q, qd, big=False):
    """(q_span[B, sum n_span_pos], qd_span[B, sum n_span_vel], gmax[B], kcond[B]) -- toSpanningTreeState of every cluster."""
    import struct
    q, qd = _f64(q), _f64(qd)
    nb, nc = struct.unpack_from("<ii", blob, 8)
    off = 96 + 416 * nb
    nsp = sum(struct.unpack_from("<16i", blob, off + 64 * c)[7] for c in range(nc))
    nsv = sum(struct.unpack_from("<16i", blob, off + 64 * c)[8] for c in range(nc))
    B = q.shape[0]
    qs, vs, gm, kc = np.zeros((B, nsp)), np.zeros((B, nsv)), np.zeros(B), np.zeros(B)
    L = lib(big)
    L.grbda_oracle_spanning_state.argtypes = [c_void_p, c_size_t] + [c_void_p] * 6 + [c_size_t]
    rc = L.grbda_oracle_spanning_state(blob, len(blob), q.ctypes.data, qd.ctypes.data, qs.ctypes.data, vs.ctypes.data,
                                       gm.ctypes.data, kc.ctypes.data, B)
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return qs, vs, gm, kc


def project_positions(blob, q, max_iter=50, big=False):
    q = _f64(q).copy()
    ok = np.zeros(q.shape[0], dtype=np.int32)
    rc = lib(big).grbda_oracle_project_positions(blob, len(blob), q.ctypes.data, q.shape[0], max_iter, ok.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return q, ok.astype(bool)


# ---- the same restatement in x87 extended precision (_build/libgrbda_oracle_ld.so): tests only ----------
_lib_ld = None
LD = np.longdouble


def _ld():
    global _lib_ld
    if _lib_ld is None:
        path = os.path.join(_HERE, "_build", "libgrbda_oracle_ld.so")
        if not os.path.exists(path):
            build()
        assert np.finfo(LD).nmant >= 63, "numpy's longdouble is not the x87 extended type on this machine"
        L = ctypes.CDLL(path)
        L.grbda_oracle_forward_dynamics.argtypes = [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]
        L.grbda_oracle_project_positions.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_void_p]
        _lib_ld = L
    return _lib_ld


def forward_dynamics_ld(blob, q, qd, tau):
    """forward dynamics in long double (eps 1.1e-19): arrays of np.longdouble in and out"""
    q, qd, tau = (np.ascontiguousarray(a, dtype=LD) for a in (q, qd, tau))
    out = np.empty_like(tau)
    rc = _ld().grbda_oracle_forward_dynamics(blob, len(blob), q.ctypes.data, qd.ctypes.data, tau.ctypes.data, None, out.ctypes.data,
                                             q.shape[0])
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return out


def project_positions_ld(blob, q, max_iter=60):
    """Newton projection of the dependent positions of implicit clusters in long double, to |phi| < 1e-17"""
    q = np.ascontiguousarray(q, dtype=LD).copy()
    ok = np.zeros(q.shape[0], dtype=np.int32)
    rc = _ld().grbda_oracle_project_positions(blob, len(blob), q.ctypes.data, q.shape[0], max_iter, ok.ctypes.data)
    if rc:
        raise RuntimeError(f"oracle error {rc}")
    return q, ok.astype(bool)


# ---- the reference's own closed-form codegen (oracle/_ref) -------------------------------------------
_REF_FUNCS = {
    ("rev", 2, "FD"): ("RevWithRotors2DofFwdDyn", 2), ("rev", 2, "ID"): ("RevWithRotors2DofInvDyn", 2),
    ("rev", 4, "FD"): ("RevWithRotors4DofFwdDyn", 4), ("rev", 4, "ID"): ("RevWithRotors4DofInvDyn", 4),
    ("pair", 2, "FD"): ("RevPairWithRotors2DofFwdDyn", 2), ("pair", 2, "ID"): ("RevPairWithRotors2DofInvDyn", 2),
    ("pair", 4, "FD"): ("RevPairWithRotors4DofFwdDyn", 4), ("pair", 4, "ID"): ("RevPairWithRotors4DofInvDyn", 4),
}


def ref_available() -> bool:
    return os.path.exists(REF_LIB)


def ref_codegen(family: str, n: int, kind: str, y, yd, x):
    """Call the reference's CasADi-generated closed form (include/grbda/Codegen/*.h:
    int f(const double** arg, double** res, long long* iw, double* w, int mem))."""
    L = ctypes.CDLL(REF_LIB)
    name, nd = _REF_FUNCS[(family, n, kind)]
    f = getattr(L, name)
    work = getattr(L, name + "_work")
    sa, sr, siw, sw = c_longlong(), c_longlong(), c_longlong(), c_longlong()
    work(ctypes.byref(sa), ctypes.byref(sr), ctypes.byref(siw), ctypes.byref(sw))
    y, yd, x = _f64(y), _f64(yd), _f64(x)
    out = np.empty_like(x)
    arg = (POINTER(c_double) * max(3, sa.value))()
    res = (POINTER(c_double) * max(1, sr.value))()
    iw = (c_longlong * max(1, siw.value))()
    w = (c_double * max(1, sw.value))()
    for s in range(y.shape[0]):
        arg[0] = y[s].ctypes.data_as(POINTER(c_double))
        arg[1] = yd[s].ctypes.data_as(POINTER(c_double))
        arg[2] = x[s].ctypes.data_as(POINTER(c_double))
        res[0] = out[s].ctypes.data_as(POINTER(c_double))
        f(arg, res, iw, w, 0)
    return out
