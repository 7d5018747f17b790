"""generalized_rbda_amd -- Python plumbing over the C ABI of libgrbda_hip.so.

The product is the HIP library (``csrc/``) behind ``include/grbda_hip.h``; this package only loads it,
wraps plans, and passes torch device pointers / streams through.  There is no CPU fallback:
every dynamics call raises ``GrbdaError`` when the library or a HIP device is missing.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p
from typing import Optional

from . import modeldesc  # noqa: F401  (model-description builder)

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRBDA_HIP_LIB: another build of the same library (kernel build variants, `make variant`); development A/B runs only
LIB_PATH = os.environ.get("GRBDA_HIP_LIB") or os.path.join(_HERE, "libgrbda_hip.so")

# every entry point include/grbda_hip.h declares
C_ABI_SYMBOLS = [
    "grbda_strerror", "grbda_last_error", "grbda_plan_from_blob", "grbda_plan_from_urdf", "grbda_urdf_to_blob",
    "grbda_plan_free", "grbda_plan_release_work", "grbda_plan_dims", "grbda_plan_set_gravity", "grbda_plan_get_gravity", "grbda_plan_blob",
    "grbda_plan_info", "grbda_aba_f64", "grbda_aba_f32", "grbda_rnea_f64", "grbda_rnea_f32",
    "grbda_aba_host_f64", "grbda_rnea_host_f64", "grbda_time_kernel", "grbda_device_count",
    "grbda_bias_f64", "grbda_bias_f32", "grbda_mass_matrix_f64", "grbda_mass_matrix_f32",
    "grbda_fd_dtau_f64", "grbda_fd_dtau_f32", "grbda_fd_dqd_f64", "grbda_fd_dqd_f32",
    "grbda_aba_sharded_f32", "grbda_aba_sharded_f64", "grbda_rnea_sharded_f32", "grbda_rnea_sharded_f64",
    "grbda_aba_sharded_dev_f32", "grbda_aba_sharded_dev_f64", "grbda_rnea_sharded_dev_f32", "grbda_rnea_sharded_dev_f64",
    "grbda_debug_dump_plan", "grbda_body_poses_host_f64", "grbda_apply_test_force_host_f64",
    "grbda_inv_osim_host_f64", "grbda_fd_dq_f64", "grbda_fd_dq_f32", "grbda_body_poses_f64", "grbda_body_poses_f32",
    "grbda_apply_test_force_f64", "grbda_apply_test_force_f32", "grbda_inv_osim_f64", "grbda_inv_osim_f32",
    "grbda_project_positions_f64", "grbda_project_positions_f32", "grbda_plan_span_dims",
    "grbda_spanning_f64", "grbda_spanning_f32", "grbda_fd_derivatives_f64", "grbda_fd_derivatives_f32",
    "grbda_mass_matrix_host_f64", "grbda_fd_derivatives_host_f64",
    "grbda_body_twists_f64", "grbda_body_twists_f32", "grbda_body_twists_host_f64",
    "grbda_state_input_dims", "grbda_state_to_independent_f64", "grbda_state_to_independent_f32",
    "grbda_state_to_independent_host_f64", "grbda_spd_bad_pivots", "grbda_kernel_name", "grbda_project_positions_host_f64",
]


class GrbdaError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        super().__init__(f"grbda error {code}: {detail}")


class PlanInfo(ctypes.Structure):
    _fields_ = [("n_slots", c_int), ("n_lds_slots_f32", c_int), ("n_lds_slots_f64", c_int),
                ("lds_bytes_f32", c_size_t), ("lds_bytes_f64", c_size_t),
                ("scratch_bytes_per_wave_f32", c_size_t), ("scratch_bytes_per_wave_f64", c_size_t),
                ("flops_aba", c_double), ("flops_rnea", c_double),
                ("bytes_aba_f32", c_double), ("bytes_aba_f64", c_double),
                ("n_axisym_bodies", c_int), ("n_carry_clusters", c_int),
                ("split_aba_f32", c_int), ("split_rnea_f32", c_int), ("n_lds_slots_split_f32", c_int),
                ("chain_aba_f32", c_int), ("n_lds_slots_chain_f32", c_int), ("n_chain_segments", c_int),
                ("chain_aba_f64", c_int), ("chain_rnea_f32", c_int), ("chain_rnea_f64", c_int),
                ("analytic_derivatives", c_int), ("n_chain_differentials", c_int),
                ("latency_mode_f32", c_int), ("latency_mode_f64", c_int), ("n_chain_generic", c_int), ("spanning_tree_route", c_int)]


_lib = None


def lib() -> ctypes.CDLL:
    """Load libgrbda_hip.so (built in-tree by ``make`` / ``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GrbdaError(-3, f"{LIB_PATH} is missing: build it with `make` (there is no CPU fallback)")
    try:
        # torch bundles its own libamdhip64; load it FIRST so that this library binds to the same
        # HIP runtime instance (device pointers and streams are then shared with torch tensors)
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    L.grbda_strerror.restype = c_char_p
    L.grbda_strerror.argtypes = [c_int]
    L.grbda_last_error.restype = c_char_p
    L.grbda_plan_from_blob.argtypes = [c_void_p, c_size_t, POINTER(c_void_p)]
    L.grbda_plan_from_urdf.argtypes = [c_char_p, c_int, POINTER(c_void_p)]
    L.grbda_urdf_to_blob.argtypes = [POINTER(c_char_p), c_int, c_int, c_void_p, c_size_t, POINTER(c_size_t)]
    L.grbda_plan_free.argtypes = [c_void_p]
    L.grbda_plan_free.restype = None
    L.grbda_plan_release_work.argtypes = [c_void_p, POINTER(ctypes.c_ulonglong)]
    for name in ("grbda_aba_sharded_dev_f32", "grbda_aba_sharded_dev_f64", "grbda_rnea_sharded_dev_f32", "grbda_rnea_sharded_dev_f64"):
        getattr(L, name).argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.grbda_plan_dims.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    L.grbda_plan_set_gravity.argtypes = [c_void_p, POINTER(c_double)]
    L.grbda_plan_get_gravity.argtypes = [c_void_p, POINTER(c_double)]
    L.grbda_plan_blob.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]
    L.grbda_plan_info.argtypes = [c_void_p, POINTER(PlanInfo)]
    for name in ("grbda_aba_f64", "grbda_aba_f32", "grbda_rnea_f64", "grbda_rnea_f32"):
        getattr(L, name).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                     c_void_p]
    for name in ("grbda_aba_host_f64", "grbda_rnea_host_f64"):
        getattr(L, name).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int]
    L.grbda_time_kernel.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                    c_void_p, c_int, POINTER(c_float)]
    L.grbda_device_count.restype = c_int
    for sfx in ("f64", "f32"):
        getattr(L, "grbda_bias_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                                    c_void_p]
        getattr(L, "grbda_mass_matrix_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]
        getattr(L, "grbda_fd_dtau_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]
        getattr(L, "grbda_fd_dqd_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                      c_int, c_void_p]
    L.grbda_plan_span_dims.argtypes = [c_void_p, POINTER(c_int)]
    for name in ("grbda_aba_sharded_f32", "grbda_aba_sharded_f64", "grbda_rnea_sharded_f32", "grbda_rnea_sharded_f64"):
        getattr(L, name).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int]
    for sfx in ("f64", "f32"):
        getattr(L, "grbda_body_poses_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]
        getattr(L, "grbda_body_twists_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]
        getattr(L, "grbda_inv_osim_" + sfx).argtypes = [c_void_p, c_void_p, c_int, POINTER(c_int), POINTER(c_double),
                                                        c_void_p, c_void_p, c_size_t, c_int, c_void_p]
        getattr(L, "grbda_apply_test_force_" + sfx).argtypes = [c_void_p, c_void_p, c_int, POINTER(c_double), c_void_p,
                                                                c_void_p, c_void_p, c_size_t, c_int, c_void_p]
    for sfx in ("f64", "f32"):
        getattr(L, "grbda_fd_dq_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p,
                                                     c_size_t, c_int, c_void_p]
    for sfx in ("f64", "f32"):
        getattr(L, "grbda_fd_derivatives_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                              c_void_p, c_size_t, c_int, c_void_p]
    for sfx in ("f64", "f32"):
        getattr(L, "grbda_project_positions_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_int,
                                                                 c_double, c_int, c_void_p]
        getattr(L, "grbda_spanning_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                        c_size_t, c_int, c_void_p]
    L.grbda_state_input_dims.argtypes = [c_void_p, c_void_p, c_void_p, POINTER(c_int), POINTER(c_int)]
    for sfx in ("f64", "f32"):
        getattr(L, "grbda_state_to_independent_" + sfx).argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                                    c_void_p, c_void_p, c_void_p, c_size_t, c_double, c_int,
                                                                    c_void_p]
    L.grbda_state_to_independent_host_f64.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                      c_size_t, c_double, c_int]
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise GrbdaError(rc, (lib().grbda_last_error() or b"").decode() or lib().grbda_strerror(rc).decode())


def urdf_to_blob(paths, ori_repr: str = "quaternion") -> bytes:
    """ClusterTreeModel::buildModelFromURDF(path | vector<path>) -> model description bytes."""
    if isinstance(paths, (str, os.PathLike)):
        paths = [paths]
    arr = (c_char_p * len(paths))(*[os.fspath(p).encode() for p in paths])
    ori = 0 if ori_repr.lower().startswith("q") else 1
    need = c_size_t(0)
    _check(lib().grbda_urdf_to_blob(arr, len(paths), ori, None, 0, byref(need)))
    buf = ctypes.create_string_buffer(need.value)
    _check(lib().grbda_urdf_to_blob(arr, len(paths), ori, buf, need.value, byref(need)))
    return buf.raw[: need.value]


class Plan:
    """An immutable compiled model (``grbda_plan``)."""

    def __init__(self, blob: bytes):
        self._h = c_void_p()
        self._blob = bytes(blob)
        _check(lib().grbda_plan_from_blob(self._blob, len(self._blob), byref(self._h)))
        nq, nv, nb, nc = c_int(), c_int(), c_int(), c_int()
        _check(lib().grbda_plan_dims(self._h, byref(nq), byref(nv), byref(nb), byref(nc)))
        self.nq, self.nv, self.n_bodies, self.n_clusters = nq.value, nv.value, nb.value, nc.value

    @classmethod
    def from_model(cls, model) -> "Plan":
        return cls(model.serialize())

    @classmethod
    def from_urdf(cls, path, ori_repr: str = "quaternion") -> "Plan":
        return cls(urdf_to_blob(path, ori_repr))

    def __del__(self):
        try:
            if self._h:
                lib().grbda_plan_free(self._h)
                self._h = c_void_p()
        except Exception:
            pass

    @property
    def blob(self) -> bytes:
        p, n = c_void_p(), c_size_t()
        _check(lib().grbda_plan_blob(self._h, byref(p), byref(n)))
        return ctypes.string_at(p, n.value)

    def set_gravity(self, g):
        arr = (c_double * 3)(*[float(x) for x in g])
        _check(lib().grbda_plan_set_gravity(self._h, arr))

    def get_gravity(self):
        arr = (c_double * 3)()
        _check(lib().grbda_plan_get_gravity(self._h, arr))
        return list(arr)

    def release_work(self) -> int:
        """Frees the work buffers the chunked pipelines keep per (device, stream) between calls (grbda_plan_release_work);
        returns the bytes handed back."""
        n = ctypes.c_ulonglong(0)
        _check(lib().grbda_plan_release_work(self._h, byref(n)))
        return int(n.value)

    def info(self) -> PlanInfo:
        info = PlanInfo()
        _check(lib().grbda_plan_info(self._h, byref(info)))
        return info

    def kernel_name(self, algo: str, dtype: str, B: int, device: int = 0) -> str:
        """The kernel forward ("aba") / inverse ("rnea") dynamics launch for B states in "f32" / "f64" (grbda_kernel_name)."""
        buf = ctypes.create_string_buffer(256)
        _check(lib().grbda_kernel_name(self._h, 0 if algo == "aba" else 1, 32 if dtype == "f32" else 64, ctypes.c_size_t(B), int(device), buf, 256))
        return buf.value.decode()

    # ---- batched dynamics on torch device tensors ------------------------------------------------
    def _launch(self, which: str, q, qd, x, out=None, stream=None, f_ext=None):
        import torch

        if q.dtype not in (torch.float32, torch.float64):
            raise TypeError("q must be float32 or float64")
        if not (q.is_cuda and qd.is_cuda and x.is_cuda):
            raise GrbdaError(-3, "inputs must be HIP device tensors (there is no CPU fallback)")
        B = q.shape[0]
        if q.shape != (B, self.nq) or qd.shape != (B, self.nv) or x.shape != (B, self.nv):
            raise ValueError(f"expected q[B,{self.nq}], qd[B,{self.nv}], x[B,{self.nv}]")
        q, qd, x = q.contiguous(), qd.contiguous(), x.contiguous()
        if qd.dtype != q.dtype or x.dtype != q.dtype:
            raise TypeError("dtype mismatch")
        if qd.device != q.device or x.device != q.device:
            raise ValueError("q, qd and x must live on one device")
        if out is None:
            out = torch.empty((B, self.nv), dtype=q.dtype, device=q.device)
        elif (out.shape != (B, self.nv) or out.dtype != q.dtype or out.device != q.device or not out.is_contiguous()):
            raise ValueError(f"out must be a contiguous [B,{self.nv}] tensor of the inputs' dtype on the inputs' device")
        fe = None
        if f_ext is not None:
            if f_ext.shape != (B, self.n_bodies, 6) or f_ext.dtype != q.dtype or not f_ext.is_cuda:
                raise ValueError(f"f_ext must be a device tensor [B,{self.n_bodies},6] of the same dtype")
            f_ext = f_ext.contiguous()
            fe = f_ext.data_ptr()
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        if stream is not None:
            # contiguous() temporaries and `out` are used on a caller-supplied stream: tell the caching allocator
            for tns in (q, qd, x, out) + (() if f_ext is None else (f_ext,)):
                tns.record_stream(s)
        fn = getattr(lib(), f"grbda_{which}_{'f32' if q.dtype == torch.float32 else 'f64'}")
        _check(fn(self._h, q.data_ptr(), qd.data_ptr(), x.data_ptr(), fe, out.data_ptr(), B,
                  q.device.index or 0, c_void_p(s.cuda_stream)))
        return out

    def forward_dynamics(self, q, qd, tau, out=None, stream=None, f_ext=None):
        """Batched ClusterTreeModel::forwardDynamics (cluster ABA): returns ydd[B, nv].
        f_ext: optional [B, n_bodies, 6] world-frame spatial forces (TreeModel::setExternalForces)."""
        return self._launch("aba", q, qd, tau, out, stream, f_ext)

    def inverse_dynamics(self, q, qd, ydd, out=None, stream=None, f_ext=None):
        """Batched ClusterTreeModel::inverseDynamics (cluster RNEA): returns tau[B, nv]."""
        return self._launch("rnea", q, qd, ydd, out, stream, f_ext)

    # ---- quantities derived from the two recursions (include/grbda_hip.h) ---------------------------
    @staticmethod
    def _floating(*tensors):
        """every tensor: float32 or float64, one dtype, one HIP device (no silent reinterpretation of other dtypes)"""
        import torch

        t0 = tensors[0]
        if t0.dtype not in (torch.float32, torch.float64) or not t0.is_cuda:
            raise GrbdaError(-3, "inputs must be float32/float64 HIP device tensors (there is no CPU fallback)")
        for t in tensors[1:]:
            if t.dtype != t0.dtype or t.device != t0.device:
                raise TypeError("all tensors of one call must share dtype and device")

    def _derived(self, name: str, q, others=(), matrix=True, stream=None, f_ext=None):
        import torch

        if not q.is_cuda or q.dtype not in (torch.float32, torch.float64):
            raise GrbdaError(-3, "inputs must be float32/float64 HIP device tensors (there is no CPU fallback)")
        B = q.shape[0]
        if q.shape != (B, self.nq) or any(o.shape != (B, self.nv) or o.dtype != q.dtype or not o.is_cuda for o in others):
            raise ValueError(f"expected q[B,{self.nq}] and [B,{self.nv}] tensors of one dtype on the device")
        q = q.contiguous()
        others = [o.contiguous() for o in others]
        out = torch.empty((B, self.nv, self.nv) if matrix else (B, self.nv), dtype=q.dtype, device=q.device)
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_{name}_{'f32' if q.dtype == torch.float32 else 'f64'}")
        args = [self._h, q.data_ptr()] + [o.data_ptr() for o in others]
        if name == "bias":
            if f_ext is not None and (f_ext.shape != (B, self.n_bodies, 6) or f_ext.dtype != q.dtype or not f_ext.is_cuda):
                raise ValueError(f"f_ext must be a device tensor [B,{self.n_bodies},6] of the same dtype")
            args.append(None if f_ext is None else f_ext.contiguous().data_ptr())
        _check(fn(*args, out.data_ptr(), B, q.device.index or 0, c_void_p(s.cuda_stream)))
        return out

    def bias_force(self, q, qd, stream=None, f_ext=None):
        """Batched getBiasForceVector: C(q, qd) = RNEA(q, qd, 0), [B, nv]."""
        return self._derived("bias", q, (qd,), matrix=False, stream=stream, f_ext=f_ext)

    def mass_matrix(self, q, stream=None):
        """Batched getMassMatrix: H(q), [B, nv, nv]."""
        return self._derived("mass_matrix", q, stream=stream)

    def fd_dtau(self, q, stream=None):
        """d ydd / d tau = H(q)^-1 through the ABA, [B, nv, nv]."""
        return self._derived("fd_dtau", q, stream=stream)

    def fd_dqd(self, q, qd, tau, stream=None):
        """d ydd / d qd of the forward dynamics, [B, nv, nv] (exact: the ABA is quadratic in qd)."""
        return self._derived("fd_dqd", q, (qd, tau), stream=stream)

    # ---- steps either side of the path ------------------------------------------------------------
    @property
    def n_span_vel(self) -> int:
        n = c_int(0)
        _check(lib().grbda_plan_span_dims(self._h, byref(n)))
        return n.value

    def project_positions(self, q, max_iter: int = 50, tol: float = 1e-8, stream=None):
        """Newton projection of the dependent coordinates of implicit clusters onto phi(q) = 0, IN PLACE
        (GenericJoint.cpp:289-385).  Returns a bool tensor [B]: the state converged to |phi| < tol."""
        import torch

        if not q.is_cuda or q.dtype not in (torch.float32, torch.float64) or not q.is_contiguous():
            raise GrbdaError(-3, "q must be a contiguous float32/float64 HIP device tensor (there is no CPU fallback)")
        B = q.shape[0]
        if q.shape != (B, self.nq):
            raise ValueError(f"expected q[B,{self.nq}]")
        ok = torch.empty((B,), dtype=torch.int32, device=q.device)
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_project_positions_{'f32' if q.dtype == torch.float32 else 'f64'}")
        _check(fn(self._h, q.data_ptr(), ok.data_ptr(), B, max_iter, tol, q.device.index or 0, c_void_p(s.cuda_stream)))
        return ok.bool()

    @staticmethod
    def _flag_bytes(flags, n):
        if flags is None:
            return None
        if len(flags) != n:
            raise ValueError(f"expected one flag per cluster ({n})")
        return (ctypes.c_uint8 * n)(*[1 if f else 0 for f in flags])

    def state_input_dims(self, pos_is_spanning=None, vel_is_spanning=None):
        """Row widths (in_nq, in_nv) of a batch whose clusters give spanning / independent coordinates as flagged."""
        a, b = c_int(0), c_int(0)
        _check(lib().grbda_state_input_dims(self._h, self._flag_bytes(pos_is_spanning, self.n_clusters),
                                            self._flag_bytes(vel_is_spanning, self.n_clusters), byref(a), byref(b)))
        return a.value, b.value

    def state_to_independent(self, q_in, qd_in=None, pos_is_spanning=None, vel_is_spanning=None, tol: float = 1e-8,
                             want_gmax: bool = False, stream=None):
        """ClusterTreeModel::setState(ModelState) for a batch (ClusterJoint.cpp:22-71): per-cluster spanning or independent
        coordinates -> the engine's (q[B,nq], qd[B,nv]) plus status[B] (0 = valid; code + 256 * cluster otherwise) and,
        optionally, cond[B, 2] = (max |K_d^-1 K_i|, max |K_d|_F |K_d^-1|_F) over the implicit clusters."""
        import torch

        self._floating(*([q_in] if qd_in is None else [q_in, qd_in]))
        B = q_in.shape[0]
        wq, wv = self.state_input_dims(pos_is_spanning, vel_is_spanning)
        if q_in.shape != (B, wq) or (qd_in is not None and qd_in.shape != (B, wv)):
            raise ValueError(f"expected q_in[B,{wq}], qd_in[B,{wv}]")
        q_in = q_in.contiguous()
        qd_in = None if qd_in is None else qd_in.contiguous()
        q = torch.empty((B, self.nq), dtype=q_in.dtype, device=q_in.device)
        qd = None if qd_in is None else torch.empty((B, self.nv), dtype=q_in.dtype, device=q_in.device)
        status = torch.empty((B,), dtype=torch.int32, device=q_in.device)
        gmax = torch.empty((B, 2), dtype=q_in.dtype, device=q_in.device) if want_gmax else None
        s = torch.cuda.current_stream(q_in.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_state_to_independent_{'f32' if q_in.dtype == torch.float32 else 'f64'}")
        _check(fn(self._h, self._flag_bytes(pos_is_spanning, self.n_clusters), self._flag_bytes(vel_is_spanning, self.n_clusters),
                  q_in.data_ptr(), None if qd_in is None else qd_in.data_ptr(), q.data_ptr(),
                  None if qd is None else qd.data_ptr(), status.data_ptr(), None if gmax is None else gmax.data_ptr(), B, tol,
                  q_in.device.index or 0, c_void_p(s.cuda_stream)))
        return (q, qd, status, gmax) if want_gmax else (q, qd, status)

    def constraint_gain(self, q, tol: float = 1e-8, stream=None):
        """(gmax[B], kcond[B], status[B]) of engine-coordinate positions q[B,nq]: max |K_d^-1 K_i| and max |K_d|_F |K_d^-1|_F
        over the implicit clusters, and the validity status of the positions (|phi| < tol)."""
        _, _, status, cond = self.state_to_independent(q, None, None, None, tol=tol, want_gmax=True, stream=stream)
        return cond[:, 0].contiguous(), cond[:, 1].contiguous(), status

    def spanning(self, q, qd, ydd, stream=None):
        """qd_span = G yd and qdd_span = G ydd + g for every body joint: two tensors [B, n_span_vel]."""
        import torch

        self._floating(q, qd, ydd)

        B = q.shape[0]
        if not q.is_cuda or q.shape != (B, self.nq) or qd.shape != (B, self.nv) or ydd.shape != (B, self.nv):
            raise ValueError(f"expected device tensors q[B,{self.nq}], qd[B,{self.nv}], ydd[B,{self.nv}]")
        q, qd, ydd = q.contiguous(), qd.contiguous(), ydd.contiguous()
        n = self.n_span_vel
        v = torch.empty((B, n), dtype=q.dtype, device=q.device)
        a = torch.empty((B, n), dtype=q.dtype, device=q.device)
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_spanning_{'f32' if q.dtype == torch.float32 else 'f64'}")
        _check(fn(self._h, q.data_ptr(), qd.data_ptr(), ydd.data_ptr(), v.data_ptr(), a.data_ptr(), B,
                  q.device.index or 0, c_void_p(s.cuda_stream)))
        return v, a

    # ---- contact side ---------------------------------------------------------------------------------
    def body_poses(self, q, stream=None):
        """Absolute transforms world -> body (TreeNode::Xa_): [B, n_bodies, 12] = E (9, row-major) then r (3)."""
        import torch

        self._floating(q)

        B = q.shape[0]
        if not q.is_cuda or q.shape != (B, self.nq):
            raise ValueError(f"expected a device tensor q[B,{self.nq}]")
        q = q.contiguous()
        out = torch.empty((B, self.n_bodies, 12), dtype=q.dtype, device=q.device)
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_body_poses_{'f32' if q.dtype == torch.float32 else 'f64'}")
        _check(fn(self._h, q.data_ptr(), out.data_ptr(), B, q.device.index or 0, c_void_p(s.cuda_stream)))
        return out

    def body_twists(self, q, qd, ydd, stream=None):
        """Spatial velocity and acceleration of every body in its own coordinates (TreeNode::v_ / a_ after
        forwardAccelerationKinematics): [B, n_bodies, 12] = v (6) then a (6), each [angular; linear]; the acceleration
        carries the base's -gravity as in the reference."""
        import torch

        self._floating(q, qd, ydd)
        B = q.shape[0]
        if not q.is_cuda or q.shape != (B, self.nq) or qd.shape != (B, self.nv) or ydd.shape != (B, self.nv):
            raise ValueError(f"expected device tensors q[B,{self.nq}], qd[B,{self.nv}], ydd[B,{self.nv}]")
        q, qd, ydd = q.contiguous(), qd.contiguous(), ydd.contiguous()
        out = torch.empty((B, self.n_bodies, 12), dtype=q.dtype, device=q.device)
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_body_twists_{'f32' if q.dtype == torch.float32 else 'f64'}")
        _check(fn(self._h, q.data_ptr(), qd.data_ptr(), ydd.data_ptr(), out.data_ptr(), B, q.device.index or 0,
                  c_void_p(s.cuda_stream)))
        return out

    def apply_test_force(self, q, body: int, offset, force, stream=None):
        """Batched ClusterTreeModel::applyTestForce: world-frame force[B,3] at the body-fixed point `offset` of
        `body`; returns (lambda_inv[B], dstate[B,nv]) = (f^T J H^-1 J^T f, H^-1 J^T f)."""
        import torch

        self._floating(q, force)

        B = q.shape[0]
        if not q.is_cuda or q.shape != (B, self.nq) or force.shape != (B, 3) or force.dtype != q.dtype:
            raise ValueError(f"expected device tensors q[B,{self.nq}], force[B,3] of one dtype")
        q, force = q.contiguous(), force.contiguous()
        lam = torch.empty((B,), dtype=q.dtype, device=q.device)
        ds = torch.empty((B, self.nv), dtype=q.dtype, device=q.device)
        off = (c_double * 3)(*[float(x) for x in offset])
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_apply_test_force_{'f32' if q.dtype == torch.float32 else 'f64'}")
        _check(fn(self._h, q.data_ptr(), body, off, force.data_ptr(), lam.data_ptr(), ds.data_ptr(), B,
                  q.device.index or 0, c_void_p(s.cuda_stream)))
        return lam, ds

    def inv_osim(self, q, bodies, offsets, with_jacobian: bool = False, stream=None):
        """Batched inverseOperationalSpaceInertiaMatrix for contact frames (body index, body-fixed offset):
        Linv[B, 6n, 6n] (and the frame Jacobians J[B, 6n, nv] when asked)."""
        import torch

        self._floating(q)

        B, n = q.shape[0], len(bodies)
        if not q.is_cuda or q.shape != (B, self.nq):
            raise ValueError(f"expected a device tensor q[B,{self.nq}]")
        q = q.contiguous()
        Linv = torch.empty((B, 6 * n, 6 * n), dtype=q.dtype, device=q.device)
        J = torch.empty((B, 6 * n, self.nv), dtype=q.dtype, device=q.device) if with_jacobian else None
        bod = (c_int * n)(*[int(b) for b in bodies])
        off = (c_double * (3 * n))(*[float(x) for o in offsets for x in o])
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_inv_osim_{'f32' if q.dtype == torch.float32 else 'f64'}")
        _check(fn(self._h, q.data_ptr(), n, bod, off, Linv.data_ptr(), None if J is None else J.data_ptr(), B,
                  q.device.index or 0, c_void_p(s.cuda_stream)))
        return (Linv, J) if with_jacobian else Linv

    def fd_derivatives(self, q, qd, tau, want=("dq", "dqd", "dtau"), stream=None):
        """d ydd / d q, d ydd / d qd, d ydd / d tau of the forward dynamics in one pass (grbda_fd_derivatives_*): a dict
        of [B, nv, nv] tensors for the names in `want`.  Explicit models get the analytic recursion + one SPD solve per
        state (deriv_kernels.hip); the others fall back to fd_dtau / fd_dqd / fd_dq."""
        import torch

        self._floating(q, qd, tau)
        B = q.shape[0]
        if q.shape != (B, self.nq) or qd.shape != (B, self.nv) or tau.shape != (B, self.nv):
            raise ValueError(f"expected device tensors q[B,{self.nq}], qd[B,{self.nv}], tau[B,{self.nv}]")
        q, qd, tau = q.contiguous(), qd.contiguous(), tau.contiguous()
        out = {k: torch.empty((B, self.nv, self.nv), dtype=q.dtype, device=q.device) for k in ("dq", "dqd", "dtau") if k in want}
        if B == 0:
            return out
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_fd_derivatives_{'f32' if q.dtype == torch.float32 else 'f64'}")
        ptr = lambda k: out[k].data_ptr() if k in out else None
        _check(fn(self._h, q.data_ptr(), qd.data_ptr(), tau.data_ptr(), ptr("dq"), ptr("dqd"), ptr("dtau"), B,
                  q.device.index or 0, c_void_p(s.cuda_stream)))
        return out

    def fd_dq(self, q, qd, tau, step: float = 1e-6, stream=None):
        """d ydd / d q along the reference's tangent step (testHelpers.hpp:50-112), [B, nv, nv].  Explicit models:
        analytic (`step` unused).  Models with implicit loops: central differences, always taken
        in fp64 (fp32 tensors are converted on the device), on the constraint manifold."""
        import torch

        self._floating(q, qd, tau)

        B = q.shape[0]
        if not q.is_cuda or q.shape != (B, self.nq) or qd.shape != (B, self.nv) or tau.shape != (B, self.nv):
            raise ValueError(f"expected device tensors q[B,{self.nq}], qd[B,{self.nv}], tau[B,{self.nv}]")
        q, qd, tau = q.contiguous(), qd.contiguous(), tau.contiguous()
        out = torch.empty((B, self.nv, self.nv), dtype=q.dtype, device=q.device)
        s = torch.cuda.current_stream(q.device) if stream is None else stream
        fn = getattr(lib(), f"grbda_fd_dq_{'f32' if q.dtype == torch.float32 else 'f64'}")
        _check(fn(self._h, q.data_ptr(), qd.data_ptr(), tau.data_ptr(), step, out.data_ptr(), B,
                  q.device.index or 0, c_void_p(s.cuda_stream)))
        return out

    def time_kernel(self, which: str, q, qd, x, out, iters: int = 20, stream=None) -> float:
        """Average kernel duration in ms, hipEvents on the launch stream (grbda_time_kernel)."""
        import torch

        self._floating(q, qd, x, out)

        s = torch.cuda.current_stream(q.device) if stream is None else stream
        ms = c_float(0)
        _check(lib().grbda_time_kernel(self._h, 0 if which == "aba" else 1, 32 if q.dtype == torch.float32 else 64,
                                       q.data_ptr(), qd.data_ptr(), x.data_ptr(), out.data_ptr(), q.shape[0],
                                       q.device.index or 0, c_void_p(s.cuda_stream), iters, byref(ms)))
        return ms.value

    def sharded_device(self, which: str, qs, qds, xs, outs=None, gathered=None, streams=None):
        """grbda_{aba,rnea}_sharded_dev_*: shard g = the device tensors qs[g], qds[g], xs[g] on THEIR device (one process, several
        devices), launched on streams[g] (default: each device's current stream).  outs: per-shard output tensors (made when
        None and no `gathered`); gathered: a [sum B, nv] tensor on the first shard's device that receives every slab over the
        peer links.  Enqueues only.  Returns (outs, gathered)."""
        import torch

        n = len(qs)
        if n < 1 or len(qds) != n or len(xs) != n:
            raise ValueError("one q, qd, x tensor per shard")
        dt = qs[0].dtype
        for g in range(n):
            self._floating(qs[g], qds[g], xs[g])
            if qs[g].dtype != dt or qs[g].shape[1:] != (self.nq,) or qds[g].shape != (qs[g].shape[0], self.nv) or xs[g].shape != qds[g].shape:
                raise ValueError(f"shard {g}: expected q[B,{self.nq}], qd[B,{self.nv}], x[B,{self.nv}] of one dtype")
        qs, qds, xs = [t.contiguous() for t in qs], [t.contiguous() for t in qds], [t.contiguous() for t in xs]
        Bs = [int(t.shape[0]) for t in qs]
        if outs is None and gathered is None:
            outs = [torch.empty((Bs[g], self.nv), dtype=dt, device=qs[g].device) for g in range(n)]
        if gathered is not None and outs is None:
            # shards on the gather device write straight into their place; the others need a slab of their own
            outs = [None if qs[g].device == gathered.device else torch.empty((Bs[g], self.nv), dtype=dt, device=qs[g].device) for g in range(n)]
        if gathered is not None and (gathered.shape != (sum(Bs), self.nv) or gathered.dtype != dt or not gathered.is_contiguous()
                                     or gathered.device != qs[0].device):
            raise ValueError("gathered must be a contiguous [sum B, nv] tensor of the shards' dtype on the first shard's device")
        if streams is None:
            streams = [torch.cuda.current_stream(t.device) for t in qs]
        P = ctypes.c_void_p
        arr = lambda ts: (P * n)(*[None if t is None else t.data_ptr() for t in ts])
        devs = (c_int * n)(*[t.device.index or 0 for t in qs])
        Bc = (ctypes.c_size_t * n)(*Bs)
        st = (P * n)(*[s.cuda_stream for s in streams])
        fn = getattr(lib(), f"grbda_{which}_sharded_dev_{'f32' if dt == torch.float32 else 'f64'}")
        _check(fn(self._h, n, devs, arr(qs), arr(qds), arr(xs), arr(outs), Bc, st, None if gathered is None else gathered.data_ptr()))
        return outs, gathered

    def sharded_host(self, which: str, q, qd, x, n_gpus: int):
        """Host numpy arrays (float32 or float64), batch split over devices 0 .. n_gpus-1 in this process."""
        import numpy as np

        dt = np.float32 if q.dtype == np.float32 else np.float64
        q, qd, x = (np.ascontiguousarray(a, dtype=dt) for a in (q, qd, x))
        out = np.empty_like(x)
        fn = getattr(lib(), f"grbda_{which}_sharded_{'f32' if dt == np.float32 else 'f64'}")
        _check(fn(self._h, q.ctypes.data, qd.ctypes.data, x.ctypes.data, out.ctypes.data, q.shape[0], n_gpus))
        return out

    # ---- host convenience (numpy, fp64) -----------------------------------------------------------
    def forward_dynamics_host(self, q, qd, tau, device: int = 0, f_ext=None):
        import numpy as np

        q, qd, tau = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, qd, tau))
        fe = None if f_ext is None else np.ascontiguousarray(f_ext, dtype=np.float64)
        out = np.empty_like(tau)
        _check(lib().grbda_aba_host_f64(self._h, q.ctypes.data, qd.ctypes.data, tau.ctypes.data,
                                        None if fe is None else fe.ctypes.data, out.ctypes.data, q.shape[0], device))
        return out

    def inverse_dynamics_host(self, q, qd, ydd, device: int = 0, f_ext=None):
        import numpy as np

        q, qd, ydd = (np.ascontiguousarray(a, dtype=np.float64) for a in (q, qd, ydd))
        fe = None if f_ext is None else np.ascontiguousarray(f_ext, dtype=np.float64)
        out = np.empty_like(ydd)
        _check(lib().grbda_rnea_host_f64(self._h, q.ctypes.data, qd.ctypes.data, ydd.ctypes.data,
                                         None if fe is None else fe.ctypes.data, out.ctypes.data, q.shape[0], device))
        return out


def device_count() -> int:
    return lib().grbda_device_count()


def spd_bad_pivots(device: int = 0, reset: bool = True) -> int:
    """States whose joint-space inertia was not positive definite in the SPD solves of the derivative entry points since the
    last reset (their results are NaN / Inf); synchronises the device (grbda_spd_bad_pivots)."""
    n = ctypes.c_ulonglong(0)
    _check(lib().grbda_spd_bad_pivots(int(device), ctypes.byref(n), 1 if reset else 0))
    return int(n.value)
