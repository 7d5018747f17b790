/*
 * grbda_hip.h -- C ABI of the MI355X batched cluster-ABA / cluster-RNEA engine.
 *
 * The reference (ROAM-Lab-ND/generalized_rbda) has no FFI: its boundary is the C++ class
 * grbda::ClusterTreeModel (include/grbda/Dynamics/ClusterTreeModel.h:24-165,
 * include/grbda/Dynamics/TreeModel.h:15-143).  Each entry point below cites the member it
 * replaces; the C++17 facade in generalized_rbda_amd/include/grbda keeps the reference's class
 * names and calls these functions, and INTEGRATION.md shows the few lines a maintainer of the
 * reference would add to route ClusterTreeModel::forwardDynamics through them.
 *
 * Conventions
 *   - all batched arrays are row-major: q[B][nq], qd[B][nv], tau[B][nv], ydd[B][nv];
 *     positions are *independent* coordinates for explicit clusters and *spanning* positions
 *     for implicit-loop clusters (GenericJoint.cpp:246-249); velocities / accelerations /
 *     torques are always independent (TreeModel.h:83-84);
 *   - `_f32/_f64` device entry points take DEVICE pointers on `device`, enqueue on `stream`
 *     (a hipStream_t, may be NULL) and return without synchronising;
 *   - the caller owns every buffer; a plan is immutable after creation and may be shared by
 *     threads and streams (the reference model is a stateful cache and is not thread safe,
 *     TreeModel.h:138-142);
 *   - functions return 0 (GRBDA_OK) or a negative error code and never throw;
 *   - there is NO CPU fallback: every compute entry point fails with GRBDA_ENODEVICE when no
 *     HIP device is usable.
 */
#ifndef GRBDA_HIP_H
#define GRBDA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    GRBDA_OK = 0,
    GRBDA_EINVAL = -1,       /* bad argument / malformed blob                                  */
    GRBDA_EUNSUPPORTED = -2, /* model uses a topology / constraint the kernels do not cover yet */
    GRBDA_ENODEVICE = -3,    /* no usable HIP device                                           */
    GRBDA_EHIP = -4,         /* a HIP runtime call failed (see grbda_last_error)               */
    GRBDA_ENOMEM = -5,
    GRBDA_EPARSE = -6,       /* URDF+ file could not be parsed                                 */
    GRBDA_ESTATE = -7        /* invalid spanning state (ClusterJoint.cpp:43-46,62-65)          */
};

typedef struct grbda_plan grbda_plan;

/* human-readable text for an error code; thread-local detail of the last failure */
const char *grbda_strerror(int code);
const char *grbda_last_error(void);

/* ---- model -> plan ------------------------------------------------------------------------ */

/* Replaces the state held by a constructed ClusterTreeModel (ClusterTreeModel.cpp:10-67):
 * `blob` is a model description (include/grbda_model_desc.h) written by the C++ facade
 * (grbda::ClusterTreeModel::serialize) or by grbda_urdf_to_blob.  Host-only: needs no GPU. */
int grbda_plan_from_blob(const void *blob, size_t bytes, grbda_plan **out);

/* Replaces ClusterTreeModel::buildModelFromURDF(path) (ClusterTreeModel.h:41-46,
 * src/Dynamics/ClusterTreeParsing.cpp:5-440).  ori_repr: 0 quaternion, 1 roll-pitch-yaw. */
int grbda_plan_from_urdf(const char *path, int ori_repr, grbda_plan **out);

/* URDF+ file(s) -> model description blob (buildModelFromURDF(vector<path>),
 * ClusterTreeModel.h:48-53).  Call with buf = NULL to query *needed. */
int grbda_urdf_to_blob(const char *const *paths, int n_paths, int ori_repr, void *buf, size_t cap,
                       size_t *needed);

void grbda_plan_free(grbda_plan *plan);

/* Hands the work buffers of the chunked pipelines (derivatives, mass matrix / derivatives on the constraint manifold, position
 * projection) back to the device allocator.  They are kept per (device, stream) between calls -- up to a few GiB after a
 * million-state derivative call (16 GiB on the manifold route) -- and otherwise live until grbda_plan_free; the next call that
 * needs one allocates it again.  Waits for the work enqueued on them; refused (GRBDA_EINVAL) while one of the plan's streams is
 * capturing.  *bytes_released (may be NULL): what was freed.  No counterpart in the reference: its temporaries are Eigen
 * objects of one state (ClusterTreeModel.h:150-170). */
int grbda_plan_release_work(grbda_plan *plan, unsigned long long *bytes_released);

/* getNumPositions / getNumDegreesOfFreedom / getNumBodies / clusters().size()
 * (TreeModel.h:25-26, ClusterTreeModel.h:98,113); any out pointer may be NULL */
int grbda_plan_dims(const grbda_plan *plan, int *nq, int *nv, int *n_bodies, int *n_clusters);

/* TreeModel::setGravity / getGravity (TreeModel.h:56-57): linear part of the gravity vector.
 * Not thread safe against concurrent launches on the same plan. */
int grbda_plan_set_gravity(grbda_plan *plan, const double g[3]);
int grbda_plan_get_gravity(const grbda_plan *plan, double g[3]);

/* the model description the plan was built from (for serialisation / the oracle in tests) */
int grbda_plan_blob(const grbda_plan *plan, const void **blob, size_t *bytes);

/* kernel resource / cost figures: per-state slots, LDS bytes per wave, global scratch bytes per
 * wave, flops per forward-dynamics and inverse-dynamics evaluation (counted by the plan
 * compiler from the operation list the kernel executes) */
typedef struct {
    int n_slots;
    int n_lds_slots_f32, n_lds_slots_f64;
    size_t lds_bytes_f32, lds_bytes_f64;
    size_t scratch_bytes_per_wave_f32, scratch_bytes_per_wave_f64;
    double flops_aba, flops_rnea;
    double bytes_aba_f32, bytes_aba_f64; /* algorithmic bytes per evaluation: (nq+2nv+nv)*s */
    int n_axisym_bodies;  /* leaf bodies evaluated at q = 0 (rotors), see plan.h */
    int n_carry_clusters; /* clusters whose projected inertia is handed over in registers */
    int split_aba_f32, split_rnea_f32; /* 1: the f32 kernel runs the split layout ([K | y0] blocks in the global slab,
                                          every other object in LDS, plan.h Layout::split_*) */
    int n_lds_slots_split_f32;
    int chain_aba_f32;         /* 1: the f32 forward dynamics run the chain-structured kernel (plan.h, ChainProgram) */
    int n_lds_slots_chain_f32, n_chain_segments;
    int chain_aba_f64;
    int chain_rnea_f32, chain_rnea_f64; /* inverse dynamics on the chains (rnea_chain_kernel) */
    int analytic_derivatives;           /* grbda_fd_d* / grbda_fd_derivatives_* take the analytic route (deriv_kernels.hip) */
    int n_chain_differentials;          /* implicit two-rotor differential clusters inside the f32 chain program */
    int latency_mode_f32, latency_mode_f64; /* 1: batches of at most one tile per SIMD run a tile on two wavefronts (plan.h, ChainProgram::n_waves) */
    int n_chain_generic;                /* generic clusters inside the f32 chain program (plan.h, ChainGen): URDF+ position loops, triple / Generic clusters */
    int spanning_tree_route;            /* 1: a cluster attaches to several bodies of its parent cluster, or has more than 8 bodies / 4 independent
                                           coordinates (up to 48 / 48; models up to 128 velocities) -- forward / inverse dynamics (with external
                                           forces), mass matrix and d ydd / d tau run through the spanning tree and the per-state G
                                           (H = G^T H_s G; capi.cpp projection_run), no sweep programs; INTEGRATION.md 3 lists what the route refuses */
} grbda_plan_info_t;
int grbda_plan_info(const grbda_plan *plan, grbda_plan_info_t *info);

/* ---- batched dynamics (device pointers) ----------------------------------------------------- */
/* The device entry points only enqueue work on the stream they are given and may be captured into a hipGraph as they are.
 * The library keeps one scratch slab per (device, stream) and grows it on demand (hipFree + hipMalloc): a captured launch
 * holds the slab's address, so (i) call once with the largest batch on that stream BEFORE capturing -- a call that would have
 * to grow the slab while its stream is capturing returns GRBDA_EINVAL -- and (ii) do not make a larger eager call on the same
 * (device, stream) while the graph is alive.
 * Batches of at most one 64-state tile per SIMD (4 x the CU count) take the latency-mode forward-dynamics kernel when the
 * plan has one (grbda_plan_info: latency_mode_f32 / _f64): same results to rounding, 20-35 % less time per call. */

/* ClusterTreeModel::setState + forwardDynamics(tau) over B independent states
 * (ClusterTreeModel.cpp:256-308, ClusterTreeDynamics.cpp:85-191): cluster ABA.
 * f_ext: NULL, or [B][n_bodies][6] world-frame spatial forces (TreeModel::setExternalForces,
 * TreeModel.cpp:214-239). */
int grbda_aba_f64(const grbda_plan *plan, const double *q, const double *qd, const double *tau,
                  const double *f_ext, double *ydd, size_t B, int device, void *stream);
int grbda_aba_f32(const grbda_plan *plan, const float *q, const float *qd, const float *tau,
                  const float *f_ext, float *ydd, size_t B, int device, void *stream);

/* ClusterTreeModel::setState + inverseDynamics(ydd) (ClusterTreeDynamics.cpp:79-83,
 * TreeModel.cpp:34-57,173-212): cluster RNEA. */
int grbda_rnea_f64(const grbda_plan *plan, const double *q, const double *qd, const double *ydd,
                   const double *f_ext, double *tau, size_t B, int device, void *stream);
int grbda_rnea_f32(const grbda_plan *plan, const float *q, const float *qd, const float *ydd,
                   const float *f_ext, float *tau, size_t B, int device, void *stream);

/* ---- quantities derived from the two recursions ---------------------------------- */
/* All take DEVICE pointers like grbda_aba_* / grbda_rnea_*.  They are evaluated with the same kernels over
 * an expanded batch (one extra row per unit vector), chunked so that the work space stays below 256 MiB:
 * RNEA is affine in ydd and ABA is affine in tau and quadratic in qd, so the differences below are exact
 * (no step size enters), not finite-difference approximations.
 *
 * grbda_bias_*        C(q, qd) = RNEA(q, qd, 0)                  TreeModel::updateBiasForceVector
 *                     out[B][nv]                                   (TreeModel.cpp:162-171), getBiasForceVector
 *                                                                  (ClusterTreeModel.cpp:106-110)
 * grbda_mass_matrix_* H(q) e_j = RNEA(q, 0, e_j) - RNEA(q, 0, 0) ClusterTreeModel::getMassMatrix
 *                     out[B][nv][nv], row-major, symmetric         (ClusterTreeModel.cpp:99-104; the reference runs
 *                                                                  the CRBA, TreeModel.cpp:115-160)
 * grbda_fd_dtau_*     d ydd / d tau = H^-1:                       the exact identity the reference's derivative
 *                     column j = ABA(q, 0, e_j) - ABA(q, 0, 0)     tests check (SURVEY 8c viii)
 *                     out[B][nv][nv]
 * grbda_fd_dqd_*      d ydd / d qd, column j =                     central difference with unit step, exact
 *                     (ABA(qd + e_j) - ABA(qd - e_j)) / 2          because ABA is quadratic in qd
 *                     out[B][nv][nv]                               (testRigidBodyDynamicsAlgosDerivatives.cpp:309-380)
 * grbda_fd_dq_*       d ydd / d q along the tangent step the         central difference with the caller's step: NOT
 *                     reference's derivative test uses                exact (truncation + rounding, the reference
 *                     (testHelpers.hpp:50-112: q_i += d; free base    uses step 1e-8 in fp64 and accepts 2e-5,
 *                     pos += R^T d, quat += quat x (0, d) / 2):       testRigidBodyDynamicsAlgosDerivatives.cpp:
 *                     column j = (ABA(q + h e_j) - ABA(q - h e_j))    309-335).  Implicit-loop models: an INDEPENDENT
 *                     / (2 h), out[B][nv][nv]                         position moves, the dependent ones are re-projected
 *                                                                     onto phi(q) = 0; a roll-pitch-yaw base: plain q + dq.
 *
 * Models made of explicit (constant G) clusters with nv <= 64 -- every URDF robot of
 * the reference without <loop> elements -- do not go through those batches: the mass matrix comes from the
 * composite-rigid-body kernel, and the three derivatives from the ANALYTIC recursion of deriv_kernels.hip
 * (d ID / d q and d ID / d qd of the spanning tree projected with G, then one batched SPD solve per state:
 * d ydd / d tau = H^-1, d ydd / d q = -H^-1 dID/dq, d ydd / d qd = -H^-1 dID/dqd at ydd = FD(q, qd, tau)); `step` is then
 * not used.  grbda_fd_derivatives_* returns any subset of the three matrices from ONE pass (NULL = not wanted) and falls
 * back to the three entry points above for the other models.  The analytic route keeps its intermediate matrices in a
 * device workspace owned by the plan, one per (device, stream), of at most 2 GiB (the batch goes through in chunks of that
 * size); the output arrays must not overlap the inputs or each other.  The f32 analytic route factors H per state by Cholesky
 * on the matrix-core kernel: a state whose H is not positive definite in single precision (massless chains, garbage input)
 * gets NaN / Inf in its three matrices while the call returns GRBDA_OK -- check the outputs, or use the _f64 entry points
 * (GRBDA_SOLVE_F64=1 keeps fp32 storage with an fp64 solve).
 */
int grbda_bias_f64(const grbda_plan *plan, const double *q, const double *qd, const double *f_ext, double *out,
                   size_t B, int device, void *stream);
int grbda_bias_f32(const grbda_plan *plan, const float *q, const float *qd, const float *f_ext, float *out,
                   size_t B, int device, void *stream);
int grbda_mass_matrix_f64(const grbda_plan *plan, const double *q, double *H, size_t B, int device, void *stream);
int grbda_mass_matrix_f32(const grbda_plan *plan, const float *q, float *H, size_t B, int device, void *stream);
int grbda_fd_dtau_f64(const grbda_plan *plan, const double *q, double *Hinv, size_t B, int device, void *stream);
int grbda_fd_dtau_f32(const grbda_plan *plan, const float *q, float *Hinv, size_t B, int device, void *stream);
int grbda_fd_dqd_f64(const grbda_plan *plan, const double *q, const double *qd, const double *tau, double *J,
                     size_t B, int device, void *stream);
int grbda_fd_dqd_f32(const grbda_plan *plan, const float *q, const float *qd, const float *tau, float *J,
                     size_t B, int device, void *stream);
int grbda_fd_dq_f64(const grbda_plan *plan, const double *q, const double *qd, const double *tau, double step,
                    double *J, size_t B, int device, void *stream);
int grbda_fd_dq_f32(const grbda_plan *plan, const float *q, const float *qd, const float *tau, double step, float *J,
                    size_t B, int device, void *stream);
int grbda_fd_derivatives_f64(const grbda_plan *plan, const double *q, const double *qd, const double *tau, double *dydd_dq,
                             double *dydd_dqd, double *dydd_dtau, size_t B, int device, void *stream);
int grbda_fd_derivatives_f32(const grbda_plan *plan, const float *q, const float *qd, const float *tau, float *dydd_dq,
                             float *dydd_dqd, float *dydd_dtau, size_t B, int device, void *stream);
/* States whose joint-space inertia matrix H was not positive definite to working precision in the SPD solves behind
 * grbda_fd_dtau_* / grbda_fd_dq_* / grbda_fd_dqd_* / grbda_fd_derivatives_* (a pivot of the factorisation <= 0 or not finite: massless
 * chains, a singular pose of an implicit cluster) get NaN / Inf results; the reference's ColPivHouseholderQR would return some
 * least-squares answer there (src/Dynamics/ClusterTreeNode.cpp:33-37).  The solve kernels COUNT such states per device: this call
 * synchronises the device, returns the count since the last reset and (reset != 0) clears it. */
int grbda_spd_bad_pivots(int device, unsigned long long *count, int reset);
/* The name of the kernel grbda_aba_* (kind 0) / grbda_rnea_* (kind 1) launches for a batch of B states in this precision (32 | 64) on
 * this device, without external forces -- as rocprofv3 prints it (template arguments included; the chain kernels' inverse dynamics
 * may append a wave-count variant).  bench.py reports it next to the roofline figures instead of re-deriving the selection. */
int grbda_kernel_name(const grbda_plan *plan, int kind, int precision, size_t B, int device, char *buf, size_t cap);

/* host-array variants (allocate, copy and synchronise per call: for the facade's single-state calls and small batches) */
int grbda_mass_matrix_host_f64(const grbda_plan *plan, const double *q, double *H, size_t B, int device);
int grbda_fd_derivatives_host_f64(const grbda_plan *plan, const double *q, const double *qd, const double *tau, double *dydd_dq,
                                  double *dydd_dqd, double *dydd_dtau, size_t B, int device);

/* ---- steps either side of the path ------------------------------------------ */
/* Newton projection of the DEPENDENT spanning coordinates of every implicit-loop cluster onto phi(q) = 0, in
 * place, the independent ones held fixed: q_d <- q_d - K_d^-1 phi until |phi| < 1e-12 or max_iter steps; a
 * state is reported valid (ok[b] = 1) when every cluster ends with |phi| < tol.  This is what the reference
 * does per state when it draws random states of a GenericImplicit joint (GenericJoint.cpp:289-385 with
 * Utilities.h:124-140; it accepts |phi| < 1e-8) and what a simulator calls against constraint drift.
 * q: [B][nq] device array, updated; ok: [B] device int32 or NULL.  Models without implicit clusters: no-op. */
int grbda_project_positions_f64(const grbda_plan *plan, double *q, int32_t *ok, size_t B, int max_iter, double tol,
                                int device, void *stream);
int grbda_project_positions_f32(const grbda_plan *plan, float *q, int32_t *ok, size_t B, int max_iter, double tol,
                                int device, void *stream);
/* The same with HOST arrays (copied to the device and back; synchronises): what the C++ facade's randomJointState() of an implicit
 * cluster of a URDF-built model runs (GenericJoint.cpp:289-348: independent coordinates U(-1, 1), dependent guess U(-0.1, 0.1), Newton,
 * at most 45 draws). */
int grbda_project_positions_host_f64(const grbda_plan *plan, double *q, int32_t *ok, size_t B, int max_iter, double tol, int device);

/* State input in the reference's conventions: ClusterJoints::Base::toSpanningTreeState (ClusterJoint.cpp:22-71) behind
 * ClusterTreeModel::setState(const ModelState&) (ClusterTreeModel.cpp:256-276).  Every JointState of a ModelState flags
 * its position and its velocity as independent or SPANNING coordinates (JointCoordinate::isSpanning(),
 * StateRepresentation.h:10-36); half of the reference's dynamics tests feed spanning states
 * (testRigidBodyDynamicsAlgos.cpp:45-72,195-196).  The compute entry points take the engine's own coordinates --
 * independent ones for explicit clusters, spanning POSITIONS for implicit clusters, independent velocities -- and these
 * calls convert a batch to them, with the reference's validity checks:
 *   pos_is_spanning[c], vel_is_spanning[c]  (n_clusters bytes each; NULL = the engine's convention / all independent)
 *   q_in  [B][in_nq], qd_in [B][in_nv]      the clusters' segments back to back, each n or k entries wide
 *                                           (free base: 7 or 6 positions, 6 velocities); widths: grbda_state_input_dims
 *   q [B][nq], qd [B][nv]                   engine coordinates (either may be NULL; qd_in may be NULL with qd)
 *   status[B]                               0, or code + 256 * cluster of the first failure in cluster order:
 *                                           1 "Spanning position is not valid"  (implicit cluster, |phi(q)|_2 >= tol;
 *                                             LoopConstraint.cpp:15-19.  Explicit clusters are taken as they are, like
 *                                             the reference: ClusterJoint.cpp:37-41),
 *                                           2 "Spanning velocity is not valid"  (|K qd_span|_2 >= tol, LoopConstraint.cpp:22-26)
 *   cond[B][2] (or NULL)                    maxima over the implicit clusters of [0] |K_d^-1 K_i| (the state-dependent
 *                                           entries of G: how strongly the loop amplifies rates) and [1] |K_d|_F |K_d^-1|_F
 *                                           (condition number of the inverted block; NaN where K_d is singular) -- what
 *                                           the synthetic-state generator gates on
 *   tol                                     the reference's nearZero tolerance is 1e-8 (Utilities.h:124-130)
 * Independent positions for an implicit cluster are refused with GRBDA_ESTATE before anything is launched
 * (ClusterJoint.cpp:32-35).  Device arrays; the _host_ variant takes host arrays, synchronises and returns GRBDA_ESTATE
 * with the reference's message in grbda_last_error() if any state is invalid (what the facade's setState throws). */
int grbda_state_input_dims(const grbda_plan *plan, const uint8_t *pos_is_spanning, const uint8_t *vel_is_spanning, int *in_nq,
                           int *in_nv);
int grbda_state_to_independent_f64(const grbda_plan *plan, const uint8_t *pos_is_spanning, const uint8_t *vel_is_spanning,
                                   const double *q_in, const double *qd_in, double *q, double *qd, int32_t *status, double *cond,
                                   size_t B, double tol, int device, void *stream);
int grbda_state_to_independent_f32(const grbda_plan *plan, const uint8_t *pos_is_spanning, const uint8_t *vel_is_spanning,
                                   const float *q_in, const float *qd_in, float *q, float *qd, int32_t *status, float *cond,
                                   size_t B, double tol, int device, void *stream);
int grbda_state_to_independent_host_f64(const grbda_plan *plan, const uint8_t *pos_is_spanning, const uint8_t *vel_is_spanning,
                                        const double *q_in, const double *qd_in, double *q, double *qd, size_t B, double tol,
                                        int device);

/* Spanning-tree velocities and accelerations of every body joint from the independent ones:
 * qd_span = G yd, qdd_span = G ydd + g (ClusterJoint.cpp:55-58, GenericJoint.cpp:57-90; what the reference's
 * benchmarks do with the result of forwardDynamics, pinocchioBenchmark.cpp:168-176).  Order: clusters in model
 * order, bodies by sub-index; the free base contributes its 6 components unchanged.
 * qd_span (may be NULL), qdd_span: [B][n_span_vel] device arrays (grbda_plan_span_dims). */
int grbda_plan_span_dims(const grbda_plan *plan, int *n_span_vel);
int grbda_spanning_f64(const grbda_plan *plan, const double *q, const double *qd, const double *ydd, double *qd_span,
                       double *qdd_span, size_t B, int device, void *stream);
int grbda_spanning_f32(const grbda_plan *plan, const float *q, const float *qd, const float *ydd, float *qd_span,
                       float *qdd_span, size_t B, int device, void *stream);

/* ---- contact side --------------------------------------------------------------------- */
/* Absolute transform world -> body of every body, TreeNode::Xa_ after TreeModel::forwardKinematics
 * (TreeModel.cpp:6-32): Xa[B][n_bodies][12] = rotation E (9, row-major: v_body = E v_world) then the position r
 * of the body origin in world coordinates (spatial::Transform, SpatialTransforms.cpp:13-40).  A point fixed in
 * the body at `offset` sits at r + E^T offset in the world (contact-point kinematics, TreeModel.cpp:59-76). */
int grbda_body_poses_f64(const grbda_plan *plan, const double *q, double *Xa, size_t B, int device, void *stream);
int grbda_body_poses_f32(const grbda_plan *plan, const float *q, float *Xa, size_t B, int device, void *stream);

/* Spatial velocity and acceleration of every body in its own coordinates, TreeNode::v_ / a_ after
 * TreeModel::forwardAccelerationKinematics(ydd) (TreeModel.cpp:6-57): V[B][n_bodies][12] = [v 6 | a 6], each [angular 3;
 * linear 3].  As in the reference the acceleration is the one the recursions carry -- the base starts from -gravity, so a
 * body at rest has a = X (0, 0, 0, 0, 0, 9.81) -- which is what ClusterTreeModel::getLinearAcceleration /
 * getAngularAcceleration (ClusterTreeModel.cpp:376-404) rotate into world axes and what
 * TreeModel::contactPointForwardAccelerationKinematics (TreeModel.cpp:78-99) adds gravity back to.  q, qd, ydd as for the
 * inverse dynamics (implicit clusters: spanning positions on the manifold). */
int grbda_body_twists_f64(const grbda_plan *plan, const double *q, const double *qd, const double *ydd, double *V, size_t B,
                          int device, void *stream);
int grbda_body_twists_f32(const grbda_plan *plan, const float *q, const float *qd, const float *ydd, float *V, size_t B,
                          int device, void *stream);

/* ClusterTreeModel::applyTestForce (ClusterTreeDynamics.cpp:194-233) for B states: a world-frame Cartesian force
 * force[B][3] acts at the point `offset` (host, body coordinates) of body `body`;
 * dstate[B][nv] = H^-1 J^T f and lambda_inv[B] = f^T J H^-1 J^T f.  Evaluated with the two kernels (ABA and RNEA
 * with and without the force as an external wrench), not with the reference's force propagators. */
int grbda_apply_test_force_f64(const grbda_plan *plan, const double *q, int body, const double offset[3],
                               const double *force, double *lambda_inv, double *dstate, size_t B, int device,
                               void *stream);
int grbda_apply_test_force_f32(const grbda_plan *plan, const float *q, int body, const double offset[3],
                               const float *force, float *lambda_inv, float *dstate, size_t B, int device,
                               void *stream);

/* ClusterTreeModel::inverseOperationalSpaceInertiaMatrix (ClusterTreeDynamics.cpp:295-435) and the contact
 * Jacobians for B states and up to 8 contact frames per call: frame c sits on body bodies[c] (host array) at the
 * body-fixed point offsets[c][3] (host) with the body's axes -- the frame of the reference's end-effector force
 * propagators (createSXform(1, local_offset), :316-320).
 *   Linv[B][6 n][6 n] = J H^-1 J^T, rows / columns ordered frame by frame, [moment 3, force 3] each;
 *   J[B][6 n][nv] (may be NULL): the 6-D Jacobians of the frames in their own coordinates.
 * Evaluated from unit wrenches through the ABA and RNEA kernels (6 n + 1 rows per state each), not with the
 * reference's extended-force-propagator recursion; the results are the same matrices. */
int grbda_inv_osim_f64(const grbda_plan *plan, const double *q, int n_contacts, const int *bodies,
                       const double *offsets, double *Linv, double *J, size_t B, int device, void *stream);
int grbda_inv_osim_f32(const grbda_plan *plan, const float *q, int n_contacts, const int *bodies,
                       const double *offsets, float *Linv, float *J, size_t B, int device, void *stream);

/* ---- convenience: host pointers (single-state facade calls, small batches) ------------------ */
/* allocate, copy in, run on `device`, copy out, synchronise.  Still the HIP path. */
int grbda_aba_host_f64(const grbda_plan *plan, const double *q, const double *qd, const double *tau,
                       const double *f_ext, double *ydd, size_t B, int device);
int grbda_rnea_host_f64(const grbda_plan *plan, const double *q, const double *qd,
                        const double *ydd, const double *f_ext, double *tau, size_t B, int device);

/* the same on HOST arrays (allocate, copy in, run, copy out, synchronise): single-state calls of the C++ facade */
int grbda_body_poses_host_f64(const grbda_plan *plan, const double *q, double *Xa, size_t B, int device);
int grbda_body_twists_host_f64(const grbda_plan *plan, const double *q, const double *qd, const double *ydd, double *V, size_t B,
                               int device);
int grbda_apply_test_force_host_f64(const grbda_plan *plan, const double *q, int body, const double offset[3],
                                    const double *force, double *lambda_inv, double *dstate, size_t B, int device);
int grbda_inv_osim_host_f64(const grbda_plan *plan, const double *q, int n_contacts, const int *bodies,
                            const double *offsets, double *Linv, double *J, size_t B, int device);

/* ---- one process, several devices --------------------------------------------------------------- */
/* HOST arrays, batch split into n_gpus contiguous shards on devices 0 .. n_gpus-1 (states are independent, the
 * plan is replicated, no collective): per shard allocate, copy in, run, copy out on its own stream, all shards
 * in flight together; returns when every shard is back.  A convenience for callers that are not one process per
 * GPU (bench.py is: torch.distributed over RCCL); the copies cross PCIe, so this is never the benchmarked rate. */
int grbda_aba_sharded_f32(const grbda_plan *plan, const float *q, const float *qd, const float *tau, float *ydd,
                          size_t B, int n_gpus);
int grbda_aba_sharded_f64(const grbda_plan *plan, const double *q, const double *qd, const double *tau, double *ydd,
                          size_t B, int n_gpus);
int grbda_rnea_sharded_f32(const grbda_plan *plan, const float *q, const float *qd, const float *ydd, float *tau,
                           size_t B, int n_gpus);
int grbda_rnea_sharded_f64(const grbda_plan *plan, const double *q, const double *qd, const double *ydd, double *tau,
                           size_t B, int n_gpus);

/* The same split for shards that are ALREADY RESIDENT on their devices ("batches of independent states shard
 * trivially across the GPUs of one node with a gather of results only"; the reference has no counterpart -- one
 * ClusterTreeModel::forwardDynamics call is one state on one core, ClusterTreeDynamics.cpp:85-152).  One process, n_gpus
 * shards: shard g = B[g] states in DEVICE arrays q[g], qd[g], tau[g] on device devices[g] (devices == NULL: 0 .. n_gpus-1),
 * launched on streams[g] (streams == NULL or a NULL entry: that device's default stream; two shards on one device need
 * distinct streams).  The plan is replicated per device on first use; there is no data-path collective.
 *   gathered == NULL: every shard writes ydd[g] (its own device); nothing else happens.
 *   gathered != NULL: a DEVICE array [sum B][nv] on devices[0]; the slab of shard g lands at row offset B[0] + .. + B[g-1] by
 *     hipMemcpyPeerAsync on streams[g] (peer access is enabled where the devices allow it: the copy then crosses xGMI with no
 *     host hop); a shard on devices[0] computes straight into its place when ydd is NULL or ydd[g] is NULL.  Shards on other
 *     devices need their own output slab ydd[g].  streams[0] is made to wait for every shard's copy (events), so work enqueued
 *     on streams[0] after the call sees the whole gathered array.
 * Enqueues only, like every device-pointer entry point: no synchronisation, no host copies.  GRBDA_EINVAL for null / out of
 * range arguments, GRBDA_ENODEVICE without a HIP device. */
int grbda_aba_sharded_dev_f32(const grbda_plan *plan, int n_gpus, const int *devices, const float *const *q, const float *const *qd,
                              const float *const *tau, float *const *ydd, const size_t *B, void *const *streams, float *gathered);
int grbda_aba_sharded_dev_f64(const grbda_plan *plan, int n_gpus, const int *devices, const double *const *q, const double *const *qd,
                              const double *const *tau, double *const *ydd, const size_t *B, void *const *streams, double *gathered);
int grbda_rnea_sharded_dev_f32(const grbda_plan *plan, int n_gpus, const int *devices, const float *const *q, const float *const *qd,
                               const float *const *ydd, float *const *tau, const size_t *B, void *const *streams, float *gathered);
int grbda_rnea_sharded_dev_f64(const grbda_plan *plan, int n_gpus, const int *devices, const double *const *q, const double *const *qd,
                               const double *const *ydd, double *const *tau, const size_t *B, void *const *streams, double *gathered);

/* ---- measurement hook -------------------------------------------------------------------------- */
/* Average duration in milliseconds of `iters` back-to-back launches of the ABA (kind 0) or RNEA
 * (kind 1) kernel, measured with hipEvents recorded on `stream` around the launches (the stream
 * the kernel runs on).  precision: 32 or 64. */
int grbda_time_kernel(const grbda_plan *plan, int kind, int precision, const void *q, const void *qd,
                      const void *x, void *out, size_t B, int device, void *stream, int iters,
                      float *avg_ms);

/* Debug aid (tools/spec_experiment.py): the f32 fast-path tables of a plan written as C initialisers. */
int grbda_debug_dump_plan(const grbda_plan *plan, const char *path);

/* number of usable HIP devices (0 when there is none); never fails */
int grbda_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* GRBDA_HIP_H */
