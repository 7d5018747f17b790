/*
 * grbda_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64, dense formulation) of the reference's cluster-ABA /
 * cluster-RNEA path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (generalized_rbda_amd/, include/grbda_hip.h) never does.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py) against
 * golden vectors produced by the reference's own closed-form CasADi codegen
 * (/root/reference/src/Codegen/rev_*_FD.cpp / *_ID.cpp, built by oracle/Makefile into
 * oracle/_ref/), against the reference's Projection identity
 * (src/Dynamics/RigidBodyTreeDynamics.cpp:86-97) and ID(FD(tau)) == tau
 * (UnitTests/testRigidBodyDynamicsAlgos.cpp:208-235).
 *
 * All functions take a model-description blob (include/grbda_model_desc.h) and row-major
 * batches q[B][nq], qd[B][nv], tau/ydd[B][nv]; f_ext may be NULL or [B][n_bodies][6]
 * (world-frame spatial forces, TreeModel::setExternalForces semantics).
 * Return 0 on success, negative error code otherwise.
 */
#ifndef GRBDA_ORACLE_H
#define GRBDA_ORACLE_H

#include <stddef.h>

/* Scalar type of the state arrays and of all arithmetic.  double: the parity checker (the only build tests/ and smoke()
 * use).  float (-DGRBDA_ORACLE_REAL=float, _build/libgrbda_oracle_f32.so): the same code as a single-precision CPU
 * baseline for bench.py.  The model-description blob holds doubles in both builds. */
#ifndef GRBDA_ORACLE_REAL
#define GRBDA_ORACLE_REAL double
#endif
typedef GRBDA_ORACLE_REAL grbda_real;

#ifdef __cplusplus
extern "C" {
#endif

enum {
    GRBDA_ORACLE_OK = 0,
    GRBDA_ORACLE_EBADBLOB = -1,
    GRBDA_ORACLE_EUNSUPPORTED = -2,
    GRBDA_ORACLE_ESINGULAR = -3,
    GRBDA_ORACLE_EINVALIDSTATE = -4,
    GRBDA_ORACLE_ENOMEM = -5
};

/* cluster ABA: ClusterTreeModel::forwardDynamics (src/Dynamics/ClusterTreeDynamics.cpp:85-191) */
int grbda_oracle_forward_dynamics(const void *blob, size_t bytes, const grbda_real *q, const grbda_real *qd,
                                  const grbda_real *tau, const grbda_real *f_ext, grbda_real *ydd, size_t B);

/* cluster RNEA: TreeModel::recursiveNewtonEulerAlgorithm (src/Dynamics/TreeModel.cpp:173-212) */
int grbda_oracle_inverse_dynamics(const void *blob, size_t bytes, const grbda_real *q, const grbda_real *qd,
                                  const grbda_real *ydd, const grbda_real *f_ext, grbda_real *tau, size_t B);

/* independent check: spanning-tree CRBA + RNEA + Projection
 * (RigidBodyTreeDynamics.cpp:86-97, TreeModel.cpp:115-171); returns independent ydd */
int grbda_oracle_forward_dynamics_projection(const void *blob, size_t bytes, const grbda_real *q,
                                             const grbda_real *qd, const grbda_real *tau,
                                             const grbda_real *f_ext, grbda_real *ydd, size_t B);

/* same algorithm as grbda_oracle_forward_dynamics, batch statically partitioned over
 * n_threads pthreads (cpu_baseline leg of bench.py) */
int grbda_oracle_forward_dynamics_mt(const void *blob, size_t bytes, const grbda_real *q,
                                     const grbda_real *qd, const grbda_real *tau, grbda_real *ydd, size_t B,
                                     int n_threads);

/* loop-constraint quantities of cluster c for one state (tests of K G = 0, K g = k):
 * G[n_span_vel*n_vel], g[n_span_vel], K[rows*n_span_vel], k[rows], phi[rows] (any may be NULL) */
int grbda_oracle_cluster_constraint(const void *blob, size_t bytes, int cluster, const grbda_real *q,
                                    const grbda_real *qd, grbda_real *G, grbda_real *g, grbda_real *K, grbda_real *k,
                                    grbda_real *phi);

/* spanning state of every cluster, cluster after cluster (ClusterJoint.cpp:22-71): q_span[B][sum n_span_pos],
 * qd_span[B][sum n_span_vel]; gmax[B]: largest |G| entry of the dependent rows of the implicit clusters; kcond[B]: largest
 * |Kd|_F |Kd^-1|_F of their dependent blocks.  Any output may be NULL. */
int grbda_oracle_spanning_state(const void *blob, size_t bytes, const grbda_real *q, const grbda_real *qd,
                                grbda_real *q_span, grbda_real *qd_span, grbda_real *gmax, grbda_real *kcond, size_t B);

/* Newton projection of the dependent spanning positions of every implicit cluster onto
 * phi(q) = 0 (GenericJoint.cpp:289-385); q is [B][nq], modified in place.
 * ok[B] (may be NULL) receives 1 when ||phi|| < 1e-8 was reached for every cluster. */
/* absolute transforms world -> body (E 9 row-major, r 3) of every body: out[B][n_bodies][12] */
int grbda_oracle_body_poses(const void *blob, size_t bytes, const grbda_real *q, grbda_real *out, size_t B);
int grbda_oracle_project_positions(const void *blob, size_t bytes, grbda_real *q, size_t B,
                                   int max_iter, int *ok);

#ifdef __cplusplus
}
#endif
#endif
