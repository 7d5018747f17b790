/*
 * grbda_model_desc.h -- flat, position-independent description of a cluster tree model.
 *
 * This is the *raw* model (bodies, single joints, loop constraints) exactly as a
 * grbda::ClusterTreeModel holds it after construction
 * (reference: include/grbda/Dynamics/Body.h:16-43, ClusterTreeModel.cpp:10-67,
 *  ClusterJoints/LoopConstraint.h:14-59).  It is NOT the optimised device plan:
 *   - the C++ facade (generalized_rbda_amd/include/grbda) and the URDF+ reader WRITE it,
 *   - grbda_plan_from_blob() (include/grbda_hip.h) compiles it into the device plan,
 *   - the CPU oracle (oracle/) reads it directly, so the product's plan compiler and
 *     the checker never share derived data.
 *
 * Layout of a blob (little endian, 8-byte aligned):
 *   grbda_desc_header
 *   grbda_desc_body    bodies[n_bodies]      (cluster-major: a cluster's bodies are contiguous,
 *                                             ordered by sub_index_within_cluster)
 *   grbda_desc_cluster clusters[n_clusters]  (topological: parent cluster index < own index)
 *   int32_t            ints[n_ints]          (padded to a multiple of 2)
 *   double             doubles[n_doubles]
 *   char               names[n_name_bytes]   (n_bodies + n_clusters NUL-terminated strings; optional)
 */
#ifndef GRBDA_MODEL_DESC_H
#define GRBDA_MODEL_DESC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRBDA_DESC_MAGIC 0x44425247u /* "GRBD" */
#define GRBDA_DESC_VERSION 1u

/* single-joint types (reference: include/grbda/Dynamics/Joints/Joint.h:43-102) */
enum { GRBDA_JOINT_REVOLUTE = 0, GRBDA_JOINT_FREE = 1 };

/* orientation representation of a Free joint (OrientationRepresentation.h:11-49) */
enum { GRBDA_ORI_QUATERNION = 0, GRBDA_ORI_RPY = 1 };

/* loop-constraint kinds */
enum {
    /* LoopConstraint::Static (LoopConstraint.cpp:38-52): constant G (n_span_vel x n_vel) and
       K (n_rows x n_span_vel); g = k = 0.  doubles = G row-major, then K row-major. */
    GRBDA_CONSTRAINT_STATIC = 0,
    /* LoopConstraint::Free (ClusterJoints/FreeJoint.h): identity, no payload */
    GRBDA_CONSTRAINT_FREE = 1,
    /* LoopConstraint::GenericImplicit built by ClusterTreeParsing.cpp:310-376 from URDF+ <loop>
       elements: phi = translation(via predecessor) - translation(via successor).
       ints    = [n_loops, is_independent[k],
                  per loop: n_pred, pred_sub[n_pred] (nca->predecessor order),
                            n_succ, succ_sub[n_succ], axis_mask (bit j set = row j enforced)]
       doubles = per loop: pred origin E[9] r[3], succ origin E[9] r[3]                      */
    GRBDA_CONSTRAINT_LOOP_POSITION = 2,
    /* Trig-polynomial implicit constraint -- the data form of hand-written phi lambdas such as the
       Tello hip / knee-ankle differentials (src/Robots/Tello.cpp:139-163,237-261):
         phi_r(q) = sum_t coef * prod_f F_f(w_f . q + b_f),  F in {0: identity, 1: sin, 2: cos}
       ints    = [is_independent[n_span_vel], per row: n_terms, per term: n_factors, type[n_factors]]
       doubles = per row, per term: coef, per factor: w[n_span_pos], b                              */
    GRBDA_CONSTRAINT_TRIG_POLY = 3
};

typedef struct {
    uint32_t magic;
    uint32_t version;
    int32_t n_bodies;
    int32_t n_clusters;
    int32_t nq; /* total positions  (ClusterTreeModel::getNumPositions)        */
    int32_t nv; /* total velocities (ClusterTreeModel::getNumDegreesOfFreedom) */
    int32_t ori_repr;
    int32_t n_ints;
    int32_t n_doubles;
    int32_t n_name_bytes;
    int32_t reserved[2];
    double gravity[6]; /* TreeModel::gravity_ (TreeModel.h:19-22): default {0,0,0,0,0,-9.81} */
} grbda_desc_header;

typedef struct {
    int32_t parent;     /* global body index of the tree parent, -1 = ground (Body::parent_index_) */
    int32_t cluster;    /* index of the containing cluster */
    int32_t sub_index;  /* Body::sub_index_within_cluster_ */
    int32_t joint_type; /* GRBDA_JOINT_* */
    int32_t axis;       /* revolute: 0 = X, 1 = Y, 2 = Z (ori::CoordinateAxis) */
    int32_t reserved[3];
    double Xtree_E[9];   /* rotation, row-major (spatial::Transform::E_) */
    double Xtree_r[3];   /* translation (spatial::Transform::r_)         */
    double inertia[36];  /* 6x6 spatial inertia, row-major (SpatialInertia::getMatrix) */
} grbda_desc_body;

typedef struct {
    int32_t parent_cluster; /* -1 = root */
    int32_t first_body;
    int32_t n_bodies;
    int32_t q_index; /* TreeNode::position_index_ */
    int32_t n_pos;   /* ClusterJoints::Base::numPositions() (spanning count for implicit kinds) */
    int32_t v_index; /* TreeNode::velocity_index_ */
    int32_t n_vel;   /* numVelocities(): independent velocities */
    int32_t n_span_pos;
    int32_t n_span_vel;
    int32_t constraint_type; /* GRBDA_CONSTRAINT_* */
    int32_t n_constraint_rows;
    int32_t int_offset;
    int32_t n_int;
    int32_t dbl_offset;
    int32_t n_dbl;
    int32_t reserved;
} grbda_desc_cluster;

#ifdef __cplusplus
}
#endif

#endif /* GRBDA_MODEL_DESC_H */
