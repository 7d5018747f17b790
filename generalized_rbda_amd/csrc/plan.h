// plan.h -- the immutable device plan a ClusterTreeModel is compiled into.
//
// A plan is a small wave-uniform "program" for the batched dynamics kernels (kernels.hip):
//   * steps[]     : ordered list of (op, cluster) pairs -- forward / backward / acceleration
//                   sweeps of cluster ABA (reference: src/Dynamics/ClusterTreeDynamics.cpp:85-191)
//                   and of cluster RNEA (src/Dynamics/TreeModel.cpp:34-57,173-212),
//   * clusters[]  : per-cluster integer record,
//   * bodies[]    : per-body integer record (tree topology + state-slot numbers),
//   * consts[]    : model constants in the kernel's scalar type (Xtree, inertias, G rows),
// All of it is wave-uniform: every lane of a wavefront evaluates a different robot state of the
// SAME model, so the program is fetched through the scalar unit (the tables are addressed in the
// constant address space) while the vector unit does the per-state arithmetic.
//
// Per-state intermediates live in numbered "slots": slot s of lane l is element [s][l] of a
// per-wavefront array, so a wave's access to one slot is a fully coalesced 64-element row.
// The plan compiler schedules the sweeps depth-first (a limb is swept forward, then backward,
// before the next limb starts), computes the live range of every intermediate and packs them
// with an interval allocator: the hottest objects into the LDS budget, the rest into a small
// per-wave global slab.  A slot number with kSlotGlobal set addresses the global slab.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace grbda_hip {

enum StepOp : int32_t {
    OP_ABA_FWD = 0,   // velocity propagation for one cluster            (TreeModel.cpp:6-32)
    OP_ABA_BWD = 1,   // articulated inertia + bias back-propagation     (ClusterTreeDynamics.cpp:108-129,157-191)
    OP_ABA_ACC = 2,   // acceleration forward-propagation, writes ydd    (ClusterTreeDynamics.cpp:131-152)
    OP_RNEA_FWD = 3,  // velocity + acceleration + body force            (TreeModel.cpp:34-57,181-186)
    OP_RNEA_BWD = 4   // joint torque projection + force back-propagation (TreeModel.cpp:196-209)
};
// ORed into Step::op: the fast kernels have nothing to do in this step (forward ABA step / backward RNEA
// step of a SHAPE_REV / SHAPE_REV_ROTOR cluster whose link is a leaf); the general kernels run it
constexpr int32_t kOpSkipFast = 1 << 8;
constexpr int32_t kOpMask = 0xff;

enum ClusterKind : int32_t {
    CK_STATIC = 0,  // revolute single joints, constant G (every explicit ClusterJoint type)
    CK_FREE = 1,    // floating base root (ClusterJoints::Free)
    CK_LOOP = 2     // implicit position-loop constraint (LoopConstraint::GenericImplicit built from URDF+
                    // <loop> elements, ClusterTreeParsing.cpp:310-376): G(q), g(q, qd) per state
};

constexpr int kMaxClusterDof = 4;     // n of a non-free cluster handled in registers
constexpr int kMaxClusterBodies = 8;  // k
constexpr int kMaxConstraintRows = 6; // rows of an implicit cluster: 1 - 3 in the structured kernels, 4 - 6 through the spanning tree (wide kernels)
// clusters beyond those two limits run through the spanning tree only (HostPlan::big_clusters; manifold_kernels.hip's wide variants)
constexpr int kBigClusterBodies = 48;
constexpr int kBigClusterDof = 48;
constexpr int kWave = 64;
constexpr int32_t kSlotGlobal = 1 << 30;

struct Step {
    int32_t op;
    int32_t cluster;
    int32_t reserved[2];
};

struct ClusterRec {
    int32_t kind;
    int32_t first_body;
    int32_t k;
    int32_t n;
    int32_t q_index;
    int32_t v_index;
    int32_t parent_body;  // global index of the single body of the parent cluster all in-cluster
                          // roots hang off, or -1 (ground)
    int32_t chained;      // 1: some body has an in-cluster parent
    int32_t slot_K;       // n x 6 : D^-1 F^T   (ABA_BWD -> ABA_ACC)
    int32_t slot_y0;      // n     : D^-1 u'
    int32_t parent_slot_IA;   // backward-sweep accumulators of parent_body (or -1)
    int32_t parent_slot_psi;
    int32_t parent_slot_v3;   // acceleration-sweep velocity / acceleration of parent_body (or -1)
    int32_t parent_slot_a3;
    int32_t carry_out;        // 1: the contribution to parent_body is handed over in registers to the
                              //    next step (= backward step of the parent cluster) instead of slots
    int32_t rows;             // implicit clusters: number of constraint rows (= dependent coordinates)
    // implicit clusters (kernels.hip, ImpLayout): slot_imp_fwd = block kept across the sweeps
    // [G rows k*(n+1)][qd_span k][q_span k], slot_imp_bwd = work space of the evaluation [K rows*k][chain 6k]
    int32_t slot_imp_fwd, slot_imp_bwd, slot_imp_acc;
    int32_t iofs;             // offset into cints[]: n_loops, n_ind, ind[], n_dep, dep[], loops...
    int32_t dofs;             // offset into consts[]: per loop pred origin E[9] r[3], succ origin E[9] r[3]
    int32_t corr_first_IA;    // 1: the cluster's -F D^-1 F^T term is the first inertia written to the parent's slot
    int32_t cons_type;        // implicit clusters: 0 position loops (URDF+ <loop>), 1 trig-polynomial phi
    int32_t child_mask;       // bit i: body i of the cluster has children (forward / acceleration sweeps skip the rest
                              // without fetching their records)
    int32_t shape;            // ClusterShape: clusters the fast kernels run through straight-line handlers
    int32_t link_body;        // SHAPE_REV / SHAPE_REV_ROTOR: global index of the link ...
    int32_t rotor_body;       // ... and of the rotor
    int32_t hot;              // shape clusters inside a chain: the link has children, its backward accumulators arrive in
                              // registers and its contribution leaves in registers (kernels.hip, aba_bwd_rev<.., HOT>)
};

// Shapes with a dedicated handler.  SHAPE_REV: one revolute body, one coordinate.  SHAPE_REV_ROTOR: a link
// and an axisymmetric leaf (rotor) driven by one coordinate, both children of the cluster's parent body.
enum ClusterShape : int32_t { SHAPE_GENERIC = 0, SHAPE_REV = 1, SHAPE_REV_ROTOR = 2 };

struct BodyRec {
    int32_t parent;       // global tree-parent body index or -1
    int32_t lam;          // in-cluster parent: global body index, or -1
    int32_t axis;         // revolute axis 0/1/2
    int32_t jtype;        // 0 revolute, 1 free
    int32_t has_child;    // some body (any cluster) has this body as tree parent
    int32_t cofs;         // offset into consts[]: Et[9] rt[3] I[21] G_row[n]
    int32_t slot_sc;      // sin, cos of the spanning joint angle (2) [free: E (9) + r (3)]; has_child only
    int32_t slot_v;       // spatial velocity (6), forward sweep -> backward sweep; has_child only
    int32_t slot_IA;      // composite articulated inertia accumulator (21, packed upper); has_child only
    int32_t slot_psi;     // bias force accumulator (6); has_child only
    int32_t slot_ccl;     // in-cluster bias acceleration (6), chained clusters only
    int32_t slot_v3;      // velocity (6) and acceleration (6) in the acceleration sweep; has_child only
    int32_t slot_a3;
    int32_t slot_f;       // RNEA: body force (6), all bodies
    int32_t parent_slot_v;    // the tree parent's slots, -1 if none
    int32_t parent_slot_IA;
    int32_t parent_slot_psi;
    int32_t parent_slot_v3;
    int32_t parent_slot_a3;
    int32_t parent_slot_f;
    int32_t acc_first;        // 1: first contributor to the tree parent's backward accumulators
    int32_t carry_in;         // 1: backward accumulators arrive in registers (single child cluster, adjacent step)
    int32_t axisym;           // 1: leaf body whose inertia is invariant under rotation about its joint axis
                              //    (a rotor): X(q)^T I X(q) and X(q)^T (v x* I v) do not depend on q, so the
                              //    body is evaluated at q = 0 and its inertia contribution is a plan constant
    int32_t xofs;             // offset into consts[] of the 21 constants I + sum_children X0^T I X0 (or -1)
    int32_t acc_first_IA;     // like acc_first, counting only children that really accumulate an inertia
    int32_t slot_Xa;          // absolute transform world -> body (E 9, r 3); layouts with external forces, has_child only
    int32_t parent_slot_Xa;
    int32_t canon_axis;       // original joint axis when the body frame was re-expressed with the axis on z (plan.cpp,
                              // canonical joint axes), else 2: the reference's body frame is Rc(canon_axis)^T x this one
};

// number of constants per body before the G row
constexpr int kBodyConstFixed = 9 + 3 + 21;

// slot layout for one scalar width (the LDS budget in slots depends on sizeof(T))
struct Layout {
    std::vector<ClusterRec> clusters;       // ABA slots
    std::vector<ClusterRec> rnea_clusters;  // RNEA slots (slot_imp_fwd / slot_imp_bwd used)
    std::vector<BodyRec> bodies;       // ABA slots
    std::vector<BodyRec> rnea_bodies;  // RNEA slots (slot_sc, slot_v, slot_a3, slot_f used)
    int n_lds_aba = 0, n_glb_aba = 0;
    int n_lds_rnea = 0, n_glb_rnea = 0;
    // per ABA step: slot of the [K][y0] block the step reads when it is an acceleration step of a
    // straight-line shape, else -1; padded by one.  The kernel fetches entry s + 1 while it runs step s.
    std::vector<int32_t> acc_k;
    // split layouts (kernels.hip, Slots<T, true>): every [K | y0] block in the global slab, every other object in LDS;
    // false when the layout was not built that way or the LDS-only objects did not fit the budget
    bool split_aba = false, split_rnea = false;
};

// ---------------------------------------------------------------------------------------------------------------
// Chain program (chain_kernels.hip): the fast path of models made of a floating or fixed base, single revolute
// links, RevoluteWithRotor clusters (link + axisymmetric rotor) and leaf RevolutePairWithRotor clusters -- every
// legged robot of the reference that has a URDF.  The cluster tree is cut into CHAINS: maximal paths along which
// every link carries exactly one child cluster.  A sweep over a chain is ONE tight loop over its links (a run):
// the velocity / acceleration going down and the projected inertia / bias going up stay in registers from link to
// link, and only what another segment needs goes through LDS.  Between segments everything is handed over through
// LDS slots, so the segment loop of the kernel carries no vector state at all.
// ---------------------------------------------------------------------------------------------------------------
enum ChainOp : int32_t {
    SEG_FREE_FWD = 0,   // floating base: velocity to its LDS slot
    SEG_RUN_FWD = 1,    // links of a chain, root side first: sin / cos and velocity of every link with children
    SEG_RUN_BWD = 2,    // leaf side first: [K | y0] of every link; (IA, psi) accumulated into the parent's slot at the end
    SEG_FREE_BWD = 3,
    SEG_FREE_ACC = 4,
    SEG_RUN_ACC = 5,    // root side first: ydd of every link; (v, a) of the tip to LDS when child segments follow
    SEG_PAIR_ACC = 6,
    SEG_DIFF_FWD = 7,   // two-rotor differential cluster (ChainDiff): constraint Jacobian, G, g; velocity of its tip link
    SEG_DIFF_BWD = 8,
    SEG_DIFF_ACC = 9,
    SEG_BARRIER = 10,   // latency-mode programs (ChainProgram::n_waves > 1): every wavefront of the workgroup meets here
    SEG_GEN_FWD = 11,   // generic cluster (ChainGen): constraint Jacobian / G / g of an implicit cluster, velocities of its bodies
    SEG_GEN_BWD = 12,   // ... articulated inertia / bias of its bodies, [K | y0], hand-over to the parent body
    SEG_GEN_ACC = 13    // ... ydd; (v, a) of the bodies that carry child clusters
};
enum ChainHead : int32_t {
    HEAD_LEAF = 0,      // the first link of a backward run is a leaf: its accumulators start from its own inertia
    HEAD_SLOT = 1,      // ... carries several child segments: accumulators come from its LDS slot
    HEAD_PAIR = 2       // ... carries one leaf RevolutePairWithRotor cluster, evaluated at the start of the run
};

// one REV / REV_ROTOR cluster inside a run (16 ints: one s_load_dwordx16)
struct ChainLink {
    int32_t q_index, v_index;
    int32_t cofs;       // link constants: Et[9] rt[3] I[21] g0 (consts[])
    int32_t rofs;       // rotor constants (same layout), or -1: plain revolute cluster
    int32_t iofs;       // the 21 constants I_link + sum over axisymmetric leaf children X0^T I X0 (BodyRec::xofs or cofs + 12)
    int32_t lds_sv;     // LDS slot of [sin, cos, v 6], forward run -> backward run
    int32_t lds_pv;     // LDS slot of the parent body's velocity (the v part of its lds_sv block), -1: ground
    int32_t glb_k;      // global slab slot of [K 6][y0][sin][cos], backward run -> acceleration run
    int32_t has_child;
    int32_t lds_va;     // acceleration sweep: LDS slot of [v 6][a 6] when another segment reads them, else -1
    int32_t rpre;       // axisymmetric rotor: state-independent constants [X0^T h (6)][h_z] with h = I_rotor[:, z] (plan.cpp);
                        // -1 with rofs >= 0: a general rotor, evaluated at its own angle
    int32_t perm;       // 0, 1, 2: the tree rotation Et is the cyclic axis permutation (Et x)_i = x_((i + perm) % 3), the kernels use
                        // the permutation-structured transforms (devmath.h, rzp_*); -1: general rotation
    int32_t rperm;      // the same for the tree rotation of a general rotor (rofs >= 0, rpre < 0)
    int32_t reserved[3];
};

// a RevolutePairWithRotor-shaped leaf cluster (32 ints)
struct ChainPair {
    int32_t q_index, v_index;
    int32_t cofs[4];    // constants of link1, link2, rotor1, rotor2: Et[9] rt[3] I[21] G row [2]
    int32_t lds_pv;     // parent body's velocity
    int32_t glb_k;      // [K 12][y0 2] (+ 7 rows written by the OSIM pass)
    int32_t lds_pva;    // acceleration sweep: parent body's [v 6][a 6]
    int32_t rpre[2];    // rotor1, rotor2: [X0^T h (6)][h_z]
    int32_t perm[2];    // link1, link2: ChainLink::perm
    int32_t reserved[19];
};

// An implicit two-rotor differential (the Tello hip and knee-ankle differentials, src/Robots/Tello.cpp:77-261 with
// TelloDifferential / GenericImplicit constraints): link1 and two axisymmetric rotors on the parent body, link2 on link1,
// a trig-polynomial constraint phi(q) = 0 (2 rows) that makes the two ROTOR angles the independent coordinates and the two
// link angles the dependent ones.  Kernel coordinate order: [rotor1, rotor2, link1, link2]; G = [1; X], g = (0, 0, g1, g2)
// with X = -Kd^-1 Ki, g = -Kd^-1 Kdot qd evaluated per state by the forward segment (32 ints).
struct ChainDiff {
    int32_t q_index, v_index;
    int32_t qpos[4];    // offsets of the spanning positions of rotor1, rotor2, link1, link2 from q_index
    int32_t cofs[4];    // constants of link1, link2, rotor1, rotor2
    int32_t rpre[2];    // rotor1, rotor2: [X0^T h (6)][h_z]
    int32_t iofs;       // link2: the 21 constants I + sum over axisymmetric leaf children X0^T I X0
    int32_t lds_pv;     // parent body's velocity
    int32_t lds_sv;     // [sin, cos, v 6] of link2 when child segments read it, else -1
    int32_t lds_acc;    // accumulator [IA 21][psi 6] of link2 (child segments), -1: link2 is a leaf
    int32_t lds_acc_out, acc_first;  // accumulator of the parent body; 1: this segment is its first writer
    int32_t glb_k;      // [K 12][y0 2][X 4][g 2][s1 c1 s2 c2]
    int32_t lds_pva;    // acceleration sweep: parent body's [v 6][a 6]
    int32_t lds_va;     // own [v 6][a 6] of link2 when child segments follow, else -1
    int32_t lds_w;      // work space of the constraint evaluation: 3 slots per atom (DiffProgram)
    int32_t tofs_i, tofs_d;  // constraint program in cints[] / consts[] (plan.cpp, emit_diff_program); tofs_i = -1: an EXPLICIT
                             // cluster of the same shape -- RevolutePairWithRotor with child clusters on link2 or in a place the
                             // leaf-pair head of a run does not cover: the link angles are the coordinates (X = 1, g = 0)
    int32_t gofs;            // consts[]: G rows of the two rotors, (1, 0, 0, 1) for a differential
    int32_t reserved[7];
};
// ---------------------------------------------------------------------------------------------------------------
// Generic cluster inside a chain program (chain_kernels.hip, gen_segments.h): k <= 8 revolute bodies in any in-cluster tree,
// n <= 4 independent coordinates, and either a constant G (every explicit ClusterJoint type that has no segment type of
// its own: RevoluteTripleWithRotor, Generic + LoopConstraint::Static, pairs and rotors in unusual places) or an implicit
// constraint evaluated per state -- URDF+ position loops (ClusterTreeParsing.cpp:310-376: four-bar, six-bar, planar leg
// linkage) and trig-polynomial phi that is not the two-rotor differential's shape (GenericJoint.cpp:57-90,387-469).
// Child clusters may hang off ANY of its bodies.  One body of the cluster (16 ints):
struct ChainGenBody {
    int32_t cofs;       // Et[9] rt[3] I[21] (+ G row [n] for explicit clusters)
    int32_t iofs;       // the 21 constants I + sum over axisymmetric leaf children X0^T I X0 (BodyRec::xofs or cofs + 12)
    int32_t lam;        // in-cluster parent: index within the cluster, or -1 (the body hangs off the cluster's parent body)
    int32_t axis;       // joint axis: always 2 (canonical joint axes, plan.cpp); kept for the record
    int32_t axisym;     // axisymmetric leaf (rotor): evaluated at q = 0, X0^T I X0 is part of the parent's constants
    int32_t carry_up;   // 1: lam is the body right before this one and (IA, psi) go up in registers
    int32_t carry_in;   // 1: the body right behind this one hands its (IA, psi) up in registers
    int32_t acc_w;      // work-area offset of the accumulator [IA 21][psi 6] in-cluster children without carry_up add into, or -1
    int32_t up_w;       // (no carry_up, lam >= 0) work-area offset of lam's accumulator ...
    int32_t up_first;   // ... and 1 when this body is its first writer
    int32_t lds_acc;    // accumulator [IA 21][psi 6] the child CLUSTERS of this body add into (LDS / global slab), or -1
    int32_t lds_va;     // acceleration sweep: [v 6][a 6] of this body (it or an in-cluster descendant carries child clusters), or -1
    int32_t pva;        // acceleration sweep: lds_va of lam (lam >= 0)
    int32_t ind_a;      // implicit clusters: this body carries independent coordinate ind_a (G row = unit vector, g = 0), or -1
    int32_t dep_r;      // implicit clusters: this body is dependent coordinate dep_r (its [G row n][g][qd_span] is row dep_r of the kept block), or -1
    int32_t lds_v;      // [v 6] of this body for the segments of its child clusters (forward segment -> backward segment), or -1
};
// the cluster (32 ints)
struct ChainGen {
    int32_t q_index, v_index;
    int32_t k, n, rows;
    int32_t kind;       // 0 explicit (constant G rows behind the bodies' constants), 1 URDF+ position loops, 2 trig-polynomial phi
    int32_t first;      // gbodies[first .. first + k)
    int32_t iofs, dofs; // implicit: constraint payload in cints[] / consts[] (ClusterRec::iofs / dofs)
    int32_t lds_pv;     // velocity of the parent body, -1: ground
    int32_t has_parent; // the cluster hangs off a body (0: off the ground -- nobody to hand the projected inertia to)
    int32_t lds_acc_out, acc_first;  // accumulator of the parent body; 1: this segment is its first writer
    int32_t lds_pva;    // acceleration sweep: [v 6][a 6] of the parent body, -1: ground
    int32_t glb_k;      // [K 6 n][y0 n]
    int32_t keep;       // implicit: per DEPENDENT body [G row n][g][qd_span], forward / backward segment -> acceleration segment (LDS inside
                        // the work area for a cluster without child clusters, else the global slab)
    int32_t lds_w;      // work area of the backward segment (LDS): [sin, cos] x k | [v 6] x k (the constraint evaluation's scratch
                        // aliases it) | in-cluster accumulators | kept block of a childless implicit cluster
    int32_t has_fwd;    // 1: a SEG_GEN_FWD segment exists (child clusters need the velocities): it evaluates the constraint (kept block in
                        // the slab) and leaves the velocities of the bodies with child clusters in their lds_v; the backward segment
                        // repeats the downward pass without the constraint -- no work area stays alive across the child segments
    int32_t need_acc;   // 1: some body has lds_va >= 0
    int32_t lds_wf;     // work area of the forward segment (has_fwd)
    int32_t reserved[12];
};

// the same cluster in the inverse-dynamics program (20 ints)
struct RneaDiff {
    int32_t q_index, v_index;
    int32_t qpos[4];
    int32_t cofs[4];
    int32_t lds_pva;    // parent body's [v 6][a 6]
    int32_t lds_pf;     // parent body's force
    int32_t lds_blk;    // [f2 6][s1 c1 s2 c2][X 4]  (14), forward -> backward segment
    int32_t lds_va;     // [v 6][a 6] of link2 for child segments (lds_blk + 14), else -1
    int32_t lds_w;      // work space of the constraint evaluation (shares lds_blk: plan.cpp)
    int32_t tofs_i;     // constraint program in cints[], -1: explicit pair (see ChainDiff)
    int32_t gofs;       // consts[]: G rows of the two rotors
    int32_t reserved[3];
};

struct ChainSeg {       // 16 ints
    int32_t op;
    int32_t first, count;   // runs: links[first .. first + count) in sweep order; pair / free: record index
    int32_t head;           // SEG_RUN_BWD: ChainHead
    int32_t head_arg;       // HEAD_SLOT: accumulator slot [IA 21][psi 6] (LDS, or global slab when kSlotGlobal is set);
                            // HEAD_PAIR: index into pairs[]
    int32_t lds_acc_out;    // SEG_RUN_BWD: accumulator slot of the body the chain hangs off, -1: ground
    int32_t acc_first;      // 1: this segment is the first writer of that slot
    int32_t lds_pva;        // SEG_RUN_ACC: [v 6][a 6] of the body the chain hangs off, -1: ground (v = 0, a = a_root)
    int32_t owner;          // latency-mode programs: the wavefront of the workgroup that runs this segment (base segments: 0)
    int32_t rot_kind;       // runs: 1 -- every link carries an axisymmetric rotor (rofs >= 0, rpre >= 0), 2 -- no link carries a rotor,
                            // 0 -- mixed: the backward run picks a branch-free link body for 1 (chain_kernels.hip, run_bwd<.., ROT>)
    int32_t reserved[6];
};

struct ChainFree {      // 16 ints
    int32_t q_index, v_index, cofs, iofs;
    int32_t lds_v;      // forward: own velocity (children read it)
    int32_t lds_acc;    // backward: accumulator slot, -1 when the base has no children
    int32_t glb_y0;     // [y0 6]
    int32_t lds_va;     // acceleration sweep: own [v 6][a 6], -1 when no children
    int32_t lds_acc2;   // latency-mode programs: the accumulator the SECOND wavefront's limbs add into (-1: none)
    int32_t lds_acc3, lds_acc4;  // ... the third's and the fourth's (ChainProgram::n_waves = 4)
    int32_t reserved[5];
};

// ---- inverse dynamics on the same chains (chain_kernels.hip, rnea_chain_kernel) ------------------------------------
// forward run: v, a, body force f = I a + v x* I v of every link, the rotor's torque and its force on the parent body;
// backward run: tau = S^T f, f_parent += X^T f.  A leaf pair cluster is finished in one segment of the forward pass.
enum RneaChainOp : int32_t { RSEG_FREE_FWD = 0, RSEG_RUN_FWD = 1, RSEG_PAIR = 2, RSEG_RUN_BWD = 3, RSEG_FREE_BWD = 4,
                             RSEG_DIFF_FWD = 5, RSEG_DIFF_BWD = 6, RSEG_GEN_FWD = 7, RSEG_GEN_BWD = 8,
                             RSEG_BARRIER = 9 };  // latency-mode programs (RneaChainProgram::n_waves > 1): every wavefront of the workgroup meets here
struct RneaLink {       // 16 ints
    int32_t q_index, v_index;
    int32_t cofs, rofs;     // link / rotor constants (rofs -1: plain revolute cluster)
    int32_t lds_blk;        // LDS slot of [f 6][sin, cos][rotor torque]  (9)
    int32_t lds_va;         // LDS slot of [v 6][a 6] when child segments read them, else -1
    int32_t lds_pf;         // LDS slot of the parent body's force (first 6 of its block / the base's), -1: ground
    int32_t general_rotor;  // 1: the rotor is not axisymmetric about its axis and is evaluated at its own angle
    int32_t perm, rperm;    // ChainLink::perm / rperm
    int32_t reserved[6];
};
struct RneaPair {       // 16 ints
    int32_t q_index, v_index;
    int32_t cofs[4];        // link1, link2, rotor1, rotor2
    int32_t lds_pva;        // parent body's [v 6][a 6]
    int32_t lds_pf;         // parent body's force
    int32_t reserved[8];
};
struct RneaSeg {        // 8 ints
    int32_t op, first, count;
    int32_t lds_pva;        // RSEG_RUN_FWD: [v][a] of the body the chain hangs off, -1: ground
    int32_t lds_pf;         // RSEG_RUN_BWD: force slot of that body, -1: ground
    int32_t owner;          // latency-mode programs: the wavefront of the workgroup that runs this segment (the base's segments: 0)
    int32_t reserved[2];
};
struct RneaFree {       // 8 ints
    int32_t q_index, v_index, cofs;
    int32_t lds_va;         // own [v][a], -1 when no children
    int32_t lds_f;          // own force (children add theirs)
    int32_t lds_f2, lds_f3, lds_f4;  // latency-mode programs: what the limbs of the second .. fourth wavefront add into (-1: none)
};
struct RneaChainProgram {
    bool ok = false;
    std::vector<RneaSeg> segs;
    std::vector<RneaLink> links;
    std::vector<RneaPair> pairs;
    std::vector<RneaFree> frees;
    std::vector<RneaDiff> diffs;
    std::vector<ChainGen> gens;          // generic clusters: the records of the forward-dynamics program with the field use of
    std::vector<ChainGenBody> gbodies;   // gen_rnea_segments.h
    int n_lds = 0;
    int n_glb = 0;  // > 0: some link blocks live in the wave's global slab (their slot numbers carry kSlotGlobal)
    bool single_gen = false;  // the whole model is ONE generic cluster on the ground: rnea_gen1_kernel (no slab, fused sweeps)
    // Latency mode (n_waves = 2 or 4, as ChainProgram::n_waves): the limbs below the floating base are dealt to the wavefronts of a workgroup (RneaSeg::owner), the base's
    // forward segment runs first on wavefront 0, its backward segment last; two RSEG_BARRIER segments order the hand-overs.  Links and leaf pairs only.
    int n_waves = 1;
};

struct ChainProgram {
    bool ok = false;                 // the model is covered and its LDS objects fit the budget
    std::vector<ChainSeg> segs;
    std::vector<ChainLink> links;    // in sweep order of each run (a cluster appears once per sweep it takes part in)
    std::vector<ChainPair> pairs;
    std::vector<ChainFree> frees;
    std::vector<ChainDiff> diffs;
    std::vector<ChainGen> gens;
    std::vector<ChainGenBody> gbodies;
    bool single_gen = false;         // the whole model is ONE generic cluster on the ground: aba_gen1_kernel (no slab, fused sweeps)
    int n_lds = 0, n_glb = 0;        // slots
    int out_lds = -1;                // first of the nv LDS rows the acceleration sweep writes its results to (-1: slab rows)
    bool sv_global = false;          // the [sin, cos, v] blocks of the links live in the global slab (chains too long for LDS)
    // Latency mode (n_waves = 2 or 4): a tile is run by a WORKGROUP of n_waves wavefronts -- the limbs below the floating base are
    // dealt out to them, the base's own segments run on wavefront 0, SEG_BARRIER segments order the hand-overs (base velocity
    // -> limbs, limb accumulators -> base, base acceleration -> limbs).  For batches that do not fill the chip (fewer tiles
    // than SIMDs: BASELINE config 2) this halves the instruction stream a SIMD sees per tile.  LDS objects of limbs that run
    // on different wavefronts never share slots (separate pools); the base's accumulators live in the global slab.
    int n_waves = 1;
};

// composite-rigid-body algorithm (crba_kernels.hip): per body, where its per-state scratch rows live in the wave's slab
struct CrbaBody {
    int32_t cluster;  // index of the containing cluster
    int32_t acc_row;  // first of the 21 rows of the composite-inertia accumulator (bodies with non-axisymmetric children), or -1
    int32_t sc_row;   // rows [sin, cos] of the spanning joint angle
    int32_t reserved;
};
struct CrbaProgram {
    bool ok = false;  // explicit (constant G) clusters only
    std::vector<CrbaBody> bodies;
    int n_rows = 0;
};

// inverse-dynamics derivatives (deriv_kernels.hip): per body, where its per-state rows live in the wave's slab
// rnea_deriv_kernel (fp32) keeps the [S | Sd | Pdd] rows of the current root path in LDS: a body `level` bodies below the base (or the
// ground) owns block `level` of kDerivAncLevels; deeper bodies stay in the slab.  Which blocks are valid when a cluster walks up its
// ancestors is known at plan time (the clusters are processed in a fixed order): DerivBody::walk_resident.
constexpr int kDerivAncLevels = 8;
struct DerivBody {     // 10 ints
    int32_t cluster;
    int32_t kin_row;    // bodies with children: 24 rows [E 9][p 3][v 6][a 6], transform from / motion in the common frame F; else -1
    int32_t anc_row;    // revolute bodies with children: 18 rows [S 6][Sd 6][Pdd 6]; else -1
    int32_t acc_row;    // bodies with children: 63 rows [Ic 21][Bc 36][Fc 6], composite sums over the descendants; else -1
    int32_t acc_first;  // this body is the first writer of the accumulator of its in-cluster parent
    int32_t cluster_acc_first;  // (first body of a cluster) the cluster is the first writer of the accumulator of its parent body
    int32_t carry_out;  // (first body of a cluster) the cluster is the only contributor to its parent body and that body's
                        // cluster is processed next: the composites stay in registers instead of going through the slab
    int32_t carry_body; // (first body of a cluster) the body that receives them (processed first in its cluster), or -1
    int32_t anc_lds;    // bodies with an anc_row: their block of the LDS cache (their depth below the base), or -1: slab only
    int32_t walk_resident;  // (first body of a cluster) bit l: block l holds the cluster's ancestor at that depth when its walk starts
};

// H^-1 from the articulated-body quantities (minv_kernels.hip).  The cluster ABA's own factorisation of the joint-space inertia,
//   H^-1 = W^T W,   W = D^-1/2 (1 - psi):   row block of cluster a, column j (j in a or below it)
// (Rodriguez / Jain's innovations factorisation; what Carpentier's computeMinverse sweeps), replaces the dense Cholesky factorisation and the
// inversion of its factor in the derivative pipeline.  abi_factor_kernel (one state per lane) runs the articulated-inertia recursion in the common
// frame F of the derivative recursion and writes, per state, a RECORD BLOCK of n_entries scalars:
//   per non-free cluster c (n coordinates) one BLOCK per body that carries child clusters (one block when none does):
//       [K = F D^-1 (6 x n, column by column: force at the parent body per unit sigma)][L^-1 (D = L L^T; packed lower triangle, row-major)]
//       [S_ab (6 x n, column by column): motion of that body per unit cluster coordinate]        (K and L^-1 repeat in every block of c)
//   floating base at base_off:  L^-1 of its articulated inertia (packed lower triangle, 21)
// so that one offset names everything a step of the walk needs.
// minv_mfma_kernel (one state per wavefront, lane = column j) walks column j up its root path -- f = K_c e_j; at every ancestor cluster a
// (attached through body b): sigma = S_ab^T f, W[a rows][j] = -L_a^-1 sigma, f -= K_a sigma -- and multiplies on the matrix cores.
// coltab (uploaded as it is): per lane j < 64, kMinvColInts int32:
//   [0] offset of column e of K_c   [1] offset of L_c^-1   [2] n | e << 4 | v_index << 8 | ends_at_base << 20 | valid << 21
//   [3 + t] (t < kMinvMaxDepth): step t of the path, leaf side first:  block offset | n << 16 | v_index << 20 | valid << 31
constexpr int kMinvMaxDepth = 16;
constexpr int kMinvColInts = 3 + kMinvMaxDepth;
struct MinvBody {      // 3 ints per body
    int32_t blk_off;   // bodies that carry child clusters: offset of their block [K | L^-1 | S_ab]; else -1
    int32_t clus_off;  // (first body of a non-free cluster) offset of the cluster's first block; (base) offset of its L^-1; else -1
    // abi_factor_kernel stores the kinematics [E | p] of this body in its slab row (pass 1) because somebody LOADS them: a body of pass 1
    // whose parent is not the body processed right before it, an in-cluster child, or -- pass 2 -- a cluster that receives no carried
    // hand-over from a child cluster.  Along chains nobody does: pass 2 derives a parent's kinematics from its child's
    // (E_p = E_l^T E_i, p_p = p_i - E_p^T r) and the row is never written (JVRC-1: 11 of 33 rows stay).
    int32_t keep;
};
struct MinvProgram {
    bool ok = false;
    int n_entries = 0;   // scalars per state
    int base_off = -1;   // floating base: offset of its L^-1 (21 scalars)
    int max_depth = 0;   // longest path (ancestor clusters of a column, the base not counted)
    std::vector<MinvBody> bodies;
    std::vector<int32_t> coltab;   // [64][kMinvColInts]
};

struct DerivProgram {
    bool ok = false;  // explicit (constant G) clusters
    MinvProgram minv;
    std::vector<DerivBody> bodies;
    int n_rows = 0;
    int n_max = 1;    // largest number of coordinates of a cluster
    // (nv <= 64) bit i of related[j]: velocity coordinates i and j are on one root path (same cluster, ancestor or
    // descendant) -- the only entries of H, dID/dq and dID/dqd that are not structural zeros
    std::vector<uint64_t> related;
};

// LDS budget per wavefront, in slots, of each kernel (ABA / RNEA x f32 / f64).  Few slots mean more
// wavefronts per CU and more state in the global slab; the best trade differs per kernel.
struct LdsBudget {
    int aba32 = 0, aba64 = 0, rnea32 = 0, rnea64 = 0;
    int chain32w = 0;  // chain program at four wavefronts per SIMD
};

struct HostPlan {
    int nq = 0, nv = 0, n_bodies = 0, n_clusters = 0;
    int ori_repr = 0;
    // The cluster recursions do not cover this model (a cluster attached to several bodies of its parent cluster: the projected
    // inertia then couples those bodies, SpatialTransforms.cpp:312-344): forward / inverse dynamics, mass matrix and derivatives
    // run through the SPANNING TREE and the per-state G instead -- H = G^T H_s G, ydd = H^-1 (tau - G^T (C_s + H_s g)), the
    // reference's own cross-check (RigidBodyTreeDynamics.cpp:86-97) -- capi.cpp projection_run; no sweep programs are built.
    bool projection_only = false;
    bool big_clusters = false;  // a cluster exceeds kMaxClusterBodies / kMaxClusterDof: spanning-tree route, forward / inverse dynamics and H only
    // (plans of that route with more than 64 velocities -- and their spanning plans -- where the one-word masks of DerivProgram::related
    // end: capi.cpp, build_related_table) nv x nv, 1 where two coordinates lie on one root path
    std::vector<int32_t> related_table;
    double gravity[6] = {0, 0, 0, 0, 0, -9.81};
    std::vector<Step> aba_steps;
    std::vector<Step> rnea_steps;
    std::vector<double> consts;  // converted to float on upload for the f32 kernels
    std::vector<int32_t> cints;  // integer payload of implicit constraints
    Layout lay32, lay64;      // fast path
    Layout lay32x, lay64x;    // with absolute transforms kept for external forces (TreeNode::Xa_)
    Layout lay32s;            // f32 fast path, split layout (used when split_aba / split_rnea)
    ChainProgram chain32;     // f32 ABA, chain-structured fast path (chain_kernels.hip), two wavefronts per SIMD
    ChainProgram chain32w;    // the same laid out for four wavefronts per SIMD (half the LDS per wavefront)
    ChainProgram chain64;     // f64 ABA (slots are twice as large: the LDS budget holds half as many)
    ChainProgram chain32p, chain64p;  // latency mode: two wavefronts per tile (ChainProgram::n_waves)
    ChainProgram chain32q, chain64q;  // latency mode, four wavefronts per tile (batches of at most two tiles per CU)
    CrbaProgram crba;
    DerivProgram deriv;
    RneaChainProgram rchain32, rchain64;  // inverse dynamics on the chains
    RneaChainProgram rchain32w;           // f32 laid out for four wavefronts per SIMD (half the LDS per wavefront)
    RneaChainProgram rchain32p, rchain64p, rchain32q, rchain64q;  // latency mode: two / four wavefronts per tile (RneaChainProgram::n_waves)
    // statistics for DESIGN.md / bench.py
    double flops_aba = 0, flops_rnea = 0;
};

// Compile a model-description blob (include/grbda_model_desc.h) into a HostPlan.
// Returns 0 or a negative GRBDA_E* code (include/grbda_hip.h); msg receives a diagnostic.
int compile_plan(const void *blob, size_t bytes, const LdsBudget &lds, int sweep_mask, HostPlan &out,
                 char *msg, size_t msg_cap);

// The spanning-tree model (every revolute body its own Revolute cluster) as a model-description blob; span_q / span_v: per body,
// its first position / velocity index in that model (plan.cpp; capi.cpp manifold_derivs).
int make_spanning_blob(const void *blob, size_t bytes, std::vector<unsigned char> &out, std::vector<int32_t> &span_q,
                       std::vector<int32_t> &span_v, char *msg, size_t msg_cap);

}  // namespace grbda_hip
