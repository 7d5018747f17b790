// plan.cpp -- host-side plan compiler: model description blob -> HostPlan (plan.h).
//
// What it does, in reference terms: it freezes everything ClusterTreeModel keeps in its
// object graph (ClusterTreeModel.cpp:10-67: bodies_, cluster_nodes_, position/velocity
// indices; ClusterTreeNode.cpp:6-24: I_, Xup_ parent sub-indices; the LoopConstraint::Static
// G matrix of every explicit cluster joint) into flat tables, assigns a state slot to every
// per-state intermediate, and emits the sweep order the kernels execute.
#include "plan.h"

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../../include/grbda_hip.h"
#include "../../include/grbda_model_desc.h"

namespace grbda_hip {

namespace {

struct Blob {
    const grbda_desc_header *h;
    const grbda_desc_body *bodies;
    const grbda_desc_cluster *clusters;
    const int32_t *ints;
    const double *dbls;
};

int fail(char *msg, size_t cap, int code, const char *fmt, int a = 0, int b = 0)
{
    if (msg && cap) std::snprintf(msg, cap, fmt, a, b);
    return code;
}

int parse(const void *blob, size_t bytes, Blob &m, char *msg, size_t cap)
{
    if (!blob || bytes < sizeof(grbda_desc_header)) return fail(msg, cap, GRBDA_EINVAL, "blob too small");
    const auto *h = static_cast<const grbda_desc_header *>(blob);
    if (h->magic != GRBDA_DESC_MAGIC || h->version != GRBDA_DESC_VERSION)
        return fail(msg, cap, GRBDA_EINVAL, "bad magic/version");
    if (h->n_bodies <= 0 || h->n_clusters <= 0 || h->nq <= 0 || h->nv <= 0 || h->n_ints < 0 || h->n_doubles < 0)
        return fail(msg, cap, GRBDA_EINVAL, "bad header counts");
    size_t off = sizeof(*h);
    m.h = h;
    m.bodies = reinterpret_cast<const grbda_desc_body *>(static_cast<const char *>(blob) + off);
    off += sizeof(grbda_desc_body) * static_cast<size_t>(h->n_bodies);
    m.clusters = reinterpret_cast<const grbda_desc_cluster *>(static_cast<const char *>(blob) + off);
    off += sizeof(grbda_desc_cluster) * static_cast<size_t>(h->n_clusters);
    m.ints = reinterpret_cast<const int32_t *>(static_cast<const char *>(blob) + off);
    off += sizeof(int32_t) * static_cast<size_t>((h->n_ints + 1) & ~1);
    m.dbls = reinterpret_cast<const double *>(static_cast<const char *>(blob) + off);
    off += sizeof(double) * static_cast<size_t>(h->n_doubles);
    if (off > bytes) return fail(msg, cap, GRBDA_EINVAL, "blob truncated");
    return 0;
}

}  // namespace

int compile_plan(const void *blob, size_t bytes, HostPlan &P, char *msg, size_t cap)
{
    Blob m;
    if (int rc = parse(blob, bytes, m, msg, cap)) return rc;
    const int nb = m.h->n_bodies, nc = m.h->n_clusters;
    P = HostPlan();
    P.nq = m.h->nq;
    P.nv = m.h->nv;
    P.n_bodies = nb;
    P.n_clusters = nc;
    P.ori_repr = m.h->ori_repr;
    std::memcpy(P.gravity, m.h->gravity, sizeof P.gravity);
    P.clusters.assign(nc, ClusterRec());
    P.bodies.assign(nb, BodyRec());

    // ---- validation + topology ------------------------------------------------------------
    int q_end = 0, v_end = 0, b_end = 0;
    for (int c = 0; c < nc; c++) {
        const grbda_desc_cluster &cl = m.clusters[c];
        if (cl.first_body != b_end || cl.n_bodies < 1 || cl.first_body + cl.n_bodies > nb)
            return fail(msg, cap, GRBDA_EINVAL, "cluster %d: bodies not contiguous", c);
        if (cl.q_index != q_end || cl.v_index != v_end)
            return fail(msg, cap, GRBDA_EINVAL, "cluster %d: q/v index not cumulative", c);
        if (cl.parent_cluster >= c) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: parent not earlier", c);
        b_end += cl.n_bodies;
        q_end += cl.n_pos;
        v_end += cl.n_vel;
        if (cl.n_bodies > kMaxClusterBodies)
            return fail(msg, cap, GRBDA_EUNSUPPORTED, "cluster %d: %d bodies exceed the kernel limit", c, cl.n_bodies);

        ClusterRec &cr = P.clusters[c];
        cr.first_body = cl.first_body;
        cr.k = cl.n_bodies;
        cr.n = cl.n_vel;
        cr.q_index = cl.q_index;
        cr.v_index = cl.v_index;
        cr.parent_body = -2;
        cr.chained = 0;

        if (cl.constraint_type == GRBDA_CONSTRAINT_FREE) {
            const grbda_desc_body &b = m.bodies[cl.first_body];
            if (cl.n_bodies != 1 || b.joint_type != GRBDA_JOINT_FREE || b.parent != -1 || cl.n_vel != 6)
                return fail(msg, cap, GRBDA_EINVAL, "cluster %d: free joint must be a single root body", c);
            const int npos = m.h->ori_repr == GRBDA_ORI_QUATERNION ? 7 : 6;
            if (cl.n_pos != npos) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: free joint position count", c);
            cr.kind = CK_FREE;
        } else if (cl.constraint_type == GRBDA_CONSTRAINT_STATIC) {
            if (cl.n_vel < 1 || cl.n_vel > kMaxClusterDof)
                return fail(msg, cap, GRBDA_EUNSUPPORTED, "cluster %d: %d DoF exceed the kernel limit", c, cl.n_vel);
            if (cl.n_span_vel != cl.n_bodies || cl.n_span_pos != cl.n_bodies || cl.n_pos != cl.n_vel)
                return fail(msg, cap, GRBDA_EINVAL, "cluster %d: static cluster must have one revolute joint per body", c);
            if (cl.n_dbl < cl.n_span_vel * cl.n_vel || cl.dbl_offset < 0 || cl.dbl_offset + cl.n_dbl > m.h->n_doubles)
                return fail(msg, cap, GRBDA_EINVAL, "cluster %d: G payload missing", c);
            cr.kind = CK_STATIC;
        } else {
            return fail(msg, cap, GRBDA_EUNSUPPORTED,
                        "cluster %d: implicit loop constraint kind %d is not covered by the HIP kernels yet", c,
                        cl.constraint_type);
        }
        for (int i = 0; i < cl.n_bodies; i++) {
            const int gb = cl.first_body + i;
            const grbda_desc_body &b = m.bodies[gb];
            if (b.cluster != c || b.sub_index != i) return fail(msg, cap, GRBDA_EINVAL, "body %d: cluster/sub index", gb);
            if (b.parent >= gb) return fail(msg, cap, GRBDA_EINVAL, "body %d: parent not earlier", gb);
            if (cr.kind == CK_STATIC && (b.joint_type != GRBDA_JOINT_REVOLUTE || b.axis < 0 || b.axis > 2))
                return fail(msg, cap, GRBDA_EINVAL, "body %d: bad joint", gb);
            BodyRec &br = P.bodies[gb];
            br.parent = b.parent;
            br.axis = b.axis;
            br.jtype = b.joint_type;
            const bool in_cluster = b.parent >= cl.first_body;
            br.lam = in_cluster ? b.parent : -1;
            if (in_cluster) {
                cr.chained = 1;
            } else {
                // reference rule (ClusterTreeModel.cpp:112-126): the parent is in the parent cluster
                if (b.parent >= 0 && m.bodies[b.parent].cluster != cl.parent_cluster)
                    return fail(msg, cap, GRBDA_EINVAL, "body %d: parent outside the parent cluster", gb);
                if (b.parent < 0 && cl.parent_cluster >= 0)
                    return fail(msg, cap, GRBDA_EINVAL, "body %d: ground parent in a non-root cluster", gb);
                if (cr.parent_body == -2) cr.parent_body = b.parent;
                else if (cr.parent_body != b.parent)
                    return fail(msg, cap, GRBDA_EUNSUPPORTED,
                                "cluster %d attaches to more than one body of its parent cluster (body %d)", c, gb);
            }
        }
        if (cr.parent_body == -2) return fail(msg, cap, GRBDA_EINVAL, "cluster %d has no root body", c);
    }
    if (q_end != m.h->nq || v_end != m.h->nv || b_end != nb)
        return fail(msg, cap, GRBDA_EINVAL, "header totals do not match clusters");

    for (int b = 0; b < nb; b++)
        if (P.bodies[b].parent >= 0) P.bodies[P.bodies[b].parent].has_child = 1;

    // ---- slots ------------------------------------------------------------------------------
    // Hot, small objects first (they land in LDS): velocities and joint sin/cos, then the
    // accumulators, then the per-cluster solve products.
    int slot = 0;
    auto take = [&slot](int n) { int s = slot; slot += n; return s; };
    for (int b = 0; b < nb; b++) {
        BodyRec &br = P.bodies[b];
        br.slot_sc = take(br.jtype == GRBDA_JOINT_FREE ? 12 : 2);
        br.slot_v = take(6);
    }
    for (int b = 0; b < nb; b++) {
        BodyRec &br = P.bodies[b];
        const ClusterRec &cr = P.clusters[m.bodies[b].cluster];
        br.slot_psi = br.has_child ? take(6) : -1;
        br.slot_a = br.slot_psi;  // psi is dead once the body's cluster finished its backward step
        br.slot_ccl = cr.chained ? take(6) : -1;
    }
    const int ia_region = slot;
    for (int b = 0; b < nb; b++) {
        BodyRec &br = P.bodies[b];
        br.slot_IA = br.has_child ? take(21) : -1;
    }
    for (int c = 0; c < nc; c++) {
        ClusterRec &cr = P.clusters[c];
        cr.slot_K = take(6 * cr.n);
        cr.slot_y0 = take(cr.n);
    }
    P.n_slots = slot;
    // RNEA keeps one body force per body; it never touches the inertia accumulators or the
    // per-cluster solve products, so the forces alias that region.
    P.rnea_slot_f.resize(nb);
    for (int b = 0; b < nb; b++) P.rnea_slot_f[b] = ia_region + 6 * b;
    if (ia_region + 6 * nb > P.n_slots) P.n_slots = ia_region + 6 * nb;

    // parent slots + "first contributor" flags.  Backward sweeps visit clusters in reverse index
    // order and bodies in reverse order inside a cluster, so the first contributor to a body's
    // accumulators is its highest-indexed tree child.
    std::vector<int> last_child(nb, -1);
    for (int b = 0; b < nb; b++)
        if (P.bodies[b].parent >= 0) last_child[P.bodies[b].parent] = b;
    for (int b = 0; b < nb; b++) {
        BodyRec &br = P.bodies[b];
        if (br.parent >= 0) {
            const BodyRec &pr = P.bodies[br.parent];
            br.parent_slot_v = pr.slot_v;
            br.parent_slot_a = pr.slot_a;
            br.parent_slot_IA = pr.slot_IA;
            br.parent_slot_psi = pr.slot_psi;
            br.acc_first = last_child[br.parent] == b;
        } else {
            br.parent_slot_v = br.parent_slot_a = br.parent_slot_IA = br.parent_slot_psi = -1;
            br.acc_first = 0;
        }
    }
    for (int c = 0; c < nc; c++) {
        ClusterRec &cr = P.clusters[c];
        if (cr.parent_body >= 0) {
            const BodyRec &pr = P.bodies[cr.parent_body];
            cr.parent_slot_IA = pr.slot_IA;
            cr.parent_slot_psi = pr.slot_psi;
            cr.parent_slot_a = pr.slot_a;
        } else {
            cr.parent_slot_IA = cr.parent_slot_psi = cr.parent_slot_a = -1;
        }
        cr.acc_first = 0;
    }

    // ---- constants --------------------------------------------------------------------------
    for (int b = 0; b < nb; b++) {
        const grbda_desc_body &bd = m.bodies[b];
        const grbda_desc_cluster &cl = m.clusters[bd.cluster];
        BodyRec &br = P.bodies[b];
        br.cofs = static_cast<int>(P.consts.size());
        for (int i = 0; i < 9; i++) P.consts.push_back(bd.Xtree_E[i]);
        for (int i = 0; i < 3; i++) P.consts.push_back(bd.Xtree_r[i]);
        for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++) {
                const double a = bd.inertia[i * 6 + j], t = bd.inertia[j * 6 + i];
                if (std::fabs(a - t) > 1e-9 * (1.0 + std::fabs(a)))
                    return fail(msg, cap, GRBDA_EINVAL, "body %d: spatial inertia is not symmetric", b);
                P.consts.push_back(0.5 * (a + t));
            }
        if (cl.constraint_type == GRBDA_CONSTRAINT_STATIC) {
            const double *G = m.dbls + cl.dbl_offset;  // n_span_vel x n_vel
            for (int j = 0; j < cl.n_vel; j++) P.consts.push_back(G[bd.sub_index * cl.n_vel + j]);
        }
    }

    // ---- step programs ------------------------------------------------------------------------
    for (int c = 0; c < nc; c++) P.aba_steps.push_back({OP_ABA_FWD, c});
    for (int c = nc - 1; c >= 0; c--) P.aba_steps.push_back({OP_ABA_BWD, c});
    for (int c = 0; c < nc; c++) P.aba_steps.push_back({OP_ABA_ACC, c});
    for (int c = 0; c < nc; c++) P.rnea_steps.push_back({OP_RNEA_FWD, c});
    for (int c = nc - 1; c >= 0; c--) P.rnea_steps.push_back({OP_RNEA_BWD, c});

    // ---- operation count (mul + add, as executed by kernels.hip) --------------------------------
    // per-body costs: E build 12, motion xform 39, force xform 39, sym6*vec 66, force cross 30,
    // congruence 470, plus per-cluster solve terms.  Kept as a model-dependent estimate; the
    // exact figures are listed in DESIGN.md.
    double fa = 0, fr = 0;
    for (int c = 0; c < nc; c++) {
        const ClusterRec &cr = P.clusters[c];
        const int n = cr.n;
        for (int i = 0; i < cr.k; i++) {
            const BodyRec &br = P.bodies[cr.first_body + i];
            const bool fr_ee = br.jtype == GRBDA_JOINT_FREE;
            fa += (fr_ee ? 60 : 40 + 12) + 39 + 2 * n;           // FWD: sincos, E, v, qd
            fa += 12 + 8 + 66 + 30 + 12 + 66 + 6 + 39 * 2;        // BWD: E, c, Iv, cross, psi, IA*c, push h, t
            fa += br.parent >= 0 ? 470 + 39 : 0;                  // congruence + bias to parent
            fa += 6 * n * 2 + n * n * 2;                          // F, D accumulation
            fa += 12 + 39 + 8 + 2 * n + 6;                        // ACC: E, a, c, qdd
            fr += (fr_ee ? 60 : 52) + 39 * 2 + 8 + 2 * n * 2 + 66 * 2 + 30 + 39 + 2 * n;
        }
        fa += n * n * n / 3.0 + 2.0 * n * n * 7 + 21 * 2 * n + 12 * n + 12 * n;  // solve, K, IA -= F K, psi += F y0, ydd
    }
    P.flops_aba = fa;
    P.flops_rnea = fr;
    return 0;
}

}  // namespace grbda_hip
