// plan.cpp -- host-side plan compiler: model description blob -> HostPlan (plan.h).
//
// What it does, in reference terms: it freezes everything ClusterTreeModel keeps in its
// object graph (ClusterTreeModel.cpp:10-67: bodies_, cluster_nodes_, position/velocity
// indices; ClusterTreeNode.cpp:6-24: I_, Xup_ parent sub-indices; the LoopConstraint::Static
// G matrix of every explicit cluster joint) into flat tables, assigns a state slot to every
// per-state intermediate, and emits the sweep order the kernels execute.
#include "plan.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#include "../../include/grbda_hip.h"
#include "../../include/grbda_model_desc.h"

namespace grbda_hip {
// Is the tree rotation Et (row-major at consts[cofs]) a cyclic permutation of the axes, (Et x)_i = x_((i + k) % 3), to rounding (1e-14)?
// Returns the shift k, or -1.  The chain kernels then use the permutation-structured transforms (devmath.h, rzp_*).
// GRBDA_NO_PERM_LINKS: A/B switch, every link on the general rotation path.
static int cyclic_shift(const std::vector<double> &consts, int cofs)
{
    if (std::getenv("GRBDA_NO_PERM_LINKS")) return -1;
    for (int k = 0; k < 3; k++) {
        bool is = true;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) is = is && std::fabs(consts[cofs + 3 * i + j] - (j == (i + k) % 3 ? 1.0 : 0.0)) < 1e-14;
        if (is) return k;
    }
    return -1;
}


namespace {

struct Blob {
    const grbda_desc_header *h;
    const grbda_desc_body *bodies;
    const grbda_desc_cluster *clusters;
    const int32_t *ints;
    const double *dbls;
};

int fail(char *msg, size_t cap, int code, const char *fmt, int a = 0, int b = 0, int c = 0)
{
    if (msg && cap) std::snprintf(msg, cap, fmt, a, b, c);
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

int compile_plan(const void *blob, size_t bytes, const LdsBudget &lds, int sweep_mask, HostPlan &P, char *msg,
                 size_t cap)
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
    std::vector<ClusterRec> clusters(nc, ClusterRec());
    std::vector<BodyRec> bodies(nb, BodyRec());

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
        // clusters beyond the structured kernels' limits (kMaxClusterBodies, kMaxClusterDof) go through the spanning tree:
        // HostPlan::big_clusters, manifold_kernels.hip's wide variants
        const bool no_big = std::getenv("GRBDA_NO_PROJECTION") != nullptr;
        if (cl.n_bodies > (no_big ? kMaxClusterBodies : kBigClusterBodies))
            return fail(msg, cap, GRBDA_EUNSUPPORTED, "cluster %d: %d bodies exceed the kernel limit", c, cl.n_bodies);
        const int max_dof = no_big ? kMaxClusterDof : kBigClusterDof;
        if (cl.constraint_type != GRBDA_CONSTRAINT_FREE && (cl.n_bodies > kMaxClusterBodies || cl.n_vel > kMaxClusterDof)) {
            P.projection_only = true;
            P.big_clusters = true;
        }

        ClusterRec &cr = clusters[c];
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
            if (cl.n_vel < 1 || cl.n_vel > max_dof)
                return fail(msg, cap, GRBDA_EUNSUPPORTED, "cluster %d: %d DoF exceed the kernel limit", c, cl.n_vel);
            if (cl.n_span_vel != cl.n_bodies || cl.n_span_pos != cl.n_bodies || cl.n_pos != cl.n_vel)
                return fail(msg, cap, GRBDA_EINVAL, "cluster %d: static cluster must have one revolute joint per body", c);
            if (cl.n_dbl < cl.n_span_vel * cl.n_vel || cl.dbl_offset < 0 || cl.dbl_offset + cl.n_dbl > m.h->n_doubles)
                return fail(msg, cap, GRBDA_EINVAL, "cluster %d: G payload missing", c);
            cr.kind = CK_STATIC;
        } else if (cl.constraint_type == GRBDA_CONSTRAINT_LOOP_POSITION || cl.constraint_type == GRBDA_CONSTRAINT_TRIG_POLY) {
            if (cl.n_vel < 1 || cl.n_vel > max_dof)
                return fail(msg, cap, GRBDA_EUNSUPPORTED, "cluster %d: %d DoF exceed the kernel limit", c, cl.n_vel);
            if (cl.n_span_vel != cl.n_bodies || cl.n_span_pos != cl.n_bodies || cl.n_pos != cl.n_bodies)
                return fail(msg, cap, GRBDA_EINVAL, "cluster %d: loop cluster must have one revolute joint per body", c);
            // 1 - 3 rows: the structured kernels (closed-form inverses of K_d).  4 - kMaxConstraintRows rows of URDF+ position loops -- two planar
            // loops that share bodies, a spatial loop beside a planar one; the reference inverts a K_d of any size, GenericJoint.cpp:57-90 --:
            // the spanning-tree route with the wide kernels (K_d by elimination), like clusters beyond 8 bodies / 4 coordinates
            const int max_rows = (cl.constraint_type == GRBDA_CONSTRAINT_LOOP_POSITION && !no_big) ? kMaxConstraintRows : 3;
            if (cl.n_constraint_rows < 1 || cl.n_constraint_rows > max_rows || cl.n_constraint_rows != cl.n_bodies - cl.n_vel)
                return fail(msg, cap, GRBDA_EUNSUPPORTED, "cluster %d: %d constraint rows (1..%d supported)", c, cl.n_constraint_rows, max_rows);
            if (cl.n_constraint_rows > 3) {
                P.projection_only = true;
                P.big_clusters = true;
            }
            if (cl.int_offset < 0 || cl.int_offset + cl.n_int > m.h->n_ints || cl.n_int < cl.n_bodies)
                return fail(msg, cap, GRBDA_EINVAL, "cluster %d: loop payload missing", c);
            cr.kind = CK_LOOP;
            cr.rows = cl.n_constraint_rows;
            cr.cons_type = cl.constraint_type == GRBDA_CONSTRAINT_TRIG_POLY ? 1 : 0;
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
            if (cr.kind != CK_FREE && (b.joint_type != GRBDA_JOINT_REVOLUTE || b.axis < 0 || b.axis > 2))
                return fail(msg, cap, GRBDA_EINVAL, "body %d: bad joint", gb);
            BodyRec &br = bodies[gb];
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
                else if (cr.parent_body != b.parent) {
                    // several parent bodies: the structured recursions do not apply (plan.h, HostPlan::projection_only)
                    if (std::getenv("GRBDA_NO_PROJECTION"))
                        return fail(msg, cap, GRBDA_EUNSUPPORTED,
                                    "cluster %d attaches to more than one body of its parent cluster (body %d)", c, gb);
                    P.projection_only = true;
                }
            }
        }
        if (cr.parent_body == -2) return fail(msg, cap, GRBDA_EINVAL, "cluster %d has no root body", c);
    }
    if (q_end != m.h->nq || v_end != m.h->nv || b_end != nb)
        return fail(msg, cap, GRBDA_EINVAL, "header totals do not match clusters");

    for (int b = 0; b < nb; b++)
        if (bodies[b].parent >= 0) bodies[bodies[b].parent].has_child = 1;
    for (int b = 0; b < nb; b++)
        if (bodies[b].has_child && m.bodies[b].sub_index < 31) clusters[m.bodies[b].cluster].child_mask |= 1 << m.bodies[b].sub_index;

    // ---- canonical joint axes -------------------------------------------------------------------
    // The model is re-expressed so that every revolute joint turns about the z axis of its body frame
    // (bodies of URDF+ position-loop clusters included since round 4: their loop origins -- points of the link frames -- and the axis
    // masks of their constraint rows -- axes of the nearest common ancestor -- are rotated with the frames where the loop
    // payload is emitted below): body i's coordinates are rotated by the cyclic permutation Rc_i that maps its
    // joint axis onto z (v_new = Rc_i v_old).  Then R_a(q) E_tree becomes R_z(q) (Rc_i E_tree Rc_p^T),
    // the tree offset r (parent coordinates) becomes Rc_p r and the spatial inertia D I D^T with
    // D = blockdiag(Rc_i, Rc_i).  Joint coordinates, torques and accelerations are unchanged, and
    // world-frame external forces reach the bodies through the (equally rotated) absolute transforms.
    // The fast kernels and the straight-line handlers rely on it (kernels.hip, load_body): the joint axis
    // is a constant there.
    std::vector<grbda_desc_body> canon;
    std::vector<std::array<double, 9>> Rc(nb);
    std::vector<int> canon_shift(nb, 0);  // v_new[i] = v_old[(i + shift) % 3]
    {
        canon.assign(m.bodies, m.bodies + nb);
        auto perm = [](int axis, double R[9]) {
            for (int i = 0; i < 9; i++) R[i] = 0;
            if (axis == 0) { R[0 * 3 + 1] = 1; R[1 * 3 + 2] = 1; R[2 * 3 + 0] = 1; }        // x -> z
            else if (axis == 1) { R[0 * 3 + 2] = 1; R[1 * 3 + 0] = 1; R[2 * 3 + 1] = 1; }   // y -> z
            else { R[0] = R[4] = R[8] = 1; }
        };
        for (int b = 0; b < nb; b++) {
            const bool keep = m.bodies[b].joint_type != GRBDA_JOINT_REVOLUTE;
            perm(keep ? 2 : m.bodies[b].axis, Rc[b].data());
            canon_shift[b] = keep ? 0 : (m.bodies[b].axis == 0 ? 1 : (m.bodies[b].axis == 1 ? 2 : 0));
        }
        for (int b = 0; b < nb; b++) {
            grbda_desc_body &bd = canon[b];
            const double *Ri = Rc[b].data();
            double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (bd.parent >= 0) std::memcpy(Rp, Rc[bd.parent].data(), sizeof Rp);
            double E[9], t[9], r[3];
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {  // t = Ri * E_tree
                    double sacc = 0;
                    for (int k2 = 0; k2 < 3; k2++) sacc += Ri[i * 3 + k2] * bd.Xtree_E[k2 * 3 + j];
                    t[i * 3 + j] = sacc;
                }
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {  // E = t * Rp^T
                    double sacc = 0;
                    for (int k2 = 0; k2 < 3; k2++) sacc += t[i * 3 + k2] * Rp[j * 3 + k2];
                    E[i * 3 + j] = sacc;
                }
            for (int i = 0; i < 3; i++) r[i] = Rp[i * 3] * bd.Xtree_r[0] + Rp[i * 3 + 1] * bd.Xtree_r[1] + Rp[i * 3 + 2] * bd.Xtree_r[2];
            std::memcpy(bd.Xtree_E, E, sizeof E);
            std::memcpy(bd.Xtree_r, r, sizeof r);
            double I[36];
            for (int bi = 0; bi < 2; bi++)
                for (int bj = 0; bj < 2; bj++)
                    for (int i = 0; i < 3; i++)
                        for (int j = 0; j < 3; j++) {
                            double sacc = 0;  // Ri * block * Ri^T
                            for (int k2 = 0; k2 < 3; k2++)
                                for (int l2 = 0; l2 < 3; l2++)
                                    sacc += Ri[i * 3 + k2] * bd.inertia[(3 * bi + k2) * 6 + 3 * bj + l2] * Ri[j * 3 + l2];
                            I[(3 * bi + i) * 6 + 3 * bj + j] = sacc;
                        }
            std::memcpy(bd.inertia, I, sizeof I);
            bodies[b].canon_axis = 2;
            if (bd.joint_type == GRBDA_JOINT_REVOLUTE) {
                bodies[b].canon_axis = bd.axis;
                bd.axis = 2;
                bodies[b].axis = 2;
            }
        }
        m.bodies = canon.data();
    }

    // ---- constants --------------------------------------------------------------------------
    for (int b = 0; b < nb; b++) {
        const grbda_desc_body &bd = m.bodies[b];
        const grbda_desc_cluster &cl = m.clusters[bd.cluster];
        BodyRec &br = bodies[b];
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

    // ---- explicit clusters: K and a left inverse of G for spanning-state input (kernels.hip, state_kernel) -------
    // ClusterJoints::Base::toSpanningTreeState (ClusterJoint.cpp:22-71) accepts spanning positions / velocities; the
    // engine's coordinates are the independent ones, y = G+ q_span.  consts[dofs]: K (rows x k) then G+ (n x k).  G+ picks
    // the rows of G that are unit vectors (every cluster joint of the reference has them: link coordinates), and is the
    // least-squares inverse (G^T G)^-1 G^T otherwise.  A description without K gets an orthonormal basis of null(G^T).
    for (int c = 0; c < nc; c++) {
        const grbda_desc_cluster &cl = m.clusters[c];
        if (clusters[c].kind != CK_STATIC) continue;
        const int k = cl.n_bodies, n = cl.n_vel, rows = k - n;
        const double *G = m.dbls + cl.dbl_offset;
        clusters[c].rows = rows;
        clusters[c].dofs = static_cast<int>(P.consts.size());
        std::vector<double> Gp(static_cast<size_t>(n) * k, 0.0);
        bool picked = true;
        for (int a = 0; a < n && picked; a++) {
            int row = -1;
            for (int i = 0; i < k && row < 0; i++) {
                bool unit = true;
                for (int b2 = 0; b2 < n; b2++) unit = unit && G[i * n + b2] == (b2 == a ? 1.0 : 0.0);
                if (unit) row = i;
            }
            if (row < 0) picked = false;
            else Gp[static_cast<size_t>(a) * k + row] = 1.0;
        }
        if (!picked) {  // (G^T G)^-1 G^T by Gauss-Jordan with partial pivoting, n <= 4
            std::vector<double> A(static_cast<size_t>(n) * (n + k), 0.0);
            for (int a = 0; a < n; a++) {
                for (int b2 = 0; b2 < n; b2++)
                    for (int i = 0; i < k; i++) A[a * (n + k) + b2] += G[i * n + a] * G[i * n + b2];
                for (int i = 0; i < k; i++) A[a * (n + k) + n + i] = G[i * n + a];
            }
            for (int col = 0; col < n; col++) {
                int piv = col;
                for (int r = col + 1; r < n; r++)
                    if (std::fabs(A[r * (n + k) + col]) > std::fabs(A[piv * (n + k) + col])) piv = r;
                if (std::fabs(A[piv * (n + k) + col]) < 1e-300) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: G is rank deficient", c);
                for (int j = 0; j < n + k; j++) std::swap(A[col * (n + k) + j], A[piv * (n + k) + j]);
                const double d = 1.0 / A[col * (n + k) + col];
                for (int j = 0; j < n + k; j++) A[col * (n + k) + j] *= d;
                for (int r = 0; r < n; r++) {
                    if (r == col) continue;
                    const double f = A[r * (n + k) + col];
                    for (int j = 0; j < n + k; j++) A[r * (n + k) + j] -= f * A[col * (n + k) + j];
                }
            }
            for (int a = 0; a < n; a++)
                for (int i = 0; i < k; i++) Gp[static_cast<size_t>(a) * k + i] = A[a * (n + k) + n + i];
        }
        if (cl.n_dbl >= k * n + rows * k && cl.n_constraint_rows == rows) {
            const double *K = G + k * n;
            for (int i = 0; i < rows * k; i++) P.consts.push_back(K[i]);
        } else {  // orthonormal complement of range(G): Gram-Schmidt over the unit vectors
            std::vector<std::vector<double>> basis;  // orthonormal
            auto orthonormalise = [&](std::vector<double> &v) {
                for (int pass = 0; pass < 2; pass++)
                    for (const auto &bv : basis) {
                        double dot = 0;
                        for (int i = 0; i < k; i++) dot += v[i] * bv[i];
                        for (int i = 0; i < k; i++) v[i] -= dot * bv[i];
                    }
                double nn = 0;
                for (int i = 0; i < k; i++) nn += v[i] * v[i];
                if (nn < 1e-20) return false;
                for (int i = 0; i < k; i++) v[i] /= std::sqrt(nn);
                return true;
            };
            for (int a = 0; a < n; a++) {
                std::vector<double> v(k);
                for (int i = 0; i < k; i++) v[i] = G[i * n + a];
                if (orthonormalise(v)) basis.push_back(v);
            }
            int have = 0;
            for (int e = 0; e < k; e++) {
                std::vector<double> v(k, 0.0);
                v[e] = 1.0;
                if (!orthonormalise(v)) continue;
                basis.push_back(v);
                if (have < rows) for (int i = 0; i < k; i++) P.consts.push_back(v[i]);
                have++;
            }
            if (have != rows) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: G is rank deficient", c);
        }
        P.consts.insert(P.consts.end(), Gp.begin(), Gp.end());
    }

    // ---- axisymmetric leaf bodies (rotors) ------------------------------------------------------------
    // If I is invariant under rotation about the joint axis (COM on the axis, equal transverse
    // inertias) then X(q)^T I X(q), X(q)^T I s and X(q)^T (v x* I v) are independent of q: the body is
    // evaluated at q = 0, and its inertia contribution to the parent is the constant X0^T I X0.
    for (int b = 0; b < nb; b++) bodies[b].xofs = -1;
    {
        std::vector<std::vector<double>> extra(nb);
        auto at = [](const double *M, int i, int j) { return M[i * 6 + j]; };
        for (int b = 0; b < nb; b++) {
            const grbda_desc_body &bd = m.bodies[b];
            if (bodies[b].has_child || bd.joint_type != GRBDA_JOINT_REVOLUTE) continue;
            // (bodies of URDF+ position-loop clusters are always evaluated at their own angle: they carry constraint points)
            if (clusters[bd.cluster].kind == CK_LOOP && clusters[bd.cluster].cons_type == 0) continue;
            bool inv = true;
            double scale = 0;
            for (int i = 0; i < 36; i++) scale = std::max(scale, std::fabs(bd.inertia[i]));
            for (double th : {0.7, 1.9}) {
                // spatial rotation about the coordinate axis through the origin: blockdiag(R, R)
                const double sn = std::sin(th), cs = std::cos(th);
                double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                const int a1 = (bd.axis + 1) % 3, a2 = (bd.axis + 2) % 3;
                R[a1 * 3 + a1] = cs; R[a1 * 3 + a2] = sn; R[a2 * 3 + a1] = -sn; R[a2 * 3 + a2] = cs;
                for (int bi = 0; bi < 2 && inv; bi++)
                    for (int bj = 0; bj < 2 && inv; bj++)
                        for (int i = 0; i < 3 && inv; i++)
                            for (int j = 0; j < 3; j++) {
                                double sacc = 0;  // (R^T M R)_ij of block (bi, bj)
                                for (int k2 = 0; k2 < 3; k2++)
                                    for (int l2 = 0; l2 < 3; l2++)
                                        sacc += R[k2 * 3 + i] * at(bd.inertia, 3 * bi + k2, 3 * bj + l2) * R[l2 * 3 + j];
                                if (std::fabs(sacc - at(bd.inertia, 3 * bi + i, 3 * bj + j)) > 1e-12 * (scale + 1e-300)) { inv = false; break; }
                            }
            }
            if (!inv) continue;
            bodies[b].axisym = 1;
            // X0 = (E_tree, r_tree): 6x6 motion transform [[E,0],[-E r^,E]]; B = X0^T I X0
            const double *E = bd.Xtree_E, *r = bd.Xtree_r;
            const double rh[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
            double X[36] = {0};
            for (int i = 0; i < 3; i++)
                for (int j = 0; j < 3; j++) {
                    X[i * 6 + j] = E[i * 3 + j];
                    X[(i + 3) * 6 + j + 3] = E[i * 3 + j];
                    double er = 0;
                    for (int k2 = 0; k2 < 3; k2++) er += E[i * 3 + k2] * rh[k2 * 3 + j];
                    X[(i + 3) * 6 + j] = -er;
                }
            if (bd.parent < 0) continue;  // a rotor on the ground: nobody to hand X0^T I X0 to
            std::vector<double> &ex = extra[bd.parent];
            if (ex.empty()) ex.assign(21, 0.0);
            int idx = 0;
            for (int i = 0; i < 6; i++)
                for (int j = i; j < 6; j++, idx++) {
                    double sacc = 0;
                    for (int k2 = 0; k2 < 6; k2++)
                        for (int l2 = 0; l2 < 6; l2++) sacc += X[k2 * 6 + i] * at(bd.inertia, k2, l2) * X[l2 * 6 + j];
                    ex[idx] += sacc;
                }
        }
        for (int b = 0; b < nb; b++)
            if (!extra[b].empty()) {
                // stored as I_b + sum_children X0^T I X0: the backward step starts its accumulation from it
                bodies[b].xofs = static_cast<int>(P.consts.size());
                for (int j = 0; j < 21; j++) P.consts.push_back(P.consts[bodies[b].cofs + 12 + j] + extra[b][j]);
            }
    }

    // ---- cluster shapes with a straight-line handler (plan.h, ClusterShape) ---------------------------
    for (int c = 0; c < nc; c++) {
        ClusterRec &cr = clusters[c];
        cr.shape = SHAPE_GENERIC;
        cr.link_body = cr.rotor_body = -1;
        if (cr.kind != CK_STATIC || cr.n != 1 || cr.chained || cr.parent_body < 0) continue;
        const int f = cr.first_body;
        if (cr.k == 1 && !bodies[f].axisym) {
            cr.shape = SHAPE_REV;
            cr.link_body = f;
        } else if (cr.k == 2) {
            for (int rot = 0; rot < 2; rot++) {
                const int r = f + rot, l = f + 1 - rot;
                if (bodies[r].axisym && !bodies[r].has_child && !bodies[l].axisym) {
                    cr.shape = SHAPE_REV_ROTOR;
                    cr.link_body = l;
                    cr.rotor_body = r;
                    break;
                }
            }
        }
    }

    // ---- implicit-loop payload -------------------------------------------------------------------
    std::vector<int> trig_args(nc, 0);  // distinct factor arguments of a trig-polynomial constraint
    for (int c = 0; c < nc; c++) {
        const grbda_desc_cluster &cl = m.clusters[c];
        if (clusters[c].kind != CK_LOOP) continue;
        if (cl.dbl_offset < 0 || cl.n_dbl < 0 || cl.dbl_offset + cl.n_dbl > m.h->n_doubles)
            return fail(msg, cap, GRBDA_EINVAL, "cluster %d: constraint doubles outside the blob", c);
        const int32_t *ip = m.ints + cl.int_offset;
        const double *dp = m.dbls + cl.dbl_offset;
        const int k = cl.n_bodies;
        const bool trig = clusters[c].cons_type == 1;
        const int32_t *flags = trig ? ip : ip + 1;
        clusters[c].iofs = static_cast<int>(P.cints.size());
        clusters[c].dofs = static_cast<int>(P.consts.size());
        P.cints.push_back(trig ? cl.n_constraint_rows : ip[0]);
        std::vector<int> ind, dep;
        for (int i = 0; i < k; i++) (flags[i] ? ind : dep).push_back(i);
        if (static_cast<int>(ind.size()) != cl.n_vel || static_cast<int>(dep.size()) != cl.n_constraint_rows)
            return fail(msg, cap, GRBDA_EINVAL, "cluster %d: independent/dependent coordinate counts", c);
        P.cints.push_back(static_cast<int>(ind.size()));
        for (int i : ind) P.cints.push_back(i);
        P.cints.push_back(static_cast<int>(dep.size()));
        for (int i : dep) P.cints.push_back(i);
        if (trig) {
            // blob: per row n_terms; per term n_factors, type[n_factors] | doubles: coef, per factor w[k], b.
            // plan: the DISTINCT factor arguments a = w.q + b first (the kernel evaluates each sin/cos once:
            // the Tello hip differential has 4 distinct angles in 30 factors) --
            //   ints:    n_args, per row: n_terms, per term: n_factors, (type, argument index)[n_factors]
            //   doubles: per argument w[k], b; then per term coef
            const int32_t *tp = ip + k;
            const int32_t *tend = ip + cl.n_int;
            std::vector<std::vector<double>> args;
            std::vector<int32_t> prog;
            std::vector<double> coefs;
            int nd = 0;
            for (int r = 0; r < cl.n_constraint_rows; r++) {
                if (tp >= tend) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: trig-poly payload truncated", c);
                const int nt = *tp++;
                prog.push_back(nt);
                if (nt < 0 || nt > cl.n_int) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: trig-poly term count", c);
                for (int t = 0; t < nt; t++) {
                    if (tp >= tend) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: trig-poly payload truncated", c);
                    const int nf = *tp++;
                    if (nf < 0 || nf > 4) return fail(msg, cap, GRBDA_EUNSUPPORTED, "cluster %d: a term has %d factors (max 4)", c, nf);
                    if (tp + nf > tend) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: trig-poly payload truncated", c);
                    if (nd + 1 + nf * (k + 1) > cl.n_dbl) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: trig-poly doubles truncated", c);
                    prog.push_back(nf);
                    coefs.push_back(dp[nd++]);
                    for (int f = 0; f < nf; f++) {
                        const std::vector<double> a(dp + nd, dp + nd + k + 1);
                        nd += k + 1;
                        int idx = -1;
                        for (size_t i = 0; i < args.size(); i++)
                            if (args[i] == a) idx = static_cast<int>(i);
                        if (idx < 0) {
                            idx = static_cast<int>(args.size());
                            args.push_back(a);
                        }
                        if (*tp < 0 || *tp > 2) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: trig-poly factor type %d", c, *tp);
                        prog.push_back(*tp++);
                        prog.push_back(idx);
                    }
                }
            }
            trig_args[c] = static_cast<int>(args.size());
            P.cints.push_back(trig_args[c]);
            P.cints.insert(P.cints.end(), prog.begin(), prog.end());
            for (const auto &a : args) P.consts.insert(P.consts.end(), a.begin(), a.end());
            P.consts.insert(P.consts.end(), coefs.begin(), coefs.end());
        } else {
            const int n_loops = ip[0];
            const int32_t *lp = ip + 1 + k;
            const int32_t *lend = ip + cl.n_int;
            int rows = 0;
            if (n_loops < 1 || n_loops > 3 || cl.n_int < 1 + k || cl.n_dbl < 24 * n_loops)
                return fail(msg, cap, GRBDA_EINVAL, "cluster %d: loop payload truncated", c);
            for (int l = 0; l < n_loops; l++) {
                // (every index is checked against the end of the cluster's integer payload before it is used)
                if (lp + 1 > lend) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: loop payload truncated", c);
                const int np = lp[0];
                if (np < 0 || np > k || lp + 2 + np > lend) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: bad loop chain", c);
                const int ns = lp[1 + np];
                if (ns < 0 || ns > k || lp + 3 + np + ns > lend) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: bad loop chain", c);
                const int mask = lp[2 + np + ns];
                for (int t = 0; t < np + ns; t++) {
                    const int sub = lp[t < np ? 1 + t : 2 + t];
                    if (sub < 0 || sub >= k) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: loop chain names body %d", c, sub);
                }
                // canonical joint axes: the constraint rows are components along the axes of the nearest common ancestor (the tree
                // parent of the first joint of either sub-chain, or the ground), the origins are points of the predecessor / successor
                // link (the last body of each sub-chain, or the ancestor itself when the sub-chain is empty)
                const int fb = cl.first_body;
                const int nca = np > 0 ? canon[fb + lp[1]].parent : (ns > 0 ? canon[fb + lp[2 + np]].parent : -1);
                const int sh = nca >= 0 ? canon_shift[nca] : 0;
                int mask2 = 0;
                for (int i = 0; i < 3; i++)
                    if ((mask >> ((i + sh) % 3)) & 1) mask2 |= 1 << i;
                for (int t = 0; t < 2 + np + ns; t++) P.cints.push_back(lp[t]);
                P.cints.push_back(mask2);
                const int link[2] = {np > 0 ? fb + lp[np] : nca, ns > 0 ? fb + lp[1 + np + ns] : nca};
                lp += 3 + np + ns;
                for (int a = 0; a < 3; a++) rows += (mask >> a) & 1;
                for (int side = 0; side < 2; side++) {
                    const double *o = dp + 24 * l + 12 * side;
                    double Ri[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                    if (link[side] >= 0) std::memcpy(Ri, Rc[link[side]].data(), sizeof Ri);
                    for (int i = 0; i < 3; i++)
                        for (int j = 0; j < 3; j++) {  // E_new = E Ri^T
                            double acc2 = 0;
                            for (int k2 = 0; k2 < 3; k2++) acc2 += o[3 * i + k2] * Ri[3 * j + k2];
                            P.consts.push_back(acc2);
                        }
                    for (int i = 0; i < 3; i++) P.consts.push_back(Ri[3 * i] * o[9] + Ri[3 * i + 1] * o[10] + Ri[3 * i + 2] * o[11]);
                }
            }
            if (rows != cl.n_constraint_rows) return fail(msg, cap, GRBDA_EINVAL, "cluster %d: loop rows mismatch", c);
        }
    }

    if (P.projection_only) {
        // tables for the spanning-tree route only (manifold_kernels.hip reads clusters / bodies / consts / cints); the coordinates two
        // clusters share a root path on -- the entries of H and of the derivative matrices that are not structural zeros
        for (Layout *L : {&P.lay32, &P.lay64, &P.lay32x, &P.lay64x, &P.lay32s}) {
            L->clusters = clusters;
            L->rnea_clusters = clusters;
            L->bodies = bodies;
            L->rnea_bodies = bodies;
            L->acc_k.assign(1, -1);
        }
        if (P.nv <= 64) {
            P.deriv.related.assign(P.nv, 0);
            for (int c = 0; c < nc; c++)
                for (int a = c; a >= 0; a = m.clusters[a].parent_cluster)
                    for (int i = 0; i < m.clusters[c].n_vel; i++)
                        for (int j = 0; j < m.clusters[a].n_vel; j++) {
                            P.deriv.related[m.clusters[c].v_index + i] |= uint64_t(1) << (m.clusters[a].v_index + j);
                            P.deriv.related[m.clusters[a].v_index + j] |= uint64_t(1) << (m.clusters[c].v_index + i);
                        }
        }
        return 0;
    }

    // ---- sweep schedule: depth-first, a subtree is swept forward then backward ------------------
    std::vector<std::vector<int>> kids(nc);
    std::vector<int> roots;
    for (int c = 0; c < nc; c++) {
        if (m.clusters[c].parent_cluster >= 0) kids[m.clusters[c].parent_cluster].push_back(c);
        else roots.push_back(c);
    }
    std::vector<int> tF(nc, -1), tB(nc, -1), tA(nc, -1), tRF(nc, -1), tRB(nc, -1);
    {
        std::vector<std::pair<int, int>> stack;  // (cluster, next child)
        for (int r : roots) {
            stack.push_back({r, 0});
            tF[r] = static_cast<int>(P.aba_steps.size());
            P.aba_steps.push_back({OP_ABA_FWD, r, {0, 0}});
            tRF[r] = static_cast<int>(P.rnea_steps.size());
            P.rnea_steps.push_back({OP_RNEA_FWD, r, {0, 0}});
            while (!stack.empty()) {
                auto &top = stack.back();
                const int c = top.first;
                if (top.second < static_cast<int>(kids[c].size())) {
                    const int ch = kids[c][top.second++];
                    tF[ch] = static_cast<int>(P.aba_steps.size());
                    P.aba_steps.push_back({OP_ABA_FWD, ch, {0, 0}});
                    tRF[ch] = static_cast<int>(P.rnea_steps.size());
                    P.rnea_steps.push_back({OP_RNEA_FWD, ch, {0, 0}});
                    stack.push_back({ch, 0});
                } else {
                    tB[c] = static_cast<int>(P.aba_steps.size());
                    P.aba_steps.push_back({OP_ABA_BWD, c, {0, 0}});
                    tRB[c] = static_cast<int>(P.rnea_steps.size());
                    P.rnea_steps.push_back({OP_RNEA_BWD, c, {0, 0}});
                    stack.pop_back();
                }
            }
        }
        // acceleration sweep, depth-first pre-order
        for (int r : roots) {
            std::vector<int> st{r};
            while (!st.empty()) {
                const int c = st.back();
                st.pop_back();
                tA[c] = static_cast<int>(P.aba_steps.size());
                P.aba_steps.push_back({OP_ABA_ACC, c, {0, 0}});
                for (int i = static_cast<int>(kids[c].size()) - 1; i >= 0; i--) st.push_back(kids[c][i]);
            }
        }
    }

    // steps in which the straight-line handlers have nothing to do (kOpSkipFast, plan.h)
    for (Step &st : P.aba_steps)
        if (st.op == OP_ABA_FWD && clusters[st.cluster].shape && !clusters[st.cluster].child_mask) st.op |= kOpSkipFast;
    for (Step &st : P.rnea_steps)
        if (st.op == OP_RNEA_BWD && clusters[st.cluster].shape && !clusters[st.cluster].child_mask) st.op |= kOpSkipFast;

    // ---- register hand-over along chains ------------------------------------------------------------
    // When every tree child of body p lies in ONE cluster c and the backward step of p's cluster
    // directly follows the backward step of c, the projected inertia / bias of c never touches a
    // slot: it stays in registers across the two steps (carry_out on c, carry_in on p).
    for (int c = 0; c < nc; c++) {
        const int p = clusters[c].parent_body;
        if (p < 0) continue;
        bool only = true;
        for (int j = 0; j < nb; j++)
            if (bodies[j].parent == p && m.bodies[j].cluster != c) only = false;
        const int next = tB[c] + 1;
        if (only && next < static_cast<int>(P.aba_steps.size()) && (P.aba_steps[next].op & kOpMask) == OP_ABA_BWD &&
            P.aba_steps[next].cluster == m.bodies[p].cluster) {
            clusters[c].carry_out = 1;
            bodies[p].carry_in = 1;
        }
    }

    for (int c = 0; c < nc; c++) {
        ClusterRec &cr = clusters[c];
        cr.hot = cr.shape != SHAPE_GENERIC && cr.carry_out && bodies[cr.link_body].has_child && bodies[cr.link_body].carry_in;
    }

    if (sweep_mask != 7) {  // profiling aid: drop whole sweeps (results are then meaningless)
        std::vector<Step> kept;
        for (const Step &st : P.aba_steps)
            if (((st.op & kOpMask) == OP_ABA_FWD && (sweep_mask & 1)) || (st.op == OP_ABA_BWD && (sweep_mask & 2)) ||
                (st.op == OP_ABA_ACC && (sweep_mask & 4)))
                kept.push_back(st);
        P.aba_steps = kept;
    }

    // ---- live ranges + interval allocation --------------------------------------------------------
    struct Obj {
        int *field;  // where the slot number goes (index into a flat array of fields)
        int size, prio, birth, death, slot;
        int force = 0;  // split layouts: 1 must live in LDS, 2 must live in the global slab
        int tag = 0;    // chain programs: 1 = work space that never leaves LDS whatever its size
        // latency-mode chain programs: the wavefront whose limb the object belongs to (-1: the base's, ordered against
        // everything by the barriers) and the phase it lives in (0: forward / backward runs, 1: acceleration runs).  Limbs of
        // different wavefronts run concurrently inside a phase: their objects never share slots, whatever the segment order says.
        int owner = -1, phase = 0;
        int span = 0;  // 1: alive in BOTH phases (the [K | y0] blocks of a limb: written by its backward run, read by its acceleration run)
    };
    // returns false when an object that must live in LDS does not fit the budget
    // mode 0: placement order (priority, birth); 1: longest-lived first; 2: largest first (allocate_packed below)
    auto allocate = [](std::vector<Obj> &objs, int lds_budget, int &n_lds, int &n_glb, int mode = 0) -> bool {
        bool ok = true;
        const int lds_base = 0;
        std::vector<int> order(objs.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = static_cast<int>(i);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            // (prio >= 10: extras that take what LDS the others leave, in every order -- the [K | y0] blocks of latency-mode programs)
            const bool xa = objs[a].prio >= 10, xb = objs[b].prio >= 10;
            if (xa != xb) return xb;
            if (xa && objs[a].prio != objs[b].prio) return objs[a].prio < objs[b].prio;
            if (mode == 1) {
                const int la = objs[a].death - objs[a].birth, lb = objs[b].death - objs[b].birth;
                if (la != lb) return la > lb;
                return objs[a].size > objs[b].size;
            }
            if (mode == 2) {
                if (objs[a].size != objs[b].size) return objs[a].size > objs[b].size;
                return objs[a].birth < objs[b].birth;
            }
            if (objs[a].prio != objs[b].prio) return objs[a].prio < objs[b].prio;
            return objs[a].birth < objs[b].birth;
        });
        std::vector<int> placed_lds, placed_glb;
        n_lds = lds_base;
        n_glb = 0;
        auto first_fit = [&](const Obj &o, const std::vector<int> &placed, bool global, int limit) -> int {
            std::vector<std::pair<int, int>> busy;
            for (int pi : placed) {
                const Obj &p = objs[pi];
                const bool concurrent = p.owner >= 0 && o.owner >= 0 && p.owner != o.owner && (p.phase == o.phase || p.span || o.span);
                if (!concurrent && (p.death < o.birth || o.death < p.birth)) continue;
                const int off = global ? (p.slot & ~kSlotGlobal) : p.slot - lds_base;
                busy.push_back({off, off + p.size});
            }
            std::sort(busy.begin(), busy.end());
            int at = 0;
            for (auto &b : busy) {
                if (b.first - at >= o.size) break;
                if (b.second > at) at = b.second;
            }
            if (limit >= 0 && at + o.size > limit) return -1;
            return at;
        };
        for (int oi : order) {
            Obj &o = objs[oi];
            int at = o.force == 2 ? -1 : first_fit(o, placed_lds, false, lds_budget);
            if (at < 0 && o.force == 1) ok = false;
            if (at >= 0) {
                o.slot = at + lds_base;
                placed_lds.push_back(oi);
                if (at + o.size + lds_base > n_lds) n_lds = at + o.size + lds_base;
            } else {
                at = first_fit(o, placed_glb, true, -1);
                o.slot = kSlotGlobal | at;
                placed_glb.push_back(oi);
                if (at + o.size > n_glb) n_glb = at + o.size;
            }
            *o.field = o.slot;
        }
        return ok;
    };

    // first-fit placement fragments: when the default order does not fit the objects that must live in LDS, try the others
    auto allocate_packed = [&](std::vector<Obj> &objs, int lds_budget, int &n_lds, int &n_glb) -> bool {
        for (int mode = 0; mode < 3; mode++) {
            for (Obj &o : objs) o.slot = -1;
            if (allocate(objs, lds_budget, n_lds, n_glb, mode)) return true;
        }
        return false;
    };

    auto cluster_of = [&](int b) { return m.bodies[b].cluster; };
    auto build_layout = [&](Layout &L, int lds_budget, int lds_budget_rnea, bool with_xa, bool split = false) {
        L.split_aba = L.split_rnea = false;

        L.clusters = clusters;
        L.rnea_clusters = clusters;
        L.bodies = bodies;
        L.rnea_bodies = bodies;
        // ---- ABA ----
        std::vector<Obj> objs;
        for (int b = 0; b < nb; b++) {
            BodyRec &br = L.bodies[b];
            const int c = cluster_of(b);
            br.slot_sc = br.slot_v = br.slot_IA = br.slot_psi = br.slot_ccl = br.slot_v3 = br.slot_a3 = br.slot_f = -1;
            br.slot_Xa = br.parent_slot_Xa = -1;
            if (with_xa && br.has_child) objs.push_back({&br.slot_Xa, 12, 1, tF[c], tB[c], -1});
            if (br.has_child) {
                int first_child_bwd = tB[c], last_child_acc = tA[c];
                for (int j = b + 1; j < nb; j++)
                    if (bodies[j].parent == b) {
                        first_child_bwd = std::min(first_child_bwd, tB[cluster_of(j)]);
                        last_child_acc = std::max(last_child_acc, tA[cluster_of(j)]);
                    }
                if (br.jtype != GRBDA_JOINT_FREE) objs.push_back({&br.slot_sc, 2, 0, tF[c], tB[c], -1});
                objs.push_back({&br.slot_v, 6, 0, tF[c], tB[c], -1});
                if (!br.carry_in) {
                    objs.push_back({&br.slot_psi, 6, 1, first_child_bwd, tB[c], -1});
                    objs.push_back({&br.slot_IA, 21, 2, first_child_bwd, tB[c], -1});
                }
                objs.push_back({&br.slot_v3, 6, 0, tA[c], last_child_acc, -1});
                objs.push_back({&br.slot_a3, 6, 0, tA[c], last_child_acc, -1});
            }
            if (clusters[c].chained) objs.push_back({&br.slot_ccl, 6, 1, tB[c], tB[c], -1});
        }
        for (int c = 0; c < nc; c++) {
            ClusterRec &cr = L.clusters[c];
            // one block [K 6n][y0 n] per cluster (slot_y0 is derived below): the acceleration sweep fetches the
            // block of the next step ahead of time (Layout::acc_k)
            if (cr.kind != CK_FREE) objs.push_back({&cr.slot_K, 7 * cr.n, 3, tB[c], tA[c], -1});
            else {
                cr.slot_K = -1;
                objs.push_back({&cr.slot_y0, cr.n, 3, tB[c], tA[c], -1});
            }
            cr.slot_imp_fwd = cr.slot_imp_bwd = cr.slot_imp_acc = -1;
            if (cr.kind == CK_LOOP) {
                // work space: [K rows*k][chain 6k | per distinct trig argument: a, sin a, cos a, w.qd]
                const int keep = cr.k * (cr.n + 1) + cr.k + cr.k, tmp = cr.rows * cr.k + std::max(6 * cr.k, 4 * trig_args[c]);
                // (kernels.hip, ImpLayout) kept block: forward step -> acceleration step; work space: forward step
                objs.push_back({&cr.slot_imp_fwd, keep, 1, tF[c], tA[c], -1});
                objs.push_back({&cr.slot_imp_bwd, tmp, -1, tF[c], tF[c], -1});  // first pick: it is hammered with dependent accesses
            }
        }
        int nl = 0, ng = 0;
        if (split) {
            // [K | y0] blocks in the global slab, everything else in LDS (kernels.hip, Slots<T, true>)
            for (Obj &o : objs) o.force = 1;
            for (int c = 0; c < nc; c++)
                for (Obj &o : objs)
                    if (o.field == &L.clusters[c].slot_K || o.field == &L.clusters[c].slot_y0) o.force = 2;
            // ... and so do the backward accumulators of branching bodies (rarely touched, 27 scalars each)
            for (int b = 0; b < nb; b++)
                for (Obj &o : objs)
                    if (o.field == &L.bodies[b].slot_IA || o.field == &L.bodies[b].slot_psi) o.force = 2;
        }
        L.split_aba = allocate(objs, lds_budget, nl, ng) && split;
        L.n_lds_aba = nl;
        L.n_glb_aba = ng;
        for (int c = 0; c < nc; c++)
            if (L.clusters[c].kind != CK_FREE) L.clusters[c].slot_y0 = L.clusters[c].slot_K + 6 * L.clusters[c].n;
        // acc_k[s]: K block of step s when it is an acceleration step of a straight-line shape, else -1
        L.acc_k.assign(P.aba_steps.size() + 1, -1);
        for (size_t t = 0; t < P.aba_steps.size(); t++) {
            const Step &st = P.aba_steps[t];
            if (st.op == OP_ABA_ACC && !with_xa && L.clusters[st.cluster].shape != SHAPE_GENERIC)
                L.acc_k[t] = L.clusters[st.cluster].slot_K;
        }
        // first contributor to a body's backward accumulators: earliest backward step, and inside
        // one step the highest body index (bodies are visited in reverse order)
        std::vector<int> first(nb, -1);
        for (int j = 0; j < nb; j++) {
            const int p = bodies[j].parent;
            if (p < 0) continue;
            if (first[p] < 0) { first[p] = j; continue; }
            const int a = tB[cluster_of(first[p])], t = tB[cluster_of(j)];
            if (t < a || (t == a && j > first[p])) first[p] = j;
        }
        // the same for the inertia accumulator, which axisymmetric leaves never touch.  Writers of
        // body p's inertia slot: its non-axisymmetric children (in their cluster's backward step, highest
        // index first) and, after the bodies of a step, the cluster correction -F D^-1 F^T.
        struct Writer { int t, phase, order, body, cluster; };
        std::vector<Writer> firstw(nb, Writer{1 << 30, 0, 0, -1, -1});
        auto consider = [&](int p, const Writer &w) {
            Writer &f = firstw[p];
            if (w.t < f.t || (w.t == f.t && (w.phase < f.phase || (w.phase == f.phase && w.order < f.order)))) f = w;
        };
        for (int j = 0; j < nb; j++) {
            const int p = bodies[j].parent;
            if (p < 0 || bodies[j].axisym) continue;
            consider(p, Writer{tB[cluster_of(j)], 0, -j, j, -1});
        }
        for (int c = 0; c < nc; c++) {
            const int p = L.clusters[c].parent_body;
            L.clusters[c].corr_first_IA = 0;
            if (p >= 0 && !L.clusters[c].carry_out) consider(p, Writer{tB[c], 1, c, -1, c});
        }
        for (int p2 = 0; p2 < nb; p2++)
            if (firstw[p2].cluster >= 0) L.clusters[firstw[p2].cluster].corr_first_IA = 1;
        for (int b = 0; b < nb; b++) {
            BodyRec &br = L.bodies[b];
            br.acc_first_IA = 0;
            if (br.parent >= 0) {
                const BodyRec &pr = L.bodies[br.parent];
                br.parent_slot_v = pr.slot_v;
                br.parent_slot_Xa = pr.slot_Xa;
                br.parent_slot_IA = pr.slot_IA;
                br.parent_slot_psi = pr.slot_psi;
                br.parent_slot_v3 = pr.slot_v3;
                br.parent_slot_a3 = pr.slot_a3;
                br.acc_first = first[br.parent] == b;
                br.acc_first_IA = firstw[br.parent].body == b;
            } else {
                br.parent_slot_v = br.parent_slot_IA = br.parent_slot_psi = br.parent_slot_v3 = br.parent_slot_a3 = -1;
                br.acc_first = 0;
            }
            br.parent_slot_f = -1;
        }
        for (int c = 0; c < nc; c++) {
            ClusterRec &cr = L.clusters[c];
            if (cr.parent_body >= 0) {
                const BodyRec &pr = L.bodies[cr.parent_body];
                cr.parent_slot_IA = pr.slot_IA;
                cr.parent_slot_psi = pr.slot_psi;
                cr.parent_slot_v3 = pr.slot_v3;
                cr.parent_slot_a3 = pr.slot_a3;
            } else {
                cr.parent_slot_IA = cr.parent_slot_psi = cr.parent_slot_v3 = cr.parent_slot_a3 = -1;
            }
        }
        // ---- RNEA ----
        std::vector<Obj> robjs;
        for (int b = 0; b < nb; b++) {
            BodyRec &br = L.rnea_bodies[b];
            const int c = cluster_of(b);
            br.slot_sc = br.slot_v = br.slot_IA = br.slot_psi = br.slot_ccl = br.slot_v3 = br.slot_a3 = br.slot_f = -1;
            br.slot_Xa = br.parent_slot_Xa = -1;
            // straight-line shapes (fast layouts): a rotor and a childless link finish in the forward step
            const bool lean = !with_xa && clusters[c].shape != SHAPE_GENERIC;
            if (lean && !br.has_child) continue;
            if (br.jtype != GRBDA_JOINT_FREE && br.parent >= 0) robjs.push_back({&br.slot_sc, 2, 0, tRF[c], tRB[c], -1});
            robjs.push_back({&br.slot_f, 6, 1, tRF[c], tRB[c], -1});
            if (br.has_child) {
                int last_child_fwd = tRF[c];
                for (int j = b + 1; j < nb; j++)
                    if (bodies[j].parent == b) last_child_fwd = std::max(last_child_fwd, tRF[cluster_of(j)]);
                robjs.push_back({&br.slot_v, 6, 0, tRF[c], last_child_fwd, -1});
                robjs.push_back({&br.slot_a3, 6, 0, tRF[c], last_child_fwd, -1});
                if (with_xa) robjs.push_back({&br.slot_Xa, 12, 1, tRF[c], last_child_fwd, -1});
            }
        }
        for (int c = 0; c < nc; c++) {
            ClusterRec &cr = L.rnea_clusters[c];
            cr.slot_K = cr.slot_y0 = cr.slot_imp_fwd = cr.slot_imp_bwd = cr.slot_imp_acc = -1;
            // rotor torque of a SHAPE_REV_ROTOR cluster whose link has children, forward -> backward step
            if (!with_xa && cr.shape == SHAPE_REV_ROTOR && cr.child_mask) robjs.push_back({&cr.slot_y0, 1, 1, tRF[c], tRB[c], -1});
            cr.parent_slot_IA = cr.parent_slot_psi = cr.parent_slot_v3 = cr.parent_slot_a3 = -1;
            cr.carry_out = 0;
            if (cr.kind == CK_LOOP) {
                // work space: [K rows*k][chain 6k | per distinct trig argument: a, sin a, cos a, w.qd]
                const int keep = cr.k * (cr.n + 1) + cr.k + cr.k, tmp = cr.rows * cr.k + std::max(6 * cr.k, 4 * trig_args[c]);
                robjs.push_back({&cr.slot_imp_fwd, keep, 1, tRF[c], tRB[c], -1});
                robjs.push_back({&cr.slot_imp_bwd, tmp, -1, tRF[c], tRF[c], -1});
            }
        }
        if (split)
            for (Obj &o : robjs) o.force = 1;
        L.split_rnea = allocate(robjs, lds_budget_rnea, nl, ng) && split;
        L.n_lds_rnea = nl;
        L.n_glb_rnea = ng;
        for (int b = 0; b < nb; b++) {
            BodyRec &br = L.rnea_bodies[b];
            br.acc_first = 0;
            if (br.parent >= 0) {
                const BodyRec &pr = L.rnea_bodies[br.parent];
                br.parent_slot_v = pr.slot_v;
                br.parent_slot_a3 = pr.slot_a3;
                br.parent_slot_f = pr.slot_f;
                br.parent_slot_Xa = pr.slot_Xa;
            } else {
                br.parent_slot_v = br.parent_slot_a3 = br.parent_slot_f = -1;
            }
            br.parent_slot_IA = br.parent_slot_psi = br.parent_slot_v3 = -1;
        }
    };
    build_layout(P.lay32, lds.aba32, lds.rnea32, false);
    build_layout(P.lay64, lds.aba64, lds.rnea64, false);
    build_layout(P.lay32x, lds.aba32, lds.rnea32, true);
    // split layout of the fast f32 kernels; unusable (split_* false) when the LDS-only objects exceed the budget
    build_layout(P.lay32s, lds.aba32, lds.rnea32, false, true);
    for (const ClusterRec &cr : clusters)
        if (cr.kind == CK_LOOP) P.lay32s.split_aba = P.lay32s.split_rnea = false;
    build_layout(P.lay64x, lds.aba64, lds.rnea64, true);

    // ---- chain program of the f32 fast path (plan.h, ChainProgram; chain_kernels.hip) ----------------------------
    // constants of every axisymmetric rotor that the kernels would otherwise recompute per state: with q = 0 the
    // force per unit rotor acceleration at the parent body, X0^T h with h = I[:, z], does not depend on the state
    std::vector<int> rotor_pre(nb, -1);
    auto rotor_constants = [&](int b) {
        if (rotor_pre[b] >= 0) return rotor_pre[b];
        const double *C = &P.consts[bodies[b].cofs];
        const double *E = C, *r = C + 9, *I = C + 12;
        auto sid = [](int i, int j) { return i <= j ? (i * 6 - i * (i - 1) / 2 + (j - i)) : (j * 6 - j * (j - 1) / 2 + (i - j)); };
        double h[6], n[3], l[3], f[6];
        for (int i = 0; i < 6; i++) h[i] = I[sid(i, 2)];
        for (int i = 0; i < 3; i++) {
            n[i] = E[i] * h[0] + E[3 + i] * h[1] + E[6 + i] * h[2];
            l[i] = E[i] * h[3] + E[3 + i] * h[4] + E[6 + i] * h[5];
        }
        f[0] = n[0] + (r[1] * l[2] - r[2] * l[1]);
        f[1] = n[1] + (r[2] * l[0] - r[0] * l[2]);
        f[2] = n[2] + (r[0] * l[1] - r[1] * l[0]);
        f[3] = l[0]; f[4] = l[1]; f[5] = l[2];
        rotor_pre[b] = static_cast<int>(P.consts.size());
        for (int i = 0; i < 6; i++) P.consts.push_back(f[i]);
        P.consts.push_back(h[2]);
        return rotor_pre[b];
    };
    // ---- two-rotor differential clusters (plan.h, ChainDiff): shape test and constraint program -----------------------
    // ints   [n_args, n_atoms, tofs_d, 0] [per argument: W offset of its sin, cos, id atom or -1]
    //        per row: [n1, n2, n3] then the atoms' W offsets of the 1-, 2- and 3-factor terms
    // consts [per argument: w[4], b] [per atom: w[4], kappa (-1/2 for sin / cos, 0 for id)]
    //        K stream: per row, per term: coef, w[4] of every factor;  B stream: per row, per term: coef
    // (w in the kernel's coordinate order rotor1, rotor2, link1, link2)
    struct DiffShape { bool ok = false; int l1 = -1, l2 = -1, r[2] = {-1, -1}; int tofs_i = -1, n_atoms = 0; };
    std::vector<DiffShape> diff_shape(nc);
    for (int c = 0; c < nc; c++) {
        const ClusterRec &cr = clusters[c];
        const grbda_desc_cluster &cl = m.clusters[c];
        if (cr.kind != CK_LOOP || cr.cons_type != 1 || cr.n != 2 || cr.k != 4 || cr.rows != 2 || cr.parent_body < 0) continue;
        DiffShape ds;
        int nr = 0;
        for (int i = 0; i < 4; i++) {
            const int gb = cr.first_body + i;
            const BodyRec &br = bodies[gb];
            if (br.axisym && !br.has_child && br.parent == cr.parent_body) { if (nr < 2) ds.r[nr] = gb; nr++; }
            else if (br.parent == cr.parent_body && br.lam < 0) ds.l1 = ds.l1 < 0 ? gb : -2;
            else if (br.lam >= 0) ds.l2 = ds.l2 < 0 ? gb : -2;
        }
        bool good = nr == 2 && ds.l1 >= 0 && ds.l2 >= 0 && bodies[ds.l2].lam == ds.l1 && !bodies[ds.l1].axisym && !bodies[ds.l2].axisym;
        if (good)
            for (int j = 0; j < nb; j++)
                if (bodies[j].parent == ds.l1 && j != ds.l2) good = false;
        const int32_t *ip = m.ints + cl.int_offset;
        const double *dp = m.dbls + cl.dbl_offset;
        // the rotors carry the independent coordinates
        if (good) good = ip[ds.r[0] - cr.first_body] && ip[ds.r[1] - cr.first_body] && !ip[ds.l1 - cr.first_body] && !ip[ds.l2 - cr.first_body];
        if (!good) continue;
        const int order[4] = {ds.r[0] - cr.first_body, ds.r[1] - cr.first_body, ds.l1 - cr.first_body, ds.l2 - cr.first_body};
        struct Term { double coef; std::vector<int> atoms; };
        std::vector<std::array<double, 5>> args;
        std::vector<std::pair<int, int>> atoms;  // (argument, type 0 id / 1 sin / 2 cos)
        std::vector<Term> rows[2];
        const int32_t *tp = ip + 4;
        int nd = 0;
        for (int r = 0; r < 2 && good; r++) {
            const int nt = *tp++;
            for (int t = 0; t < nt && good; t++) {
                const int nf = *tp++;
                Term term;
                term.coef = dp[nd++];
                if (nf > 3) good = false;
                for (int f = 0; f < nf; f++) {
                    std::array<double, 5> a;
                    for (int j = 0; j < 4; j++) a[j] = dp[nd + order[j]];
                    a[4] = dp[nd + 4];
                    nd += 5;
                    const int type = *tp++;
                    if (type < 0 || type > 2) good = false;
                    int ai = -1;
                    for (size_t i = 0; i < args.size(); i++)
                        if (args[i] == a) ai = static_cast<int>(i);
                    if (ai < 0) { ai = static_cast<int>(args.size()); args.push_back(a); }
                    int at = -1;
                    for (size_t i = 0; i < atoms.size(); i++)
                        if (atoms[i] == std::make_pair(ai, type)) at = static_cast<int>(i);
                    if (at < 0) { at = static_cast<int>(atoms.size()); atoms.push_back({ai, type}); }
                    term.atoms.push_back(at);
                }
                if (nf > 0) rows[r].push_back(term);  // constants differentiate to nothing
            }
        }
        if (!good) continue;
        ds.ok = true;
        ds.n_atoms = static_cast<int>(atoms.size());
        ds.tofs_i = static_cast<int>(P.cints.size());
        const int tofs_d = static_cast<int>(P.consts.size());
        P.cints.push_back(static_cast<int>(args.size()));
        P.cints.push_back(ds.n_atoms);
        P.cints.push_back(tofs_d);
        P.cints.push_back(0);
        for (size_t i = 0; i < args.size(); i++) {
            int slot[3] = {-1, -1, -1};  // sin, cos, id
            for (size_t a = 0; a < atoms.size(); a++)
                if (atoms[a].first == static_cast<int>(i)) slot[atoms[a].second == 1 ? 0 : (atoms[a].second == 2 ? 1 : 2)] = 3 * static_cast<int>(a);
            for (int j = 0; j < 3; j++) P.cints.push_back(slot[j]);
            for (int j = 0; j < 5; j++) P.consts.push_back(args[i][j]);
        }
        for (const auto &a : atoms) {
            for (int j = 0; j < 4; j++) P.consts.push_back(args[a.first][j]);
            P.consts.push_back(a.second == 0 ? 0.0 : -0.5);
        }
        std::vector<double> bstream;
        for (int r = 0; r < 2; r++) {
            int cnt[4] = {0, 0, 0, 0};
            for (const Term &t : rows[r]) cnt[t.atoms.size()]++;
            for (int nf = 1; nf <= 3; nf++) P.cints.push_back(cnt[nf]);
            for (size_t nf = 1; nf <= 3; nf++)
                for (const Term &t : rows[r]) {
                    if (t.atoms.size() != nf) continue;
                    P.consts.push_back(t.coef);
                    bstream.push_back(t.coef);
                    for (int at : t.atoms) {
                        P.cints.push_back(3 * at);
                        for (int j = 0; j < 4; j++) P.consts.push_back(args[atoms[at].first][j]);
                    }
                }
        }
        P.consts.insert(P.consts.end(), bstream.begin(), bstream.end());
        diff_shape[c] = ds;
    }

    auto build_chain = [&](ChainProgram &CP, int lds_budget, RneaChainProgram *RP, int rnea_budget, int n_waves = 1, bool k_lds = false, bool lm_diffs = false) {
        CP = ChainProgram();
        CP.n_waves = n_waves;
        const bool lm = n_waves > 1;  // latency mode (plan.h, ChainProgram::n_waves)
        bool ok = sweep_mask == 7;
        // cluster classes: 0 free, 1 revolute, 2 revolute + axisymmetric rotor, 3 leaf pair (head of its parent link's backward
        // run), 4 revolute + general rotor, 5 two-rotor differential, 6 explicit pair with child clusters on link2 or in a
        // place class 3 does not cover (runs through the differential's segments with constant G), -1 unsupported
        // 7 generic cluster (plan.h, ChainGen): any other explicit cluster, URDF+ position loops, trig-polynomial constraints that
        // are not in the differential's shape; child clusters may hang off any of its bodies (tip = -2)
        std::vector<int> cls(nc, -1), tip(nc, -1), gen_rotor(nc, -1);
        const bool no_gen = std::getenv("GRBDA_NO_CHAIN_GEN") != nullptr;  // A/B switch: such models keep the interpreter
        auto is_diff = [&](int c) { return cls[c] == 5 || cls[c] == 6; };
        auto make_gen = [&](int c) {
            if (no_gen) { ok = false; return; }
            cls[c] = 7;
            tip[c] = -2;
        };
        std::vector<ChainPair> pair_of(nc);
        std::vector<std::array<int, 2>> pair_rotors(nc, std::array<int, 2>{-1, -1});
        for (int c = 0; c < nc && ok; c++) {
            const ClusterRec &cr = clusters[c];
            if (cr.kind == CK_FREE) {
                cls[c] = 0;
                tip[c] = cr.first_body;
            } else if (cr.kind == CK_STATIC && cr.shape != SHAPE_GENERIC) {
                cls[c] = cr.shape == SHAPE_REV ? 1 : 2;
                tip[c] = cr.link_body;
            } else if (cr.kind == CK_STATIC && cr.n == 1 && cr.k == 1 && cr.parent_body < 0) {
                // a single revolute link on the ground (the straight-line shapes of the interpreter need a parent body; the
                // runs do not)
                cls[c] = 1;
                tip[c] = cr.first_body;
            } else if (cr.kind == CK_STATIC && cr.n == 1 && cr.k == 2 && !cr.chained &&
                       (!bodies[cr.first_body].has_child || !bodies[cr.first_body + 1].has_child)) {
                // (also: a link and any rotor on the ground -- rotors there are not folded into a parent's constants)
                // a link and a rotor that is NOT axisymmetric about its joint axis (JVRC-1: every rotor carries the same
                // z-symmetric inertia whatever its axis): the rotor is evaluated at its own angle
                cls[c] = 4;
                const int f = cr.first_body;
                // the rotor is the childless body; of two childless bodies the one the coordinate does not drive 1:1
                int rot = !bodies[f].has_child ? f : f + 1;
                if (!bodies[f].has_child && !bodies[f + 1].has_child && P.consts[bodies[f].cofs + kBodyConstFixed] == 1.0) rot = f + 1;
                gen_rotor[c] = rot;
                tip[c] = rot == f ? f + 1 : f;
            } else if (cr.kind == CK_STATIC && cr.n == 2 && cr.k == 4) {
                // RevolutePairWithRotor shape: link1 and two axisymmetric rotors on the parent body, link2 on link1,
                // coordinates = the two link angles (RevolutePairWithRotorJoint.cpp:10-69), no child clusters
                int l1 = -1, l2 = -1, r[2] = {-1, -1}, nr = 0;
                for (int i = 0; i < 4; i++) {
                    const int gb = cr.first_body + i;
                    const BodyRec &br = bodies[gb];
                    if (br.axisym && !br.has_child && br.parent == cr.parent_body) { if (nr < 2) r[nr] = gb; nr++; }
                    else if (br.parent == cr.parent_body && br.lam < 0) l1 = l1 < 0 ? gb : -2;
                    else if (br.lam >= 0) l2 = l2 < 0 ? gb : -2;
                }
                bool good = nr == 2 && l1 >= 0 && l2 >= 0 && bodies[l2].lam == l1 && !bodies[l1].axisym && !bodies[l2].axisym;
                if (good)
                    for (int j = 0; j < nb; j++)
                        if (bodies[j].parent == l1 && j != l2) good = false;
                if (good) {
                    const double *g1 = &P.consts[bodies[l1].cofs + kBodyConstFixed], *g2 = &P.consts[bodies[l2].cofs + kBodyConstFixed];
                    good = g1[0] == 1.0 && g1[1] == 0.0 && g2[0] == 0.0 && g2[1] == 1.0;
                }
                if (good) {
                    // (the shape in the differential's terms too: class 6, decided here when link2 carries child clusters,
                    // below when a leaf pair sits where the head of a backward run cannot be)
                    DiffShape ds;
                    ds.ok = true; ds.l1 = l1; ds.l2 = l2; ds.r[0] = r[0]; ds.r[1] = r[1]; ds.tofs_i = -1; ds.n_atoms = 0;
                    diff_shape[c] = ds;
                    cls[c] = (bodies[l2].has_child || cr.parent_body < 0) ? 6 : 3;
                    if (cls[c] == 6) tip[c] = l2;
                    ChainPair &pr = pair_of[c];
                    pr = ChainPair();
                    pr.q_index = cr.q_index;
                    pr.v_index = cr.v_index;
                    pr.cofs[0] = bodies[l1].cofs; pr.cofs[1] = bodies[l2].cofs; pr.cofs[2] = bodies[r[0]].cofs; pr.cofs[3] = bodies[r[1]].cofs;
                    pair_rotors[c] = {r[0], r[1]};
                    pr.perm[0] = cyclic_shift(P.consts, pr.cofs[0]);
                    pr.perm[1] = cyclic_shift(P.consts, pr.cofs[1]);
                    if (std::getenv("GRBDA_DEBUG_CHAIN")) std::fprintf(stderr, "chain: cluster %d pair perms %d %d\n", c, pr.perm[0], pr.perm[1]);
                } else {
                    make_gen(c);
                }
            } else if (diff_shape[c].ok) {
                cls[c] = 5;  // two-rotor differential (implicit), plan.h ChainDiff
                tip[c] = diff_shape[c].l2;
            } else {
                make_gen(c);
            }
            // (links may hang off the ground -- fixed-base models: the runs then start from v = 0, a = -gravity and hand
            // their inertia to nobody; pair and differential clusters need a parent body)
            if ((cls[c] == 3 || cls[c] == 5) && cr.parent_body < 0) make_gen(c);
            if (!ok && std::getenv("GRBDA_DEBUG_CHAIN")) std::fprintf(stderr, "chain: cluster %d (k %d n %d kind %d shape %d) not covered\n", c, cr.k, cr.n, cr.kind, cr.shape);
        }
        // every child cluster must hang off the tip body of its parent cluster; a pair must be an only child of a link
        std::vector<std::vector<int>> ckids(nc);
        for (int c = 0; c < nc && ok; c++) {
            const int pb = clusters[c].parent_body;
            if (pb < 0) continue;
            const int pc = m.bodies[pb].cluster;
            if (tip[pc] != pb && tip[pc] != -2) {
                // (a child cluster on a rotor, on the first link of a pair, ...: the parent runs as a generic cluster)
                if (std::getenv("GRBDA_DEBUG_CHAIN")) std::fprintf(stderr, "chain: cluster %d hangs off body %d, not the tip %d of cluster %d\n", c, pb, tip[pc], pc);
                if (cls[pc] == 0) { ok = false; break; }
                make_gen(pc);
                if (!ok) break;
            }
            ckids[pc].push_back(c);
        }
        for (int c = 0; c < nc && ok; c++) {
            // child order = the depth-first order of the general schedule (kids[])
            ckids[c] = kids[c];
            for (int k : ckids[c])
                if (cls[k] == 3 && (ckids[c].size() != 1 || cls[c] == 0 || cls[c] == 3 || cls[c] == 7 || is_diff(c))) {
                    cls[k] = 6;  // a leaf pair next to siblings, or on the base / a pair / a differential: standalone segments
                    tip[k] = diff_shape[k].l2;
                }
        }
        if (ok) {
            // per-cluster master records
            std::vector<ChainLink> link_of(nc);
            std::vector<ChainFree> free_of(nc);
            std::vector<ChainDiff> diff_of(nc);
            std::vector<ChainGen> gen_of(nc);
            std::vector<std::vector<ChainGenBody>> gbody_of(nc);
            std::vector<int> gen_w_size(nc, 0), gen_scratch(nc, 0);
            std::vector<int> acc_slot(nc, -1);            // accumulator [IA 21][psi 6] of the tip body (several kid chains, or a free base)
            std::vector<Obj> objs;
            int n_glb = 0;
            auto glb = [&](int size) { const int at = n_glb; n_glb += size; return at | kSlotGlobal; };
            // k_lds (the fp32 latency-mode programs; their kernels are built with GRBDA_KLDS): the [K | y0] blocks and the base's accumulators
            // are LDS objects like everything else, and what does not fit overflows to the slab.  Two four-wavefront tiles per CU otherwise keep
            // ~100 KB of slab each in flight, 6.5 MB per XCD against 4 MB of L2 (measured: MIT Humanoid, 512 tiles, 0.0388 -> 0.0332 ms
            // with the accumulators alone, 0.0297 with the blocks)
            for (int c = 0; c < nc; c++) {
                const ClusterRec &cr = clusters[c];
                if (cls[c] == 0) {
                    ChainFree &f = free_of[c];
                    f = ChainFree();
                    const BodyRec &br = bodies[cr.first_body];
                    f.q_index = cr.q_index; f.v_index = cr.v_index; f.cofs = br.cofs; f.iofs = br.xofs >= 0 ? br.xofs : br.cofs + 12;
                    f.lds_v = f.lds_acc = f.lds_va = f.lds_acc2 = f.lds_acc3 = f.lds_acc4 = -1;
                    f.glb_y0 = k_lds ? -1 : glb(33);  // [y0 6] (+ OSIM pass: Cholesky factor of the base's articulated inertia, L 21 + 1/diag 6)
                } else if (cls[c] == 1 || cls[c] == 2 || cls[c] == 4) {
                    ChainLink &l = link_of[c];
                    l = ChainLink();
                    const BodyRec &br = bodies[tip[c]];
                    l.q_index = cr.q_index; l.v_index = cr.v_index; l.cofs = br.cofs;
                    l.rofs = cls[c] == 2 ? bodies[cr.rotor_body].cofs : (cls[c] == 4 ? bodies[gen_rotor[c]].cofs : -1);
                    l.rpre = cls[c] == 2 ? rotor_constants(cr.rotor_body) : -1;  // rofs >= 0 with rpre < 0: a general rotor
                    l.iofs = br.xofs >= 0 ? br.xofs : br.cofs + 12;
                    l.has_child = br.has_child;
                    l.lds_sv = l.lds_pv = l.lds_va = -1;
                    l.glb_k = k_lds ? -1 : glb(10);  // [K 6][y0][sin][cos] (+ OSIM pass: 1 / D)
                    l.perm = cyclic_shift(P.consts, br.cofs);
                    l.rperm = l.rofs >= 0 && l.rpre < 0 ? cyclic_shift(P.consts, l.rofs) : -1;
                    if (std::getenv("GRBDA_DEBUG_CHAIN")) std::fprintf(stderr, "chain: cluster %d class %d perm %d rotor perm %d\n", c, cls[c], l.perm, l.rperm);
                } else if (is_diff(c)) {
                    const DiffShape &ds = diff_shape[c];
                    ChainDiff &d = diff_of[c];
                    d = ChainDiff();
                    d.q_index = cr.q_index; d.v_index = cr.v_index;
                    if (cls[c] == 5) {
                        d.qpos[0] = ds.r[0] - cr.first_body; d.qpos[1] = ds.r[1] - cr.first_body;
                        d.qpos[2] = ds.l1 - cr.first_body; d.qpos[3] = ds.l2 - cr.first_body;
                    } else {  // the coordinates ARE the link angles (G rows (1, 0), (0, 1) checked above)
                        d.qpos[0] = d.qpos[1] = 0;
                        d.qpos[2] = 0; d.qpos[3] = 1;
                    }
                    d.gofs = static_cast<int>(P.consts.size());
                    for (int r2 = 0; r2 < 2; r2++)
                        for (int a = 0; a < 2; a++)
                            P.consts.push_back(cls[c] == 5 ? (r2 == a ? 1.0 : 0.0) : P.consts[bodies[ds.r[r2]].cofs + kBodyConstFixed + a]);
                    d.cofs[0] = bodies[ds.l1].cofs; d.cofs[1] = bodies[ds.l2].cofs; d.cofs[2] = bodies[ds.r[0]].cofs; d.cofs[3] = bodies[ds.r[1]].cofs;
                    d.rpre[0] = rotor_constants(ds.r[0]);
                    d.rpre[1] = rotor_constants(ds.r[1]);
                    d.iofs = bodies[ds.l2].xofs >= 0 ? bodies[ds.l2].xofs : bodies[ds.l2].cofs + 12;
                    d.lds_pv = d.lds_sv = d.lds_acc = d.lds_acc_out = d.lds_pva = d.lds_va = d.lds_w = -1;
                    d.tofs_i = ds.tofs_i;
                    d.tofs_d = ds.tofs_i >= 0 ? P.cints[ds.tofs_i + 2] : 0;
                    d.glb_k = glb(27);  // [K 12][y0 2][X 4][g 2][s1 c1 s2 c2] (+ OSIM pass: D^-1 (3))
                } else if (cls[c] == 7) {
                    ChainGen &g = gen_of[c];
                    g = ChainGen();
                    const int k = cr.k, n = cr.n;
                    g.q_index = cr.q_index; g.v_index = cr.v_index;
                    g.k = k; g.n = n; g.rows = cr.rows;
                    g.kind = cr.kind == CK_LOOP ? (cr.cons_type == 1 ? 2 : 1) : 0;
                    g.iofs = cr.iofs; g.dofs = cr.dofs;
                    g.has_parent = cr.parent_body >= 0;
                    g.lds_pv = g.lds_acc_out = g.lds_pva = g.lds_w = g.lds_wf = g.keep = -1;
                    g.glb_k = glb(7 * n);
                    std::vector<ChainGenBody> &gb = gbody_of[c];
                    gb.assign(k, ChainGenBody());
                    bool kids_here = false;
                    for (int i = 0; i < k; i++) {
                        const BodyRec &br = bodies[cr.first_body + i];
                        ChainGenBody &b = gb[i];
                        b.cofs = br.cofs;
                        b.iofs = br.xofs >= 0 ? br.xofs : br.cofs + 12;
                        b.lam = br.lam >= 0 ? br.lam - cr.first_body : -1;
                        b.axis = br.axis;
                        b.axisym = br.axisym;
                        b.acc_w = b.up_w = b.lds_acc = b.lds_va = b.pva = b.ind_a = b.dep_r = b.lds_v = -1;
                        for (int j = 0; j < nb; j++)
                            if (bodies[j].parent == cr.first_body + i && m.bodies[j].cluster != c) kids_here = true;
                    }
                    // scratch of the constraint evaluation (gen_segments.h): K [rows x k], then per loop side 6 slots per joint of
                    // its path + 3, or 4 slots per distinct argument of a trig-polynomial constraint
                    int scratch = 0;
                    if (g.kind) {  // cints[iofs]: header, n_ind, ind[], n_dep, dep[]
                        const int32_t *ip = &P.cints[cr.iofs];
                        for (int a = 0; a < ip[1]; a++) gb[ip[2 + a]].ind_a = a;
                        for (int r = 0; r < cr.rows; r++) gb[ip[3 + ip[1] + r]].dep_r = r;
                    }
                    if (g.kind == 1) {
                        const int32_t *ip = &P.cints[cr.iofs];
                        const int32_t *lp = ip + 3 + ip[1] + cr.rows;
                        scratch = cr.rows * k;
                        for (int l = 0; l < ip[0]; l++) {
                            const int np = lp[0], ns = lp[1 + np];
                            scratch += 6 * (np + ns) + 6;
                            lp += 3 + np + ns;
                        }
                    } else if (g.kind == 2) {
                        scratch = cr.rows * k + 4 * trig_args[c];
                    }
                    gen_scratch[c] = scratch;
                    int w = 2 * k + std::max(6 * k, scratch);
                    // (IA, psi) inside the cluster: to the body right before in registers, else through an accumulator of the work area
                    for (int i = k - 1; i >= 0; i--) {
                        ChainGenBody &b = gb[i];
                        if (b.lam < 0) continue;
                        if (b.lam == i - 1) {
                            b.carry_up = 1;
                            gb[i - 1].carry_in = 1;
                            continue;
                        }
                        ChainGenBody &pl = gb[b.lam];
                        if (pl.acc_w < 0) {
                            pl.acc_w = w;
                            w += 27;
                            b.up_first = 1;
                        }
                        b.up_w = pl.acc_w;
                    }
                    if (g.kind && !kids_here) {  // the kept block of a childless implicit cluster lives inside the work area
                        g.keep = -2 - w;          // (resolved against lds_w after allocation)
                        w += cr.rows * (n + 2);
                    } else if (g.kind) {
                        g.keep = glb(cr.rows * (n + 2));
                    }
                    g.has_fwd = kids_here ? 1 : 0;
                    gen_w_size[c] = w;
                } else {
                    pair_of[c].glb_k = k_lds ? -1 : glb(21);  // [K 12][y0 2] (+ OSIM pass: D^-1 (3), sin / cos of the two links (4))
                    pair_of[c].rpre[0] = rotor_constants(pair_rotors[c][0]);
                    pair_of[c].rpre[1] = rotor_constants(pair_rotors[c][1]);
                    pair_of[c].lds_pv = pair_of[c].lds_pva = -1;
                }
            }
            // chains: follow single link children; a single pair child becomes the head of the backward run
            auto is_link = [](int k) { return k == 1 || k == 2 || k == 4; };
            struct Chain { std::vector<int> cl; int pair = -1; std::vector<int> kid_chains; int parent_cluster = -1; bool diff = false; bool gen = false; };
            std::vector<Chain> chains;
            std::function<int(int)> make_chain = [&](int c0) -> int {
                Chain ch;
                ch.parent_cluster = clusters[c0].parent_body >= 0 ? m.bodies[clusters[c0].parent_body].cluster : -1;
                int c = c0;
                for (;;) {
                    ch.cl.push_back(c);
                    // (links with no rotor, an axisymmetric rotor and a general rotor may share a run: the kind is a branch on
                    // the link record)
                    if (!is_diff(c) && cls[c] != 7 && ckids[c].size() == 1 && is_link(cls[ckids[c][0]])) { c = ckids[c][0]; continue; }
                    break;
                }
                ch.diff = is_diff(c0);  // a differential is a chain of its own; its child clusters hang off link2
                ch.gen = cls[c0] == 7;  // so is a generic cluster; its child clusters hang off any of its bodies
                const int tipc = ch.cl.back();
                const int id = static_cast<int>(chains.size());
                chains.push_back(ch);
                if (ckids[tipc].size() == 1 && cls[ckids[tipc][0]] == 3) {
                    chains[id].pair = ckids[tipc][0];
                } else {
                    for (int k : ckids[tipc]) {
                        const int kid = make_chain(k);
                        chains[id].kid_chains.push_back(kid);
                    }
                }
                return id;
            };
            // time stamps of segments
            struct SegTimes { int fwd = -1, bwd = -1, acc = -1, pair_acc = -1; };
            std::vector<SegTimes> ct;
            std::vector<int> root_frees, free_kids_first;
            // emission helpers work on chain ids; link records are filled after allocation, so remember (seg, cluster list)
            struct RunRef { int seg; std::vector<int> cl; };
            std::vector<RunRef> runs;
            std::vector<int> owner_of;  // latency mode: the wavefront that runs the chain (limbs below the base are dealt out)
            int cur_owner = 0;
            auto push_seg = [&](ChainSeg sg) { sg.owner = cur_owner; CP.segs.push_back(sg); return static_cast<int>(CP.segs.size()) - 1; };
            std::function<void(int)> emit_fb = [&](int id) {
                const Chain ch = chains[id];
                cur_owner = owner_of[id];
                if (ch.gen) {
                    ChainSeg sg = ChainSeg();
                    if (gen_of[ch.cl[0]].has_fwd) {
                        sg.op = SEG_GEN_FWD;
                        ct[id].fwd = push_seg(sg);
                    }
                    for (int k : ch.kid_chains) emit_fb(k);
                    cur_owner = owner_of[id];
                    sg.op = SEG_GEN_BWD;
                    ct[id].bwd = push_seg(sg);
                    if (ct[id].fwd < 0) ct[id].fwd = ct[id].bwd;
                    return;
                }
                if (ch.diff) {
                    ChainSeg sg = ChainSeg();
                    sg.op = SEG_DIFF_FWD;
                    ct[id].fwd = push_seg(sg);
                    for (int k : ch.kid_chains) emit_fb(k);
                    cur_owner = owner_of[id];
                    sg.op = SEG_DIFF_BWD;
                    ct[id].bwd = push_seg(sg);
                    return;
                }
                {   // every link of the chain, the tip included: the backward run reads [sin, cos, v] of all of them
                    ChainSeg sg = ChainSeg();
                    sg.op = SEG_RUN_FWD;
                    const int t = push_seg(sg);
                    runs.push_back({t, ch.cl});
                    ct[id].fwd = t;
                }
                for (int k : ch.kid_chains) emit_fb(k);
                cur_owner = owner_of[id];
                ChainSeg sg = ChainSeg();
                sg.op = SEG_RUN_BWD;
                const int tipc = ch.cl.back();
                sg.head = ch.pair >= 0 ? HEAD_PAIR : (ch.kid_chains.empty() ? HEAD_LEAF : HEAD_SLOT);
                const int t = push_seg(sg);
                std::vector<int> rev(ch.cl.rbegin(), ch.cl.rend());
                runs.push_back({t, rev});
                ct[id].bwd = t;
                (void)tipc;
            };
            std::function<void(int)> emit_acc = [&](int id) {
                const Chain ch = chains[id];
                cur_owner = owner_of[id];
                ChainSeg sg = ChainSeg();
                if (ch.diff || ch.gen) {
                    sg.op = ch.gen ? SEG_GEN_ACC : SEG_DIFF_ACC;
                    ct[id].acc = push_seg(sg);
                    for (int k : ch.kid_chains) emit_acc(k);
                    return;
                }
                sg.op = SEG_RUN_ACC;
                const int t = push_seg(sg);
                runs.push_back({t, ch.cl});
                ct[id].acc = t;
                if (ch.pair >= 0) {
                    ChainSeg ps = ChainSeg();
                    ps.op = SEG_PAIR_ACC;
                    ct[id].pair_acc = push_seg(ps);
                }
                for (int k : ch.kid_chains) emit_acc(k);
            };
            // roots
            std::vector<std::vector<int>> free_chains(nc);
            std::vector<int> t_free_fwd(nc, -1), t_free_bwd(nc, -1), t_free_acc(nc, -1);
            std::vector<int> ground_chains;  // chains whose first link hangs off the ground
            for (int c = 0; c < nc; c++) {
                if (cls[c] == 0)
                    for (int k : ckids[c]) free_chains[c].push_back(make_chain(k));
                else if (clusters[c].parent_body < 0)
                    ground_chains.push_back(make_chain(c));
            }
            ct.assign(chains.size(), SegTimes());
            owner_of.assign(chains.size(), 0);
            if (lm) {
                // one floating base with at least n_waves limbs, nothing on the ground, no generic segments; differential segments only in the
                // programs whose kernels carry them (lm_diffs: the fp32 latency-mode kernels); the limbs are dealt to the wavefronts by link count
                int n_free = 0, base = -1;
                for (int c = 0; c < nc; c++)
                    if (cls[c] == 0) { n_free++; base = c; }
                bool any_diff = false;
                for (const Chain &ch : chains) any_diff = any_diff || (ch.diff && !lm_diffs) || ch.gen;
                if (n_free != 1 || !ground_chains.empty() || any_diff || static_cast<int>(free_chains[base].size()) < n_waves || n_waves > 4) {
                    ok = false;
                } else {
                    std::function<int(int)> weight = [&](int id) {
                        int w = chains[id].diff ? 4 : static_cast<int>(chains[id].cl.size()) + (chains[id].pair >= 0 ? 2 : 0);  // (a differential: two links, two rotors, the constraint)
                        for (int k : chains[id].kid_chains) w += weight(k);
                        return w;
                    };
                    std::function<void(int, int)> give = [&](int id, int o) {
                        owner_of[id] = o;
                        for (int k : chains[id].kid_chains) give(k, o);
                    };
                    std::vector<int> load(n_waves, 0);
                    for (int id : free_chains[base]) {
                        int o = 0;
                        for (int w2 = 1; w2 < n_waves; w2++)
                            if (load[w2] < load[o]) o = w2;
                        give(id, o);
                        load[o] += weight(id);
                    }
                }
            }
            // global: what crosses the barrier went through the global slab (the limbs' accumulators), so the stores must have
            // landed; otherwise it is LDS only and the wavefronts need not drain their [K | y0] stores
            auto barrier = [&](bool global) {
                if (!lm) return;
                cur_owner = 0;
                ChainSeg sg = ChainSeg();
                sg.op = SEG_BARRIER;
                sg.head = global ? 1 : 0;
                push_seg(sg);
            };
            for (int id : ground_chains) emit_fb(id);
            for (int c = 0; c < nc; c++) {
                if (cls[c] != 0) continue;
                ChainSeg sg = ChainSeg();
                sg.op = SEG_FREE_FWD;
                cur_owner = 0;
                t_free_fwd[c] = push_seg(sg);
                barrier(false);  // the base's velocity is in LDS: the limbs may start
                for (int id : free_chains[c]) emit_fb(id);
                barrier(true);  // every limb has handed its inertia / bias to the base's accumulators (global slab)
                sg.op = SEG_FREE_BWD;
                cur_owner = 0;
                t_free_bwd[c] = push_seg(sg);
            }
            for (int c = 0; c < nc; c++) {
                if (cls[c] != 0) continue;
                ChainSeg sg = ChainSeg();
                sg.op = SEG_FREE_ACC;
                cur_owner = 0;
                t_free_acc[c] = push_seg(sg);
                barrier(false);  // the base's acceleration is in LDS
                for (int id : free_chains[c]) emit_acc(id);
            }
            for (int id : ground_chains) emit_acc(id);
            // ---- the inverse-dynamics program on the same chains (plan.h, RneaChainProgram) ----
            bool any_gen = false;
            for (const Chain &ch : chains) any_gen = any_gen || ch.gen;
            (void)any_gen;
            if (RP) {
                RneaChainProgram &R = *RP;
                R = RneaChainProgram();
                std::vector<RneaLink> rl(nc);
                std::vector<RneaPair> rp(nc);
                std::vector<RneaFree> rf(nc);
                std::vector<RneaDiff> rd(nc);
                std::vector<ChainGen> rg(nc);                      // generic clusters: the field use of gen_rnea_segments.h
                std::vector<std::vector<ChainGenBody>> rgb(nc);
                std::vector<Obj> robjs;
                struct RRun { int seg; std::vector<int> cl; };
                std::vector<RRun> rruns;
                std::vector<int> rt_fwd(chains.size(), -1), rt_bwd(chains.size(), -1), rt_pair(chains.size(), -1);
                std::vector<int> rt_free_fwd(nc, -1), rt_free_bwd(nc, -1);
                int r_owner = 0;
                auto rpush = [&](int op) { RneaSeg sg = RneaSeg(); sg.op = op; sg.lds_pva = sg.lds_pf = -1; sg.owner = r_owner; R.segs.push_back(sg); return static_cast<int>(R.segs.size()) - 1; };
                const bool rlm = lm && ok;  // (the limbs were dealt out above: owner_of)
                std::function<int(int)> remit = [&](int id) -> int {  // returns the last forward-type segment of the subtree
                    const Chain ch = chains[id];
                    r_owner = rlm ? owner_of[id] : 0;
                    if (ch.diff || ch.gen) {
                        int last = rt_fwd[id] = rpush(ch.gen ? RSEG_GEN_FWD : RSEG_DIFF_FWD);
                        for (int k : ch.kid_chains) last = std::max(last, remit(k));
                        rt_bwd[id] = rpush(ch.gen ? RSEG_GEN_BWD : RSEG_DIFF_BWD);
                        return last;
                    }
                    rt_fwd[id] = rpush(RSEG_RUN_FWD);
                    rruns.push_back({rt_fwd[id], ch.cl});
                    int last = rt_fwd[id];
                    if (ch.pair >= 0) last = rt_pair[id] = rpush(RSEG_PAIR);
                    for (int k : ch.kid_chains) last = std::max(last, remit(k));
                    r_owner = rlm ? owner_of[id] : 0;
                    rt_bwd[id] = rpush(RSEG_RUN_BWD);
                    std::vector<int> rev(ch.cl.rbegin(), ch.cl.rend());
                    rruns.push_back({rt_bwd[id], rev});
                    return last;
                };
                std::vector<int> last_fwd_of(chains.size(), -1);
                for (int id : ground_chains) remit(id);
                for (int c = 0; c < nc; c++) {
                    if (cls[c] != 0) continue;
                    r_owner = 0;
                    rt_free_fwd[c] = rpush(RSEG_FREE_FWD);
                    if (rlm) rpush(RSEG_BARRIER);  // the base's [v | a] and its force rows are in LDS: the limbs may start
                    for (int id : free_chains[c]) remit(id);
                    r_owner = 0;
                    if (rlm) rpush(RSEG_BARRIER);  // every limb has added its force
                    rt_free_bwd[c] = rpush(RSEG_FREE_BWD);
                }
                // last segment that reads the [v, a] of a chain's tip: the forward runs / pair of its kid chains
                for (size_t id = 0; id < chains.size(); id++) {
                    int last = rt_fwd[id];
                    if (chains[id].pair >= 0) last = rt_pair[id];
                    for (int k : chains[id].kid_chains) last = std::max(last, rt_fwd[k]);
                    last_fwd_of[id] = last;
                }
                for (int c = 0; c < nc; c++) {
                    const ClusterRec &cr = clusters[c];
                    if (cls[c] == 0) {
                        RneaFree &f = rf[c];
                        f = RneaFree();
                        f.q_index = cr.q_index; f.v_index = cr.v_index; f.cofs = bodies[cr.first_body].cofs;
                        f.lds_va = f.lds_f = f.lds_f2 = f.lds_f3 = f.lds_f4 = -1;
                        int last = rt_free_fwd[c];
                        for (int id : free_chains[c]) last = std::max(last, rt_fwd[id]);
                        if (!free_chains[c].empty()) robjs.push_back({&f.lds_va, 12, 0, rt_free_fwd[c], last, -1, 1});
                        robjs.push_back({&f.lds_f, 6, 0, rt_free_fwd[c], rt_free_bwd[c], -1, 1});
                        if (rlm) {  // one force row block per wavefront: no two wavefronts read-modify-write the same rows
                            int32_t *const extra[3] = {&f.lds_f2, &f.lds_f3, &f.lds_f4};
                            for (int o = 1; o < n_waves && o < 4; o++) robjs.push_back({extra[o - 1], 6, 0, rt_free_fwd[c], rt_free_bwd[c], -1, 1});
                        }
                    } else if (cls[c] == 1 || cls[c] == 2 || cls[c] == 4) {
                        RneaLink &l = rl[c];
                        l = RneaLink();
                        l.q_index = cr.q_index; l.v_index = cr.v_index; l.cofs = bodies[tip[c]].cofs;
                        l.rofs = cls[c] == 2 ? bodies[cr.rotor_body].cofs : (cls[c] == 4 ? bodies[gen_rotor[c]].cofs : -1);
                        l.general_rotor = cls[c] == 4;
                        l.perm = link_of[c].perm; l.rperm = link_of[c].rperm;
                        l.lds_blk = l.lds_va = l.lds_pf = -1;
                    } else if (cls[c] == 7) {
                        rg[c] = gen_of[c];
                        rg[c].lds_pv = rg[c].lds_acc_out = rg[c].lds_pva = rg[c].lds_w = rg[c].lds_wf = rg[c].keep = rg[c].glb_k = -1;
                        rg[c].has_fwd = 1;
                        rg[c].need_acc = 0;
                        rgb[c] = gbody_of[c];
                        for (ChainGenBody &b : rgb[c]) b.lds_acc = b.lds_va = b.pva = b.lds_v = b.acc_w = b.up_w = -1;
                        {   // [v 6][a 6] of the bodies with in-cluster children that do not come right after them (a body right
                            // after its parent takes the pair over in registers: up_w stays -1), behind the [sin, cos] rows of the
                            // work area
                            int w = 2 * rg[c].k;
                            for (size_t i = 0; i < rgb[c].size(); i++) {
                                ChainGenBody &b = rgb[c][i];
                                if (b.lam >= 0 && b.lam != static_cast<int>(i) - 1 && rgb[c][b.lam].acc_w < 0) { rgb[c][b.lam].acc_w = w; w += 12; }
                            }
                            for (size_t i = 0; i < rgb[c].size(); i++) {
                                ChainGenBody &b = rgb[c][i];
                                if (b.lam >= 0 && b.lam != static_cast<int>(i) - 1) b.up_w = rgb[c][b.lam].acc_w;
                            }
                            rg[c].reserved[0] = w - 2 * rg[c].k;  // slots of those pairs
                            rg[c].reserved[1] = 0;                // (1: the single-cluster layout below)
                        }
                    } else if (is_diff(c)) {
                        RneaDiff &d = rd[c];
                        d = RneaDiff();
                        d.q_index = cr.q_index; d.v_index = cr.v_index;
                        for (int i = 0; i < 4; i++) { d.qpos[i] = diff_of[c].qpos[i]; d.cofs[i] = diff_of[c].cofs[i]; }
                        d.lds_pva = d.lds_pf = d.lds_blk = d.lds_va = d.lds_w = -1;
                        d.tofs_i = diff_shape[c].tofs_i;
                        d.gofs = diff_of[c].gofs;
                    } else {
                        RneaPair &pr = rp[c];
                        pr = RneaPair();
                        pr.q_index = cr.q_index; pr.v_index = cr.v_index;
                        for (int i = 0; i < 4; i++) pr.cofs[i] = pair_of[c].cofs[i];
                        pr.lds_pva = pr.lds_pf = -1;
                    }
                }
                for (size_t id = 0; id < chains.size(); id++) {
                    const Chain &ch = chains[id];
                    struct RTag {  // on leaving the iteration: latency-mode programs tag the chain's objects with its wavefront (limbs of different
                                   // wavefronts run concurrently: their objects never share rows, whatever the segment order says)
                        std::vector<Obj> &v; size_t from; int owner; bool on;
                        ~RTag() { if (on) for (size_t i = from; i < v.size(); i++) v[i].owner = owner; }
                    } rtag{robjs, robjs.size(), rlm ? owner_of[id] : -1, rlm};
                    if (ch.gen) {
                        const int c = ch.cl[0];
                        ChainGen &g = rg[c];
                        const int k = g.k;
                        if (chains.size() == 1 && nc == 1 && !g.has_parent) {
                            // the cluster is the whole model (rnea_gen1_kernel): ONE object, [sin, cos 2k][f 6k][v | a pairs] with the
                            // constraint's scratch over the forces and pairs (it is done before the first force is written), then
                            // the kept block; the backward segment reads [sin, cos] where the forward one left them
                            const int pairs = g.reserved[0];
                            const int mid = std::max(6 * k + pairs, gen_scratch[c]);
                            for (ChainGenBody &b : rgb[c]) {
                                if (b.acc_w >= 0) b.acc_w += 6 * k;
                                if (b.up_w >= 0) b.up_w += 6 * k;
                            }
                            g.reserved[1] = 1;
                            g.reserved[2] = 2 * k + mid;  // offset of the kept block
                            robjs.push_back({&g.lds_w, 2 * k + mid + (g.kind ? g.rows * (g.n + 2) : 0), 0, rt_fwd[id], rt_bwd[id], -1, 1, 1});
                            continue;
                        }
                        robjs.push_back({&g.lds_w, 2 * k + std::max(g.reserved[0], gen_scratch[c]), 0, rt_fwd[id], rt_fwd[id], -1, 1, 1});
                        robjs.push_back({&g.glb_k, 8 * k, 0, rt_fwd[id], rt_bwd[id], -1, 1, 1});
                        if (g.kind) robjs.push_back({&g.keep, g.rows * (g.n + 2), 0, rt_fwd[id], rt_bwd[id], -1, 1, 1});
                        for (int i = 0; i < k; i++) {
                            const int body = clusters[c].first_body + i;
                            int last = -1;
                            for (int kc : ch.kid_chains)
                                if (clusters[chains[kc].cl.front()].parent_body == body) last = std::max(last, rt_fwd[kc]);
                            if (last >= 0) robjs.push_back({&rgb[c][i].lds_va, 12, 0, rt_fwd[id], last, -1, 1, 1});
                        }
                        continue;
                    }
                    if (ch.diff) {
                        RneaDiff &d = rd[ch.cl[0]];
                        // one object: [f2 6][sin, cos 4][X 4] then [v 6][a 6] when child segments follow; the work space of the
                        // constraint evaluation shares it (everything else is written after the constraint is done)
                        const int keep = 14 + (ch.kid_chains.empty() ? 0 : 12);
                        robjs.push_back({&d.lds_blk, std::max(keep, 3 * diff_shape[ch.cl[0]].n_atoms), 0, rt_fwd[id], rt_bwd[id], -1, 1, 1});
                        continue;
                    }
                    for (int c : ch.cl) robjs.push_back({&rl[c].lds_blk, 9, 0, rt_fwd[id], rt_bwd[id], -1, 1});
                    const int tipc = ch.cl.back();
                    if (!ch.kid_chains.empty() || ch.pair >= 0)
                        robjs.push_back({&rl[tipc].lds_va, 12, 0, rt_fwd[id], last_fwd_of[id], -1, 1});
                }
                int rn_lds = 0, rn_glb = 0;
                bool rok = allocate_packed(robjs, rnea_budget, rn_lds, rn_glb);
                if (!rok && !lm) {
                    // second try: the links' [f | sin, cos | rotor torque] blocks take what LDS the other objects leave
                    // and otherwise go to the wave's global slab
                    for (Obj &o : robjs)
                        if (o.size == 9 && !o.tag) { o.force = 0; o.prio = 1; o.slot = -1; }
                    rok = allocate(robjs, rnea_budget, rn_lds, rn_glb);
                }
                if (!rok && std::getenv("GRBDA_DEBUG_CHAIN")) {
                    std::fprintf(stderr, "chain (rnea): LDS objects need more than %d slots (got to %d)\n", rnea_budget, rn_lds);
                    for (const Obj &o : robjs) std::fprintf(stderr, "  obj size %d [%d, %d] slot %d\n", o.size, o.birth, o.death, o.slot);
                }
                if (rok) {
                    for (size_t id = 0; id < chains.size(); id++)
                        if (chains[id].diff) {
                            RneaDiff &d = rd[chains[id].cl[0]];
                            d.lds_w = d.lds_blk;
                            d.lds_va = chains[id].kid_chains.empty() ? -1 : d.lds_blk + 14;
                        }
                    auto f_slot_of_body = [&](int b, int owner = 0) -> int {
                        if (b < 0) return -1;  // ground
                        const int c = m.bodies[b].cluster;
                        if (is_diff(c)) return rd[c].lds_blk;
                        if (cls[c] == 7) return rg[c].glb_k + 8 * (b - clusters[c].first_body);
                        if (cls[c] != 0) return rl[c].lds_blk;
                        const RneaFree &f = rf[c];
                        return !rlm || owner == 0 ? f.lds_f : (owner == 1 ? f.lds_f2 : (owner == 2 ? f.lds_f3 : f.lds_f4));
                    };
                    // (latency mode: the wavefront of the chain a cluster belongs to)
                    std::vector<int> owner_of_cluster(nc, 0);
                    if (rlm)
                        for (size_t id2 = 0; id2 < chains.size(); id2++) {
                            for (int c2 : chains[id2].cl) owner_of_cluster[c2] = owner_of[id2];  // (a differential is a chain of its one cluster)
                            if (chains[id2].pair >= 0) owner_of_cluster[chains[id2].pair] = owner_of[id2];
                        }
                    auto va_slot_of_body2 = [&](int b) -> int {
                        if (b < 0) return -1;
                        const int c = m.bodies[b].cluster;
                        if (is_diff(c)) return rd[c].lds_va;
                        if (cls[c] == 7) return rgb[c][b - clusters[c].first_body].lds_va;
                        return cls[c] == 0 ? rf[c].lds_va : rl[c].lds_va;
                    };
                    for (int c = 0; c < nc; c++)
                        if (cls[c] == 7 && rg[c].reserved[1]) {
                            rg[c].glb_k = rg[c].lds_w + 2 * rg[c].k;
                            rg[c].keep = rg[c].kind ? rg[c].lds_w + rg[c].reserved[2] : -1;
                        }
                    for (int c = 0; c < nc; c++) {
                        const int pb = clusters[c].parent_body;
                        if (cls[c] == 7) { rg[c].lds_pva = va_slot_of_body2(pb); rg[c].lds_acc_out = f_slot_of_body(pb); }
                        if (cls[c] == 1 || cls[c] == 2 || cls[c] == 4) rl[c].lds_pf = f_slot_of_body(pb, owner_of_cluster[c]);
                        if (cls[c] == 3) { rp[c].lds_pva = va_slot_of_body2(pb); rp[c].lds_pf = f_slot_of_body(pb, owner_of_cluster[c]); }
                        if (is_diff(c)) { rd[c].lds_pva = va_slot_of_body2(pb); rd[c].lds_pf = f_slot_of_body(pb, owner_of_cluster[c]); }
                    }
                    for (const RRun &r : rruns) {
                        RneaSeg &sg = R.segs[r.seg];
                        sg.first = static_cast<int>(R.links.size());
                        sg.count = static_cast<int>(r.cl.size());
                        for (int c : r.cl) R.links.push_back(rl[c]);
                    }
                    for (size_t id = 0; id < chains.size(); id++) {
                        const Chain &ch = chains[id];
                        const int pb = clusters[ch.cl.front()].parent_body;
                        R.segs[rt_fwd[id]].lds_pva = va_slot_of_body2(pb);
                        R.segs[rt_bwd[id]].lds_pf = f_slot_of_body(pb, rlm ? owner_of[id] : 0);
                        if (ch.diff) {
                            R.segs[rt_fwd[id]].first = R.segs[rt_bwd[id]].first = static_cast<int>(R.diffs.size());
                            R.diffs.push_back(rd[ch.cl[0]]);
                        }
                        if (ch.gen) {
                            if (std::getenv("GRBDA_DEBUG_CHAIN")) {
                                const ChainGen &g = rg[ch.cl[0]];
                                std::fprintf(stderr, "rnea gen cluster %d: k %d n %d kind %d lds_w %d blk %d keep %d pva %d pf %d segs %d/%d\n", ch.cl[0], g.k, g.n, g.kind,
                                             g.lds_w, g.glb_k, g.keep, g.lds_pva, g.lds_acc_out, rt_fwd[id], rt_bwd[id]);
                                for (const ChainGenBody &b : rgb[ch.cl[0]])
                                    std::fprintf(stderr, "   body lam %d axisym %d acc_w %d up_w %d lds_va %d ind %d dep %d\n", b.lam, b.axisym, b.acc_w, b.up_w, b.lds_va, b.ind_a, b.dep_r);
                            }
                            rg[ch.cl[0]].first = static_cast<int>(R.gbodies.size());
                            R.gbodies.insert(R.gbodies.end(), rgb[ch.cl[0]].begin(), rgb[ch.cl[0]].end());
                            R.segs[rt_fwd[id]].first = R.segs[rt_bwd[id]].first = static_cast<int>(R.gens.size());
                            R.gens.push_back(rg[ch.cl[0]]);
                        }
                        if (ch.pair >= 0) {
                            R.segs[rt_pair[id]].first = static_cast<int>(R.pairs.size());
                            R.pairs.push_back(rp[ch.pair]);
                        }
                    }
                    for (int c = 0; c < nc; c++) {
                        if (cls[c] != 0) continue;
                        R.segs[rt_free_fwd[c]].first = R.segs[rt_free_bwd[c]].first = static_cast<int>(R.frees.size());
                        R.frees.push_back(rf[c]);
                    }
                    R.n_lds = rn_lds;
                    R.n_glb = rn_glb;
                    R.n_waves = rlm ? n_waves : 1;
                    if (lm && (!rlm || (!R.diffs.empty() && !lm_diffs) || !R.gens.empty())) rok = false;  // (latency mode: links, leaf pairs and -- fp32 -- differentials below one floating base)
                }
                R.ok = rok;
                R.single_gen = rok && R.gens.size() == 1 && R.segs.size() == 2 && R.links.empty() && R.pairs.empty() && R.frees.empty() &&
                               R.diffs.empty() && !R.gens[0].has_parent && R.n_glb == 0 && !std::getenv("GRBDA_NO_GEN1");
                if (!rok) { R.segs.clear(); R.links.clear(); R.pairs.clear(); R.frees.clear(); R.diffs.clear(); R.gens.clear(); R.gbodies.clear(); }
            }
            // ---- LDS objects and their live ranges ----
            // Time runs in half steps of the segment index: an ordinary object lives from the start of its first segment to the
            // end of its last one, [2 first, 2 last + 1].  An accumulator [IA | psi] is WRITTEN AT THE END of the first
            // backward segment below its body and READ AT THE START of the body's own backward segment, [2 first + 1, 2 own]:
            // the accumulator a segment consumes and the one it produces may share their slots.
            auto B0 = [](int t) { return 2 * t; };
            auto D1 = [](int t) { return 2 * t + 1; };
            for (int c = 0; c < nc; c++) {
                if (cls[c] != 0) continue;
                ChainFree &f = free_of[c];
                if (free_chains[c].empty()) continue;
                int first_bwd = 1 << 30, last_acc = t_free_acc[c];
                // (the [v, a] block is read when a kid chain's acceleration run STARTS: it lives until the last kid's own segment)
                for (int id : free_chains[c]) { first_bwd = std::min(first_bwd, ct[id].bwd); last_acc = std::max(last_acc, ct[id].acc); }
                objs.push_back({&f.lds_v, 6, 0, B0(t_free_fwd[c]), D1(t_free_bwd[c]), -1, 1});
                if (!lm) {
                    objs.push_back({&f.lds_acc, 27, 1, B0(first_bwd) + 1, B0(t_free_bwd[c]), -1, 1});
                } else {
                    // one accumulator per wavefront, in the global slab (two read-modify-writes per limb; LDS is what limits
                    // how many tiles a CU holds in this mode)
                    int32_t *const acc_of[4] = {&f.lds_acc, &f.lds_acc2, &f.lds_acc3, &f.lds_acc4};
                    for (int o = 0; o < n_waves && o < 4; o++) {
                        int fb = 1 << 30;
                        for (int id : free_chains[c])
                            if (owner_of[id] == o) fb = std::min(fb, ct[id].bwd);
                        if (fb == 1 << 30) continue;
                        // (in LDS the accumulator is alive from the base's forward segment on: the limbs of the other wavefronts run
                        // concurrently, whatever the order of the segments in the program says)
                        const bool acc_lds = k_lds;
                        objs.push_back({acc_of[o], 27, acc_lds ? 10 : 1, acc_lds ? B0(t_free_fwd[c]) : B0(fb) + 1, B0(t_free_bwd[c]), -1, acc_lds ? 0 : 2});
                    }
                }
                objs.push_back({&f.lds_va, 12, 0, B0(t_free_acc[c]), D1(last_acc), -1, 1});
                if (k_lds) objs.push_back({&f.glb_y0, 6, 11, B0(t_free_bwd[c]), D1(t_free_acc[c]), -1, 0});
            }
            int t_acc_phase = 1 << 30;  // first acceleration segment: objects born from there on live in phase 1
            for (int c = 0; c < nc; c++)
                if (cls[c] == 0) t_acc_phase = std::min(t_acc_phase, t_free_acc[c]);
            for (size_t id = 0; id < chains.size(); id++) {
                const Chain &ch = chains[id];
                const size_t o_begin = objs.size();
                struct OwnerTag {  // on leaving the iteration: the objects pushed for this chain belong to its wavefront
                    std::vector<Obj> &v; size_t from; int owner; int t_acc; bool on;
                    ~OwnerTag() { if (on) for (size_t i = from; i < v.size(); i++) { v[i].owner = owner; v[i].phase = v[i].birth / 2 >= t_acc ? 1 : 0; } }
                } owner_tag{objs, o_begin, owner_of[id], t_acc_phase, lm};
                if (ch.gen) {
                    const int c = ch.cl[0];
                    ChainGen &g = gen_of[c];
                    std::vector<ChainGenBody> &gb = gbody_of[c];
                    // (the forward segment's work area needs no in-cluster accumulators; the sizes are the same for simplicity)
                    objs.push_back({&g.lds_w, gen_w_size[c], 0, B0(ct[id].bwd), D1(ct[id].bwd), -1, 1, 1});
                    if (g.has_fwd) objs.push_back({&g.lds_wf, gen_w_size[c], 0, B0(ct[id].fwd), D1(ct[id].fwd), -1, 1, 1});
                    std::vector<int> va_until(g.k, -1);
                    for (int i = 0; i < g.k; i++) {
                        const int body = clusters[c].first_body + i;
                        int first_bwd = 1 << 30, last_acc = -1;
                        for (int k : ch.kid_chains)
                            if (clusters[chains[k].cl.front()].parent_body == body) {
                                first_bwd = std::min(first_bwd, ct[k].bwd);
                                last_acc = std::max(last_acc, ct[k].acc);
                            }
                        if (last_acc < 0) continue;
                        objs.push_back({&gb[i].lds_acc, 27, 1, B0(first_bwd) + 1, B0(ct[id].bwd), -1, 1});
                        objs.push_back({&gb[i].lds_v, 6, 0, B0(ct[id].fwd), D1(ct[id].bwd), -1, 1, 1});
                        // (v, a) of the body and of its in-cluster ancestors, until the last child segment below them has started
                        for (int j = i; j >= 0; j = gb[j].lam) va_until[j] = std::max(va_until[j], last_acc);
                    }
                    for (int i = 0; i < g.k; i++)
                        if (va_until[i] >= 0) {
                            objs.push_back({&gb[i].lds_va, 12, 0, B0(ct[id].acc), D1(va_until[i]), -1, 1});
                            g.need_acc = 1;
                        }
                    continue;
                }
                if (ch.diff) {
                    ChainDiff &d = diff_of[ch.cl[0]];
                    if (diff_shape[ch.cl[0]].n_atoms > 0)
                        objs.push_back({&d.lds_w, 3 * diff_shape[ch.cl[0]].n_atoms, 0, B0(ct[id].fwd), D1(ct[id].fwd), -1, 1, 1});
                    if (!ch.kid_chains.empty()) {
                        int first_bwd = 1 << 30, last_acc = ct[id].acc;
                        for (int k : ch.kid_chains) { first_bwd = std::min(first_bwd, ct[k].bwd); last_acc = std::max(last_acc, ct[k].acc); }
                        objs.push_back({&d.lds_sv, 8, 0, B0(ct[id].fwd), D1(ct[id].bwd), -1, 1});
                        objs.push_back({&d.lds_acc, 27, 1, B0(first_bwd) + 1, B0(ct[id].bwd), -1, 1});
                        objs.push_back({&d.lds_va, 12, 0, B0(ct[id].acc), D1(last_acc), -1, 1});
                    }
                    continue;
                }
                for (int c : ch.cl) objs.push_back({&link_of[c].lds_sv, 8, 0, B0(ct[id].fwd), D1(ct[id].bwd), -1, 1});
                if (k_lds) {
                    for (int c : ch.cl) {
                        objs.push_back({&link_of[c].glb_k, 7, 11, B0(ct[id].bwd), D1(ct[id].acc), -1, 0});
                        objs.back().span = 1;
                    }
                    if (ch.pair >= 0) {
                        objs.push_back({&pair_of[ch.pair].glb_k, 14, 11, B0(ct[id].bwd), D1(ct[id].pair_acc), -1, 0});
                        objs.back().span = 1;
                    }
                }
                const int tipc = ch.cl.back();
                if (!ch.kid_chains.empty()) {
                    int first_bwd = 1 << 30, last_acc = ct[id].acc;
                    for (int k : ch.kid_chains) { first_bwd = std::min(first_bwd, ct[k].bwd); last_acc = std::max(last_acc, ct[k].acc); }
                    objs.push_back({&acc_slot[tipc], 27, 1, B0(first_bwd) + 1, B0(ct[id].bwd), -1, 1});
                    objs.push_back({&link_of[tipc].lds_va, 12, 0, B0(ct[id].acc), D1(last_acc), -1, 1});
                } else if (ch.pair >= 0) {
                    objs.push_back({&link_of[tipc].lds_va, 12, 0, B0(ct[id].acc), D1(ct[id].pair_acc), -1, 1});
                }
            }
            int n_lds = 0, n_glb_unused = 0;
            const bool pre_ok = ok;
            ok = allocate_packed(objs, lds_budget, n_lds, n_glb_unused) && pre_ok;
            if (lm) {
                // no second tries in this mode (they move link blocks to the slab: latency on the critical path); the base's
                // accumulators are global by construction: number them behind the [K | y0] blocks
                for (Obj &o : objs)
                    if (o.slot >= 0 && (o.slot & kSlotGlobal)) *o.field = ((o.slot & ~kSlotGlobal) + n_glb) | kSlotGlobal;
                n_glb += n_glb_unused;
            }
            if (!ok && std::getenv("GRBDA_DEBUG_CHAIN")) {
                std::fprintf(stderr, "chain: LDS objects need more than %d slots (got to %d)\n", lds_budget, n_lds);
                for (const Obj &o : objs) std::fprintf(stderr, "  obj size %d [%d, %d] slot %d\n", o.size, o.birth, o.death, o.slot);
            }
            if (!ok && !lm) {
                // second try: the accumulators [IA 21][psi 6] of branching bodies -- touched once per child chain -- take
                // what LDS the other objects leave and otherwise move to the wave's global slab (their slot numbers then
                // carry kSlotGlobal)
                for (Obj &o : objs)
                    if (o.size == 27 && !o.tag) { o.force = 0; o.slot = -1; }
                ok = allocate(objs, lds_budget, n_lds, n_glb_unused);
                if (!ok) {
                    // third try (long chains, JVRC-1's arms and legs): the [sin, cos, v] blocks of the links likewise, after
                    // everything that must stay in LDS; the backward run fetches the next link's block while it computes the
                    // current one
                    for (Obj &o : objs)
                        if (o.size == 8 && !o.tag) { o.force = 0; o.prio = 1; o.slot = -1; }
                    ok = allocate(objs, lds_budget, n_lds, n_glb_unused);
                    CP.sv_global = true;
                    if (!ok && std::getenv("GRBDA_DEBUG_CHAIN")) {
                        std::fprintf(stderr, "chain: third try failed, n_lds %d\n", n_lds);
                        for (const Obj &o : objs) std::fprintf(stderr, "  obj size %d force %d [%d, %d] slot %d\n", o.size, o.force, o.birth, o.death, o.slot);
                    }
                }
                for (Obj &o : objs)
                    if (o.slot & kSlotGlobal) *o.field = ((o.slot & ~kSlotGlobal) + n_glb) | kSlotGlobal;
                n_glb += n_glb_unused;
            }
            if (ok) {
                // The result rows ydd[coordinate][lane] of the acceleration sweep go to LDS when nv free rows exist from the first
                // acceleration segment to the end of the tile (ChainProgram::out_lds; else to the slab: one more global round trip and
                // a drain of the store queue before the epilogue, 2.5-4 % of the kernel).  The epilogue transposes them through a
                // second block of nv rows -- rows [0, nv) when the result rows sit above them, else the rows behind the result rows;
                // nothing else is alive then.
                CP.out_lds = -1;
                {
                    int t_first = -1;
                    for (size_t t2 = 0; t2 < CP.segs.size(); t2++) {
                        const int op = CP.segs[t2].op;
                        if (op == SEG_FREE_ACC || op == SEG_RUN_ACC || op == SEG_PAIR_ACC || op == SEG_DIFF_ACC || op == SEG_GEN_ACC) { t_first = static_cast<int>(t2); break; }
                    }
                    const int nvr = P.nv;
                    if (t_first >= 0 && nvr > 0 && !std::getenv("GRBDA_NO_LDS_RESULTS")) {
                        std::vector<char> busy(static_cast<size_t>(lds_budget) + 1, 0);
                        for (const Obj &o : objs) {
                            if (o.slot < 0 || (o.slot & kSlotGlobal) || o.death < B0(t_first)) continue;
                            for (int r = o.slot; r < o.slot + o.size && r < lds_budget; r++) busy[r] = 1;
                        }
                        for (int r = 0; r + nvr <= lds_budget; r++) {
                            bool is_free = true;
                            for (int k2 = 0; k2 < nvr && is_free; k2++) is_free = !busy[r + k2];
                            if (!is_free) continue;
                            const int stage = r >= nvr ? 0 : r + nvr;
                            if (stage + nvr > lds_budget) continue;
                            CP.out_lds = r;
                            n_lds = std::max(n_lds, std::max(r + nvr, stage + nvr));
                            break;
                        }
                    }
                }
                if (std::getenv("GRBDA_DEBUG_CHAIN")) std::fprintf(stderr, "chain: result rows at LDS row %d (nv %d, %d rows in use of %d; %d slab rows, %d wavefronts per tile)\n", CP.out_lds, P.nv, n_lds, lds_budget, n_glb, CP.n_waves);
                CP.n_lds = n_lds;
                CP.n_glb = n_glb;
                // parent velocity / (v, a) slots
                auto v_slot_of_body = [&](int b) -> int {  // LDS slot of the velocity of body b (tip of its cluster); -1: ground
                    if (b < 0) return -1;
                    const int c = m.bodies[b].cluster;
                    if (is_diff(c)) return diff_of[c].lds_sv + 2;
                    if (cls[c] == 7) return gbody_of[c][b - clusters[c].first_body].lds_v;
                    return cls[c] == 0 ? free_of[c].lds_v : link_of[c].lds_sv + 2;  // (+2 keeps a kSlotGlobal flag intact)
                };
                auto va_slot_of_body = [&](int b) -> int {
                    if (b < 0) return -1;
                    const int c = m.bodies[b].cluster;
                    if (is_diff(c)) return diff_of[c].lds_va;
                    if (cls[c] == 7) return gbody_of[c][b - clusters[c].first_body].lds_va;
                    return cls[c] == 0 ? free_of[c].lds_va : link_of[c].lds_va;
                };
                auto acc_slot_of_body = [&](int b, int owner = 0) -> int {
                    if (b < 0) return -1;
                    const int c = m.bodies[b].cluster;
                    if (is_diff(c)) return diff_of[c].lds_acc;
                    if (cls[c] == 7) return gbody_of[c][b - clusters[c].first_body].lds_acc;
                    if (cls[c] != 0) return acc_slot[c];
                    const ChainFree &f = free_of[c];
                    return !lm || owner == 0 ? f.lds_acc : (owner == 1 ? f.lds_acc2 : (owner == 2 ? f.lds_acc3 : f.lds_acc4));
                };
                for (int c = 0; c < nc; c++) {
                    const int pb = clusters[c].parent_body;
                    if (cls[c] == 1 || cls[c] == 2 || cls[c] == 4) link_of[c].lds_pv = v_slot_of_body(pb);
                    if (cls[c] == 3) { pair_of[c].lds_pv = v_slot_of_body(pb); pair_of[c].lds_pva = va_slot_of_body(pb); }
                    if (is_diff(c)) {
                        // (a differential directly on the floating base of a latency-mode program adds into its wavefront's accumulator)
                        int own = 0;
                        for (size_t id2 = 0; id2 < chains.size(); id2++)
                            if (chains[id2].diff && chains[id2].cl[0] == c) own = owner_of[id2];
                        diff_of[c].lds_pv = v_slot_of_body(pb); diff_of[c].lds_pva = va_slot_of_body(pb); diff_of[c].lds_acc_out = acc_slot_of_body(pb, own);
                    }
                    if (cls[c] == 7) {
                        ChainGen &g = gen_of[c];
                        g.lds_pv = v_slot_of_body(pb); g.lds_pva = va_slot_of_body(pb); g.lds_acc_out = acc_slot_of_body(pb);
                        if (g.keep < -1) g.keep = g.lds_w + (-g.keep - 2);
                        for (ChainGenBody &b : gbody_of[c]) b.pva = b.lam >= 0 ? gbody_of[c][b.lam].lds_va : -1;
                    }
                }
                // first writer of every accumulator slot: the kid chain whose backward run comes first
                for (const RunRef &r : runs) {
                    ChainSeg &sg = CP.segs[r.seg];
                    sg.first = static_cast<int>(CP.links.size());
                    sg.count = static_cast<int>(r.cl.size());
                    bool all_axi = true, none = true;
                    for (int c : r.cl) {
                        CP.links.push_back(link_of[c]);
                        all_axi = all_axi && link_of[c].rofs >= 0 && link_of[c].rpre >= 0;
                        none = none && link_of[c].rofs < 0;
                    }
                    sg.rot_kind = all_axi ? 1 : (none ? 2 : 0);
                }
                for (size_t id = 0; id < chains.size(); id++) {
                    const Chain &ch = chains[id];
                    ChainSeg &bw = CP.segs[ct[id].bwd];
                    const int tipc = ch.cl.back(), topc = ch.cl.front();
                    if (ch.diff || ch.gen) bw.head = HEAD_LEAF;
                    if (bw.head == HEAD_SLOT) bw.head_arg = acc_slot[tipc];
                    if (bw.head == HEAD_PAIR) { bw.head_arg = static_cast<int>(CP.pairs.size()); }
                    const int pb = clusters[topc].parent_body;
                    bw.lds_acc_out = acc_slot_of_body(pb, owner_of[id]);
                    // first writer: earliest backward run among the sibling chains
                    bool first = true;
                    if (pb >= 0) {
                        // the sibling chains: those that hang off the same BODY (a generic cluster carries children on several)
                        const int pc = m.bodies[pb].cluster;
                        for (size_t o = 0; o < chains.size(); o++)
                            if (o != id && clusters[chains[o].cl.front()].parent_body == pb && ct[o].bwd < ct[id].bwd &&
                                (!lm || cls[pc] != 0 || owner_of[o] == owner_of[id]))
                                first = false;
                    }
                    bw.acc_first = first ? 1 : 0;
                    ChainSeg &ac = CP.segs[ct[id].acc];
                    ac.lds_pva = va_slot_of_body(pb);
                    if (ch.diff) {
                        diff_of[topc].acc_first = first ? 1 : 0;
                        CP.segs[ct[id].fwd].first = bw.first = ac.first = static_cast<int>(CP.diffs.size());
                        CP.diffs.push_back(diff_of[topc]);
                    }
                    if (ch.gen) {
                        gen_of[topc].acc_first = first ? 1 : 0;
                        gen_of[topc].first = static_cast<int>(CP.gbodies.size());
                        CP.gbodies.insert(CP.gbodies.end(), gbody_of[topc].begin(), gbody_of[topc].end());
                        CP.segs[ct[id].fwd].first = bw.first = ac.first = static_cast<int>(CP.gens.size());
                        CP.gens.push_back(gen_of[topc]);
                    }
                    if (ch.pair >= 0) {
                        ChainSeg &pa = CP.segs[ct[id].pair_acc];
                        pa.first = static_cast<int>(CP.pairs.size());
                        CP.pairs.push_back(pair_of[ch.pair]);
                    }
                }
                for (int c = 0; c < nc; c++) {
                    if (cls[c] != 0) continue;
                    const int fi = static_cast<int>(CP.frees.size());
                    CP.frees.push_back(free_of[c]);
                    CP.segs[t_free_fwd[c]].first = CP.segs[t_free_bwd[c]].first = CP.segs[t_free_acc[c]].first = fi;
                }
            }
        }
        CP.ok = ok;
        CP.single_gen = ok && CP.gens.size() == 1 && CP.segs.size() == 2 && CP.links.empty() && CP.pairs.empty() && CP.frees.empty() &&
                        CP.diffs.empty() && !CP.gens[0].has_parent && !std::getenv("GRBDA_NO_GEN1");
        if (!ok) { CP.segs.clear(); CP.links.clear(); CP.pairs.clear(); CP.frees.clear(); CP.diffs.clear(); CP.gens.clear(); CP.gbodies.clear(); }
    };
    // the RNEA chain kernels run 8 wavefronts per CU like the ABA ones: the ABA budgets apply
    build_chain(P.chain32, lds.aba32, &P.rchain32, lds.aba32);
    build_chain(P.chain32w, lds.chain32w, &P.rchain32w, lds.chain32w);
    build_chain(P.chain64, lds.aba64, &P.rchain64, lds.aba64);
    // Generic clusters (plan.h, ChainGen) keep their work area in LDS: large ones get a program laid out for twice the LDS per
    // wavefront (the launch then holds fewer wavefronts per CU, capi.cpp: still several times the interpreter's rate)
    if (!P.chain32.ok) build_chain(P.chain32, 2 * lds.aba32, P.rchain32.ok ? nullptr : &P.rchain32, 2 * lds.aba32);
    if (!P.chain64.ok) build_chain(P.chain64, 2 * lds.aba64, P.rchain64.ok ? nullptr : &P.rchain64, 2 * lds.aba64);
    // (the same for an inverse-dynamics program that did not fit: its generic clusters keep [f | sin, cos] of every body from the
    // forward to the backward segment)
    {
        ChainProgram scratch_cp;
        if (!P.rchain32.ok) build_chain(scratch_cp, 2 * lds.aba32, &P.rchain32, 2 * lds.aba32);
        if (!P.rchain64.ok) build_chain(scratch_cp, 2 * lds.aba64, &P.rchain64, 2 * lds.aba64);
    }
    // latency mode serves batches of at most one tile per SIMD, i.e. four tiles per CU: 40 KiB of LDS per tile
    build_chain(P.chain32p, 40960 / (4 * kWave), &P.rchain32p, 40960 / (4 * kWave), 2, std::getenv("GRBDA_LM2_SLAB") == nullptr, true);  // (A/B switch: blocks in the slab)
    build_chain(P.chain64p, 40960 / (8 * kWave), &P.rchain64p, 40960 / (8 * kWave), 2);
    // four wavefronts per tile: batches of at most two tiles per CU (one wavefront per SIMD in the two-wavefront mode), 80 KiB each
    build_chain(P.chain32q, 81920 / (4 * kWave), &P.rchain32q, 81920 / (4 * kWave), 4, true, true);
    build_chain(P.chain64q, 81920 / (8 * kWave), &P.rchain64q, 81920 / (8 * kWave), 4, true);

    // ---- composite-rigid-body program (crba_kernels.hip) ----------------------------------------------------------
    {
        CrbaProgram &CR = P.crba;
        CR = CrbaProgram();
        CR.ok = sweep_mask == 7;
        for (const ClusterRec &cr : clusters)
            if (cr.kind == CK_LOOP) CR.ok = false;
        CR.bodies.assign(nb, CrbaBody{0, -1, 0, 0});
        int rows = 0;
        for (int b = 0; b < nb; b++) {
            CR.bodies[b].cluster = m.bodies[b].cluster;
            CR.bodies[b].sc_row = rows;
            rows += 2;
        }
        for (int b = 0; b < nb; b++)
            if (bodies[b].parent >= 0 && !bodies[b].axisym && CR.bodies[bodies[b].parent].acc_row < 0) {
                CR.bodies[bodies[b].parent].acc_row = rows;
                rows += 21;
            }
        CR.n_rows = rows;
    }

    // ---- inverse-dynamics derivative program (deriv_kernels.hip) -----------------------------------------------------
    {
        DerivProgram &DV = P.deriv;
        DV = DerivProgram();
        DV.ok = sweep_mask == 7;
        for (const ClusterRec &cr : clusters) {
            if (cr.kind == CK_LOOP) DV.ok = false;
            if (cr.kind == CK_FREE && cr.first_body != 0) DV.ok = false;
        }
        // the recursion transforms body inertias with the rigid-body congruence (devmath.h, congruence_rigid): [[Ibar, h^], [h^T, m 1]] -- what a
        // SpatialInertia of the reference holds (SpatialInertia.h: mass, centre of mass, rotational inertia); a description with any other
        // symmetric 6 x 6 keeps the difference batches
        for (int b = 0; b < nb && DV.ok; b++) {
            const double *I = m.bodies[b].inertia;
            double scale = 0;
            for (int i = 0; i < 36; i++) scale = std::max(scale, std::fabs(I[i]));
            const double tol = 1e-10 * (scale + 1e-300), mass = I[3 * 6 + 3];
            for (int i = 0; i < 3 && DV.ok; i++)
                for (int j = 0; j < 3; j++) {
                    if (std::fabs(I[(3 + i) * 6 + 3 + j] - (i == j ? mass : 0.0)) > tol) DV.ok = false;   // lower right: m 1
                    if (std::fabs(I[i * 6 + 3 + j] + I[j * 6 + 3 + i]) > tol) DV.ok = false;               // upper right: skew
                }
        }
        DV.bodies.assign(nb, DerivBody{0, -1, -1, -1, 0, 0, 0, -1, -1, 0});
        int rows = 0;
        for (int b = 0; b < nb; b++) {
            DV.bodies[b].cluster = m.bodies[b].cluster;
            if (!bodies[b].has_child) continue;
            DV.bodies[b].kin_row = rows;
            rows += 24;
            if (bodies[b].jtype != GRBDA_JOINT_FREE) {
                DV.bodies[b].anc_row = rows;
                rows += 18;
            }
        }
        for (const ClusterRec &cr : clusters)
            if (cr.kind != CK_FREE) DV.n_max = std::max(DV.n_max, cr.n);
        {
            // LDS cache of the ancestor rows (plan.h, kDerivAncLevels): the walk of cluster c visits the body-tree ancestors of its
            // parent body up to the base; simulate the blocks over the kernel's processing order (last cluster first)
            std::vector<int> level(nb, 0);
            for (int b = 0; b < nb; b++) {
                const int pb = bodies[b].parent;
                level[b] = (pb >= 0 && bodies[pb].jtype != GRBDA_JOINT_FREE) ? level[pb] + 1 : 0;
                if (DV.bodies[b].anc_row >= 0 && level[b] < kDerivAncLevels) DV.bodies[b].anc_lds = level[b];
            }
            int owner[kDerivAncLevels];
            for (int &o : owner) o = -1;
            for (int c = nc - 1; c >= 0; c--) {
                const ClusterRec &cr = clusters[c];
                if (cr.kind == CK_FREE) continue;
                int mask = 0;
                for (int b = cr.parent_body; b >= 0 && bodies[b].jtype != GRBDA_JOINT_FREE; b = bodies[b].parent) {
                    const int l = DV.bodies[b].anc_lds;
                    if (l < 0) continue;
                    if (owner[l] == b) mask |= 1 << l;
                    else owner[l] = b;
                }
                DV.bodies[cr.first_body].walk_resident = mask;
            }
        }
        // register hand-over along chains: a cluster that is the only contributor to its parent body, directly before that
        // body's cluster in the processing order, leaves its composites in registers; the receiving body (no in-cluster
        // children) is processed first in its cluster
        std::vector<char> carried(nb, 0);
        for (int c = nc - 1; c >= 1; c--) {
            const ClusterRec &cr = clusters[c];
            const int pb = cr.parent_body;
            if (pb < 0 || m.bodies[pb].cluster != c - 1 || bodies[pb].jtype == GRBDA_JOINT_FREE || bodies[pb].lam >= 0) continue;
            int contributors = 0;
            for (int c2 = 0; c2 < nc; c2++)
                if (clusters[c2].parent_body == pb) contributors++;
            for (int b = 0; b < nb; b++)
                if (bodies[b].lam == pb) contributors += 2;
            if (contributors != 1) continue;
            DV.bodies[cr.first_body].carry_out = 1;
            DV.bodies[clusters[c - 1].first_body].carry_body = pb;
            carried[pb] = 1;
        }
        // accumulators: written by the child clusters (one combined write each, highest cluster first) and by in-cluster
        // children, read when the body itself is processed; rows are shared between accumulators that are never live together
        std::vector<int> birth(nb, -1), death(nb, -1);
        int step = 0;
        for (int c = nc - 1; c >= 0; c--) {
            const ClusterRec &cr = clusters[c];
            for (int i = cr.k - 1; i >= 0; i--, step++) {
                const int b = cr.first_body + i;
                death[b] = step;
                if (bodies[b].lam >= 0 && birth[bodies[b].lam] < 0) {
                    birth[bodies[b].lam] = step;
                    DV.bodies[b].acc_first = 1;
                }
            }
            if (cr.parent_body >= 0 && birth[cr.parent_body] < 0) {
                birth[cr.parent_body] = step - 1;
                DV.bodies[cr.first_body].cluster_acc_first = 1;
            }
        }
        std::vector<Obj> aobjs;
        for (int b = 0; b < nb; b++)
            if (bodies[b].has_child && !carried[b]) aobjs.push_back({&DV.bodies[b].acc_row, 63, 0, birth[b], death[b], -1, 1});
        int n_acc = 0, n_unused = 0;
        allocate(aobjs, 1 << 28, n_acc, n_unused);
        for (int b = 0; b < nb; b++)
            if (DV.bodies[b].acc_row >= 0) DV.bodies[b].acc_row += rows;
        DV.n_rows = rows + n_acc;
        if (P.nv <= 64) {
            DV.related.assign(P.nv, 0);
            for (int c = 0; c < nc; c++) {
                int a = c;  // c and its ancestor clusters
                for (;;) {
                    for (int i = 0; i < clusters[c].n; i++)
                        for (int j = 0; j < clusters[a].n; j++) {
                            DV.related[clusters[c].v_index + i] |= uint64_t(1) << (clusters[a].v_index + j);
                            DV.related[clusters[a].v_index + j] |= uint64_t(1) << (clusters[c].v_index + i);
                        }
                    if (clusters[a].parent_body < 0) break;
                    a = m.bodies[clusters[a].parent_body].cluster;
                }
            }
        }
        // ---- H^-1 = W^T W from the articulated-body quantities (plan.h, MinvProgram; minv_kernels.hip) ----
        {
            MinvProgram &MV = DV.minv;
            MV = MinvProgram();
            MV.ok = DV.ok && P.nv <= 64;
            MV.bodies.assign(nb, MinvBody{-1, -1, 0});
            int off = 0;
            std::vector<int> c_off(nc, -1);
            for (int c = 0; c < nc && MV.ok; c++) {
                const ClusterRec &cr = clusters[c];
                if (cr.kind == CK_FREE) {
                    MV.base_off = off;
                    MV.bodies[cr.first_body].clus_off = off;
                    c_off[c] = off;
                    off += 21;
                    continue;
                }
                if (cr.n > kMaxClusterDof) { MV.ok = false; break; }
                const int blk = 6 * cr.n + cr.n * (cr.n + 1) / 2;
                c_off[c] = off;
                MV.bodies[cr.first_body].clus_off = off;
                bool any = false;
                for (int i = 0; i < cr.k; i++) {
                    const int b = cr.first_body + i;
                    bool carries = false;
                    for (int c2 = 0; c2 < nc; c2++) carries = carries || clusters[c2].parent_body == b;
                    if (carries) {
                        MV.bodies[b].blk_off = off;
                        off += blk + 6 * cr.n;
                        any = true;
                    }
                }
                if (!any) off += blk;
            }
            MV.n_entries = off;
            if (off >= 65536) MV.ok = false;
            // which kinematics rows the factor kernel really stores (plan.h, MinvBody::keep): simulate its two passes
            {
                int last = -1;  // pass 1, clusters root side first, bodies with children: the parent comes from registers when it was the body before
                for (int c = 0; c < nc; c++) {
                    const ClusterRec &cr = clusters[c];
                    if (cr.kind == CK_FREE) { last = cr.first_body; continue; }
                    for (int i = 0; i < cr.k; i++) {
                        const int b = cr.first_body + i;
                        if (!bodies[b].has_child) continue;
                        if (bodies[b].parent >= 0 && bodies[b].parent != last) MV.bodies[bodies[b].parent].keep = 1;
                        last = b;
                    }
                }
                for (int b = 0; b < nb; b++)
                    if (bodies[b].lam >= 0) MV.bodies[bodies[b].lam].keep = 1;  // in-cluster parents: their children and the ancestor loops load them
                for (int c = 0; c < nc; c++) {  // pass 2: clusters without a carried hand-over load their parent body's row
                    const ClusterRec &cr = clusters[c];
                    if (cr.kind == CK_FREE || cr.parent_body < 0) continue;
                    if (DV.bodies[cr.first_body].carry_body < 0) MV.bodies[cr.parent_body].keep = 1;
                }
            }
            MV.coltab.assign(static_cast<size_t>(64) * kMinvColInts, 0);
            for (int c = 0; c < nc && MV.ok; c++) {
                const ClusterRec &cr = clusters[c];
                const bool is_free = cr.kind == CK_FREE;
                for (int e = 0; e < cr.n; e++) {
                    int32_t *col = &MV.coltab[static_cast<size_t>(cr.v_index + e) * kMinvColInts];
                    col[0] = is_free ? 0 : c_off[c] + 6 * e;
                    col[1] = is_free ? c_off[c] : c_off[c] + 6 * cr.n;
                    int depth = 0, pb = cr.parent_body;
                    bool at_base = false;
                    while (pb >= 0) {
                        const int a = m.bodies[pb].cluster;
                        if (clusters[a].kind == CK_FREE) { at_base = true; break; }
                        if (depth >= kMinvMaxDepth || MV.bodies[pb].blk_off < 0) { MV.ok = false; break; }
                        col[3 + depth] = static_cast<int32_t>(static_cast<uint32_t>(MV.bodies[pb].blk_off) | (static_cast<uint32_t>(clusters[a].n) << 16) |
                                                              (static_cast<uint32_t>(clusters[a].v_index) << 20) | (1u << 31));
                        depth++;
                        pb = clusters[a].parent_body;
                    }
                    MV.max_depth = std::max(MV.max_depth, depth);
                    col[2] = cr.n | (e << 4) | (cr.v_index << 8) | ((at_base ? 1 : 0) << 20) | (1 << 21);
                }
            }
        }
    }

    // ---- operation count (mul + add, as executed by kernels.hip) --------------------------------
    // per-body costs: sincos ~40, E build 12, motion / force transform 39, sym6*vec 66 (48 against a
    // revolute velocity product, two zero entries), force cross 30, congruence 385, plus the per-cluster
    // solve terms.  Axisymmetric rotors are evaluated at q = 0 (no sincos, no congruence).  Checked against
    // the executed SQ_INSTS_VALU_{FMA,MUL,ADD}_F32 counts (profiles/): 23.7e3 flop per MIT-humanoid evaluation.
    double fa = 0, fr = 0;
    for (int c = 0; c < nc; c++) {
        const ClusterRec &cr = clusters[c];
        const int n = cr.n;
        for (int i = 0; i < cr.k; i++) {
            const BodyRec &br = bodies[cr.first_body + i];
            const bool is_free = br.jtype == GRBDA_JOINT_FREE;
            if (is_free) {
                fa += 60 + 66 + 30 + 21 + 6;                       // E from the quaternion, Iv, cross, IA, u
                fr += 60 + 39 + 66 * 2 + 30 + 6;
                continue;
            }
            const double kin = br.axisym ? 39 + 2 * n : 40 + 12 + 39 + 2 * n;  // (sincos, E,) v, qd
            fa += kin + 4 + 66 + 30 + 21 + 9 + 39 + 6 * n * 2 + n * n * 2;    // c, Iv, cross, IA, b, F, D
            if (br.parent >= 0) fa += 48 + 6 + 39 + (br.axisym ? 0 : 385);    // IA c + pA, to the parent, X^T IA X
            if (br.lam >= 0) fa += 39 + 4 * n * n;                            // push up the in-cluster chain
            if (br.has_child) fa += 40 + 12 + 39 * 2 + 4 + 2 * n + 6;          // acceleration sweep (recomputes v)
            fr += kin + 39 + 4 + 2 * n + 66 * 2 + 30 + 6 + 39 + 2 * n;        // v, a, f = I a + v x* I v, tau, X^T f
        }
        if (cr.kind == CK_FREE) fa += 6 * 6 * 6 / 3.0 + 2.0 * 36 + 6;        // 6x6 Cholesky + solve
        else fa += n * n * n / 3.0 + 2.0 * n * n * 7 + 21 * 2 * n + 12 * n + 12 * n;  // solve, K, IA -= F K, psi += F y0, ydd
    }
    P.flops_aba = fa;
    P.flops_rnea = fr;
    return 0;
}

}  // namespace grbda_hip

// ---------------------------------------------------------------------------------------------------------------
// Spanning-tree model of a cluster tree model: every revolute body a cluster of its own (Revolute, G = 1), the floating base
// as it is.  The analytic derivatives of models with implicit clusters are taken on it (capi.cpp, manifold_derivs;
// manifold_kernels.hip): tau = G^T tau_span(q_span, G yd, G ydd + g), so d tau / d y needs d tau_span / d (q, qd)_span and
// H_span of the spanning tree -- which is an explicit model the inverse-dynamics derivative recursion covers -- and the
// derivatives of G and g.  span_q / span_v: per body of the model, its first position / velocity index in the spanning model.
// ---------------------------------------------------------------------------------------------------------------
namespace grbda_hip {
int make_spanning_blob(const void *blob, size_t bytes, std::vector<unsigned char> &out, std::vector<int32_t> &span_q,
                       std::vector<int32_t> &span_v, char *msg, size_t cap)
{
    Blob m;
    if (int rc = parse(blob, bytes, m, msg, cap)) return rc;
    const int nb = m.h->n_bodies;
    grbda_desc_header h = *m.h;
    std::vector<grbda_desc_body> bodies(m.bodies, m.bodies + nb);
    std::vector<grbda_desc_cluster> clusters(nb);
    std::vector<double> dbls;
    span_q.assign(nb, 0);
    span_v.assign(nb, 0);
    int nq = 0, nv = 0;
    for (int b = 0; b < nb; b++) {
        grbda_desc_cluster &c = clusters[b];
        std::memset(&c, 0, sizeof c);
        const bool is_free = bodies[b].joint_type == GRBDA_JOINT_FREE;
        c.parent_cluster = bodies[b].parent;  // one cluster per body: cluster index = body index
        c.first_body = b;
        c.n_bodies = 1;
        c.q_index = nq;
        c.v_index = nv;
        c.n_pos = is_free ? (h.ori_repr == GRBDA_ORI_QUATERNION ? 7 : 6) : 1;
        c.n_vel = is_free ? 6 : 1;
        c.n_span_pos = c.n_pos;
        c.n_span_vel = c.n_vel;
        c.constraint_type = is_free ? GRBDA_CONSTRAINT_FREE : GRBDA_CONSTRAINT_STATIC;
        c.n_constraint_rows = 0;
        c.int_offset = 0;
        c.n_int = 0;
        c.dbl_offset = static_cast<int32_t>(dbls.size());
        c.n_dbl = is_free ? 0 : 1;
        if (!is_free) dbls.push_back(1.0);
        span_q[b] = nq;
        span_v[b] = nv;
        nq += c.n_pos;
        nv += c.n_vel;
        bodies[b].cluster = b;
        bodies[b].sub_index = 0;
    }
    h.n_clusters = nb;
    h.nq = nq;
    h.nv = nv;
    h.n_ints = 0;
    h.n_doubles = static_cast<int32_t>(dbls.size());
    h.n_name_bytes = 0;
    out.clear();
    auto put = [&](const void *p, size_t n) { const auto *c = static_cast<const unsigned char *>(p); out.insert(out.end(), c, c + n); };
    put(&h, sizeof h);
    put(bodies.data(), sizeof(grbda_desc_body) * bodies.size());
    put(clusters.data(), sizeof(grbda_desc_cluster) * clusters.size());
    put(dbls.data(), sizeof(double) * dbls.size());
    return 0;
}


}  // namespace grbda_hip
