/*
 * grbda_oracle.c -- TEST INFRASTRUCTURE ONLY (see grbda_oracle.h).
 *
 * Dense, single-state-at-a-time restatement (fp64 as the parity checker; see grbda_oracle.h for the fp32 build)
 *  of the reference's cluster-ABA / RNEA.
 * It deliberately follows the reference's *dense* formulation (6k x 6k cluster inertias,
 * X_intra / X_intra_ring matrices, GeneralizedTransform block loops), not the structured
 * per-body formulation the HIP product uses, so the two are independent derivations.
 *
 * Reference lines each function follows are cited next to it; paths are relative to
 * /root/reference.
 */
#include "grbda_oracle.h"
#include "../include/grbda_model_desc.h"

typedef grbda_real real;  /* working precision; the model description stays double (md_copy) */

#include <tgmath.h>  /* type-generic sin / cos / sqrt / fabs: the working precision is a build parameter (grbda_oracle.h) */
#undef I             /* (tgmath.h brings complex.h and its imaginary unit: `I` is an inertia here) */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* (_build/libgrbda_oracle_big.so is this file with -DMAXK=48 -DMAXN=48: the checker of the clusters beyond the HIP kernels'
 * structured limits -- the per-cluster workspace below grows with (6 MAXK)^2, so the default build stays small) */
#ifndef MAXK
#define MAXK 8            /* bodies per cluster */
#endif
#define MAXD (6 * MAXK)   /* motion-subspace dimension of a cluster */
#ifndef MAXN
#define MAXN 8            /* independent velocities per cluster */
#endif
#define MAXSP (MAXK + 6)  /* spanning positions per cluster */
#ifndef MAXROWS
#define MAXROWS 8         /* constraint rows per cluster */
#endif

/* ------------------------------------------------------------------------------------------ */
/* small dense helpers (row-major)                                                            */
/* ------------------------------------------------------------------------------------------ */
/* model-description doubles -> working precision */
static void md_copy(real *dst, const double *src, int n)
{
    for (int i = 0; i < n; i++) dst[i] = (real)src[i];
}
static void mm(const real *A, const real *B, real *C, int m, int k, int n)
{ /* C[m x n] = A[m x k] B[k x n] */
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            real s = 0;
            for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static void mtm(const real *A, const real *B, real *C, int m, int k, int n)
{ /* C[k x n] = A^T B, A is m x k, B is m x n */
    for (int i = 0; i < k; i++)
        for (int j = 0; j < n; j++) {
            real s = 0;
            for (int l = 0; l < m; l++) s += A[l * k + i] * B[l * n + j];
            C[i * n + j] = s;
        }
}
static void m3mul(const real *A, const real *B, real *C) { mm(A, B, C, 3, 3, 3); }
static void m3tv(const real *A, const real *v, real *o)
{ /* o = A^T v */
    for (int i = 0; i < 3; i++) o[i] = A[0 + i] * v[0] + A[3 + i] * v[1] + A[6 + i] * v[2];
}
static void m3v(const real *A, const real *v, real *o)
{
    for (int i = 0; i < 3; i++) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
static void cross3(const real *a, const real *b, real *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* LU with partial pivoting, solves A X = Bm in place (A n x n, Bm n x m).  The reference uses
 * Eigen::ColPivHouseholderQR (include/grbda/Utils/Utilities.h:325-329); any backward-stable
 * solve agrees to rounding. */
static int lu_solve(real *A, real *Bm, int n, int m)
{
    for (int c = 0; c < n; c++) {
        int p = c;
        real best = fabs(A[c * n + c]);
        for (int r = c + 1; r < n; r++)
            if (fabs(A[r * n + c]) > best) { best = fabs(A[r * n + c]); p = r; }
        if (best == 0.0 || best != best) return GRBDA_ORACLE_ESINGULAR;
        if (p != c) {
            for (int j = 0; j < n; j++) { real t = A[c * n + j]; A[c * n + j] = A[p * n + j]; A[p * n + j] = t; }
            for (int j = 0; j < m; j++) { real t = Bm[c * m + j]; Bm[c * m + j] = Bm[p * m + j]; Bm[p * m + j] = t; }
        }
        for (int r = c + 1; r < n; r++) {
            real f = A[r * n + c] / A[c * n + c];
            if (f == 0.0) continue;
            for (int j = c; j < n; j++) A[r * n + j] -= f * A[c * n + j];
            for (int j = 0; j < m; j++) Bm[r * m + j] -= f * Bm[c * m + j];
        }
    }
    for (int c = n - 1; c >= 0; c--)
        for (int j = 0; j < m; j++) {
            real s = Bm[c * m + j];
            for (int l = c + 1; l < n; l++) s -= A[c * n + l] * Bm[l * m + j];
            Bm[c * m + j] = s / A[c * n + c];
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* spatial::Transform (src/Utils/SpatialTransforms.cpp:13-197)                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct { real E[9]; real r[3]; } xf_t;

static void xf_identity(xf_t *X)
{
    memset(X, 0, sizeof(*X));
    X->E[0] = X->E[4] = X->E[8] = 1.0;
}
/* operator* (SpatialTransforms.cpp:149-157): E = E1 E2, r = r2 + E2^T r1 */
static void xf_mul(const xf_t *A, const xf_t *B, xf_t *C)
{
    xf_t o;
    m3mul(A->E, B->E, o.E);
    real t[3];
    m3tv(B->E, A->r, t);
    for (int i = 0; i < 3; i++) o.r[i] = B->r[i] + t[i];
    *C = o;
}
/* toMatrix (SpatialTransforms.cpp:32-40): [[E,0],[-E r^, E]] */
static void xf_matrix(const xf_t *X, real *M)
{
    const real *E = X->E, *r = X->r;
    real rh[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    real Er[9];
    m3mul(E, rh, Er);
    memset(M, 0, 36 * sizeof(real));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            M[i * 6 + j] = E[i * 3 + j];
            M[(i + 3) * 6 + j + 3] = E[i * 3 + j];
            M[(i + 3) * 6 + j] = -Er[i * 3 + j];
        }
}
/* transformMotionVector (SpatialTransforms.cpp:42-50) */
static void xf_motion(const xf_t *X, const real *m, real *o)
{
    real t[3], c[3];
    m3v(X->E, m, o);
    cross3(X->r, m, c);
    for (int i = 0; i < 3; i++) t[i] = m[3 + i] - c[i];
    m3v(X->E, t, o + 3);
}
/* inverseTransformForceVector (SpatialTransforms.cpp:73-82) */
static void xf_inv_force(const xf_t *X, const real *f, real *o)
{
    real n[3], l[3], c[3];
    m3tv(X->E, f, n);
    m3tv(X->E, f + 3, l);
    cross3(X->r, l, c);
    for (int i = 0; i < 3; i++) { o[i] = n[i] + c[i]; o[3 + i] = l[i]; }
}
/* transformForceVector (SpatialTransforms.cpp:62-71) */
static void xf_force(const xf_t *X, const real *f, real *o)
{
    real c[3], t[3];
    cross3(X->r, f + 3, c);
    for (int i = 0; i < 3; i++) t[i] = f[i] - c[i];
    m3v(X->E, t, o);
    m3v(X->E, f + 3, o + 3);
}
/* motionCrossProduct / forceCrossProduct (include/grbda/Utils/Spatial.h:131-143,177-188) */
static void crm(const real *a, const real *b, real *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
    o[3] = a[1] * b[5] - a[2] * b[4] + a[4] * b[2] - a[5] * b[1];
    o[4] = a[2] * b[3] - a[0] * b[5] - a[3] * b[2] + a[5] * b[0];
    o[5] = a[0] * b[4] - a[1] * b[3] + a[3] * b[1] - a[4] * b[0];
}
static void crf(const real *a, const real *b, real *o)
{
    o[0] = b[2] * a[1] - b[1] * a[2] - b[4] * a[5] + b[5] * a[4];
    o[1] = b[0] * a[2] - b[2] * a[0] + b[3] * a[5] - b[5] * a[3];
    o[2] = b[1] * a[0] - b[0] * a[1] - b[3] * a[4] + b[4] * a[3];
    o[3] = b[5] * a[1] - b[4] * a[2];
    o[4] = b[3] * a[2] - b[5] * a[0];
    o[5] = b[4] * a[0] - b[3] * a[1];
}
/* motionCrossMatrix (Spatial.h:54-66) */
static void crm_matrix(const real *v, real *M)
{
    memset(M, 0, 36 * sizeof(real));
    real w[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
    real l[9] = {0, -v[5], v[4], v[5], 0, -v[3], -v[4], v[3], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            M[i * 6 + j] = w[i * 3 + j];
            M[(i + 3) * 6 + j + 3] = w[i * 3 + j];
            M[(i + 3) * 6 + j] = l[i * 3 + j];
        }
}
/* ori::coordinateRotation (include/grbda/Utils/OrientationTools.h:46-68) */
static void coord_rotation(int axis, real th, real *R)
{
    real s = sin(th), c = cos(th);
    if (axis == 0) { real t[9] = {1, 0, 0, 0, c, s, 0, -s, c}; memcpy(R, t, sizeof t); }
    else if (axis == 1) { real t[9] = {c, 0, -s, 0, 1, 0, s, 0, c}; memcpy(R, t, sizeof t); }
    else { real t[9] = {c, s, 0, -s, c, 0, 0, 0, 1}; memcpy(R, t, sizeof t); }
}
/* ori::quaternionToRotationMatrix (OrientationTools.h:251-269): scalar first, result transposed */
static void quat_to_rot(const real *q, real *R)
{
    real e0 = q[0], e1 = q[1], e2 = q[2], e3 = q[3];
    real M[9] = {1 - 2 * (e2 * e2 + e3 * e3), 2 * (e1 * e2 - e0 * e3), 2 * (e1 * e3 + e0 * e2),
                   2 * (e1 * e2 + e0 * e3), 1 - 2 * (e1 * e1 + e3 * e3), 2 * (e2 * e3 - e0 * e1),
                   2 * (e1 * e3 - e0 * e2), 2 * (e2 * e3 + e0 * e1), 1 - 2 * (e1 * e1 + e2 * e2)};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[i * 3 + j] = M[j * 3 + i];
}
/* ori::rpyToRotMat (OrientationTools.h:121-130): Rx Ry Rz coordinate rotations */
static void rpy_to_rot(const real *rpy, real *R)
{
    real Rx[9], Ry[9], Rz[9], T[9];
    coord_rotation(0, rpy[0], Rx);
    coord_rotation(1, rpy[1], Ry);
    coord_rotation(2, rpy[2], Rz);
    m3mul(Rx, Ry, T);
    m3mul(T, Rz, R);
}

/* ------------------------------------------------------------------------------------------ */
/* model + per-state workspace                                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const grbda_desc_header *h;
    const grbda_desc_body *bodies;
    const grbda_desc_cluster *clusters;
    const int32_t *ints;
    const double *dbls;
} model_t;

typedef struct {
    int k, n, dim, nsp, nsv, rows;
    /* ClusterJoints::Base state (ClusterJoint.h:33-98) */
    real qs[MAXSP], qds[MAXD];       /* spanning position / velocity */
    real G[MAXD * MAXN], g[MAXD];    /* nsv x n, nsv */
    real K[MAXROWS * MAXD], kb[MAXROWS], phi[MAXROWS];
    real S[MAXD * MAXN], vJ[MAXD], cJ[MAXD];
    xf_t XJ[MAXK];  /* single-joint transform * Xtree */
    /* TreeNode / ClusterTreeNode state (TreeNode.h:48-74, ClusterTreeNode.h:39-55) */
    xf_t Xup[MAXK], Xa[MAXK];
    int anc_sub[MAXK]; /* sub-index (in parent cluster) of the nearest ancestor outside the cluster */
    int lam[MAXK];     /* in-cluster tree parent sub-index or -1 */
    real v[MAXD], a[MAXD], avp[MAXD], f[MAXD], fext[MAXD];
    real I[MAXD * MAXD], IA[MAXD * MAXD], Ia[MAXD * MAXD];
    real pA[MAXD], U[MAXD * MAXN], DinvUT[MAXN * MAXD], u[MAXN], Dinvu[MAXN];
    int has_ext;
} cws_t;

static int parse_blob(const void *blob, size_t bytes, model_t *m)
{
    if (!blob || bytes < sizeof(grbda_desc_header)) return GRBDA_ORACLE_EBADBLOB;
    const grbda_desc_header *h = (const grbda_desc_header *)blob;
    if (h->magic != GRBDA_DESC_MAGIC || h->version != GRBDA_DESC_VERSION) return GRBDA_ORACLE_EBADBLOB;
    size_t off = sizeof(*h);
    m->h = h;
    m->bodies = (const grbda_desc_body *)((const char *)blob + off);
    off += sizeof(grbda_desc_body) * (size_t)h->n_bodies;
    m->clusters = (const grbda_desc_cluster *)((const char *)blob + off);
    off += sizeof(grbda_desc_cluster) * (size_t)h->n_clusters;
    m->ints = (const int32_t *)((const char *)blob + off);
    off += sizeof(int32_t) * (size_t)((h->n_ints + 1) & ~1);
    m->dbls = (const double *)((const char *)blob + off);
    off += sizeof(double) * (size_t)h->n_doubles;
    if (off > bytes) return GRBDA_ORACLE_EBADBLOB;
    for (int c = 0; c < h->n_clusters; c++) {
        const grbda_desc_cluster *cl = &m->clusters[c];
        if (cl->n_bodies < 1 || cl->n_bodies > MAXK || cl->n_vel > MAXN || cl->n_vel < 0 ||
            cl->n_span_pos > MAXSP || cl->n_span_vel > MAXD || cl->n_constraint_rows > MAXROWS ||
            cl->parent_cluster >= c)
            return GRBDA_ORACLE_EUNSUPPORTED;
    }
    return 0;
}

/* index of the first spanning velocity of body i inside its cluster */
static int span_vel_offset(const model_t *m, const grbda_desc_cluster *cl, int i)
{
    int o = 0;
    for (int j = 0; j < i; j++) o += (m->bodies[cl->first_body + j].joint_type == GRBDA_JOINT_FREE) ? 6 : 1;
    return o;
}
static int span_pos_offset(const model_t *m, const grbda_desc_cluster *cl, int i)
{
    int o = 0;
    const int nori = m->h->ori_repr == GRBDA_ORI_QUATERNION ? 4 : 3;
    for (int j = 0; j < i; j++) o += (m->bodies[cl->first_body + j].joint_type == GRBDA_JOINT_FREE) ? 3 + nori : 1;
    return o;
}

/* ------------------------------------------------------------------------------------------ */
/* loop constraints                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* Position of a loop-constraint frame origin expressed in the NCA frame, with the data needed
 * for its first and second derivatives.  Follows implicitPositionConstraint
 * (src/Dynamics/ClusterTreeParsing.cpp:310-376): X = X_origin * prod(XJ_b * Xtree_b), the
 * constraint uses X.getTranslation().  For X = (E, r), r is the origin of the final frame in
 * NCA coordinates.  Joint b rotates about axis a_b (NCA coords) through point o_b. */
typedef struct {
    int n;
    int sub[MAXK];
    real a[MAXK][3], o[MAXK][3], p[3];
} chain_t;

static void chain_eval(const model_t *m, const grbda_desc_cluster *cl, const int32_t *subs, int n,
                       const double *origin /* E[9] r[3] */, const real *qs, chain_t *ch)
{
    xf_t X;
    xf_identity(&X);
    ch->n = n;
    for (int t = 0; t < n; t++) {
        const int sub = subs[t];
        const grbda_desc_body *b = &m->bodies[cl->first_body + sub];
        xf_t Xt, XJ, Xb;
        md_copy(Xt.E, b->Xtree_E, 9);
        md_copy(Xt.r, b->Xtree_r, 3);
        xf_identity(&XJ);
        coord_rotation(b->axis, qs[sub], XJ.E);
        xf_mul(&XJ, &Xt, &Xb);
        xf_mul(&Xb, &X, &X);
        ch->sub[t] = sub;
        /* joint origin in NCA coords = translation of X (XJ has r = 0);
         * joint axis in NCA coords = E^T e_axis (row `axis` of E) */
        for (int i = 0; i < 3; i++) { ch->o[t][i] = X.r[i]; ch->a[t][i] = X.E[b->axis * 3 + i]; }
    }
    xf_t Xo, Xf;
    md_copy(Xo.E, origin, 9);
    md_copy(Xo.r, origin + 9, 3);
    xf_mul(&Xo, &X, &Xf);
    for (int i = 0; i < 3; i++) ch->p[i] = Xf.r[i];
}

/* d p / d q_b = a_b x (p - o_b);  d2 p / dq_b dq_c (b before c) = a_b x (a_c x (p - o_c)) */
static void chain_accumulate(const chain_t *ch, real sign, const real *qds, real *Krow3 /* 3 x nsv */,
                             int nsv, real *Kdqd3 /* 3 */)
{
    for (int t = 0; t < ch->n; t++) {
        real d[3], J[3];
        for (int i = 0; i < 3; i++) d[i] = ch->p[i] - ch->o[t][i];
        cross3(ch->a[t], d, J);
        for (int i = 0; i < 3; i++) Krow3[i * nsv + ch->sub[t]] += sign * J[i];
    }
    if (!Kdqd3 || !qds) return;
    for (int b = 0; b < ch->n; b++)
        for (int c = 0; c < ch->n; c++) {
            const int lo = b < c ? b : c, hi = b < c ? c : b;
            real d[3], t1[3], t2[3];
            for (int i = 0; i < 3; i++) d[i] = ch->p[i] - ch->o[hi][i];
            cross3(ch->a[hi], d, t1);
            cross3(ch->a[lo], t1, t2);
            const real w = qds[ch->sub[b]] * qds[ch->sub[c]];
            for (int i = 0; i < 3; i++) Kdqd3[i] += sign * w * t2[i];
        }
}

/* Trig-polynomial implicit constraints: each row of phi is
 *     sum_t coef_t * prod_f  F_f( w_f . q + b_f ),   F in {identity, sin, cos}.
 * This is the data-driven form of the hand-written phi lambdas of the Tello hip and
 * knee-ankle differentials (src/Robots/Tello.cpp:139-163,237-261); the reference obtains
 * K = dphi/dq and Kdot via CasADi (GenericJoint.cpp:51-64), here they are differentiated
 * analytically.  K_dot*qd is the second directional derivative of phi along qd. */
static void trig_f(int type, real a, real *f0, real *f1, real *f2)
{
    if (type == 1) { *f0 = sin(a); *f1 = cos(a); *f2 = -sin(a); }
    else if (type == 2) { *f0 = cos(a); *f1 = -sin(a); *f2 = -cos(a); }
    else { *f0 = a; *f1 = 1.0; *f2 = 0.0; }
}
static int trigpoly_eval(const int32_t *ip, const double *dp, int nsp, int rows, const real *q,
                         const real *qd, real *K, real *Kdqd, real *phi)
{
    for (int r = 0; r < rows; r++) {
        const int nt = *ip++;
        real ph = 0, kd = 0;
        if (K) for (int j = 0; j < nsp; j++) K[r * nsp + j] = 0;
        for (int t = 0; t < nt; t++) {
            const int nf = *ip++;
            if (nf > 8) return GRBDA_ORACLE_EUNSUPPORTED;
            const real coef = (real)*dp++;
            real f0[8], f1[8], f2[8], ad[8];
            const double *wv[8];
            for (int f = 0; f < nf; f++) {
                const int type = *ip++;
                wv[f] = dp;
                real a = (real)dp[nsp], d = 0;
                for (int j = 0; j < nsp; j++) { a += (real)dp[j] * q[j]; if (qd) d += (real)dp[j] * qd[j]; }
                dp += nsp + 1;
                ad[f] = d;
                trig_f(type, a, &f0[f], &f1[f], &f2[f]);
            }
            real prod = coef;
            for (int f = 0; f < nf; f++) prod *= f0[f];
            ph += prod;
            for (int f = 0; f < nf; f++) {
                real others = coef;
                for (int g = 0; g < nf; g++) if (g != f) others *= f0[g];
                if (K) for (int j = 0; j < nsp; j++) K[r * nsp + j] += others * f1[f] * (real)wv[f][j];
                kd += others * f2[f] * ad[f] * ad[f];
                for (int g = 0; g < nf; g++) {
                    if (g == f) continue;
                    real rest = coef;
                    for (int h = 0; h < nf; h++) if (h != f && h != g) rest *= f0[h];
                    kd += rest * f1[f] * ad[f] * f1[g] * ad[g];
                }
            }
        }
        if (phi) phi[r] = ph;
        if (Kdqd) Kdqd[r] = kd;
    }
    return 0;
}

/* LoopConstraint evaluation: fills G, g (and K, k, phi) for cluster c given the cluster's
 * input coordinates.  y = positions as given to the model (independent for explicit kinds,
 * spanning for implicit kinds), yd = independent velocities. */
static int constraint_eval(const model_t *m, int c, const real *y, const real *yd, cws_t *w)
{
    const grbda_desc_cluster *cl = &m->clusters[c];
    const int n = cl->n_vel, nsv = cl->n_span_vel, nsp = cl->n_span_pos;
    w->rows = cl->n_constraint_rows;
    memset(w->g, 0, sizeof(real) * (size_t)nsv);
    memset(w->kb, 0, sizeof w->kb);
    memset(w->phi, 0, sizeof w->phi);
    memset(w->K, 0, sizeof(real) * (size_t)(MAXROWS * MAXD));

    if (cl->constraint_type == GRBDA_CONSTRAINT_FREE) {
        /* ClusterJoints::Free (FreeJoint.cpp:10-36): identity */
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < 6; j++) w->G[i * 6 + j] = (i == j);
        for (int i = 0; i < nsp; i++) w->qs[i] = y[i];
        for (int i = 0; i < 6; i++) w->qds[i] = yd[i];
        return 0;
    }
    if (cl->constraint_type == GRBDA_CONSTRAINT_STATIC) {
        /* Static::gamma = G y, qd_span = G yd (LoopConstraint.cpp:38-52, ClusterJoint.cpp:22-71) */
        const double *G = m->dbls + cl->dbl_offset;
        const double *K = G + nsv * n;
        md_copy(w->G, G, nsv * n);
        for (int r = 0; r < cl->n_constraint_rows; r++)
            for (int j = 0; j < nsv; j++) w->K[r * nsv + j] = K[r * nsv + j];
        for (int i = 0; i < nsv; i++) {
            real sq = 0, sv = 0;
            for (int j = 0; j < n; j++) { sq += G[i * n + j] * y[j]; sv += G[i * n + j] * yd[j]; }
            w->qs[i] = sq;
            w->qds[i] = sv;
        }
        return 0;
    }

    /* implicit kinds: positions are spanning (GenericJoint.cpp:246-249) */
    const int32_t *ip = m->ints + cl->int_offset;
    const double *dp = m->dbls + cl->dbl_offset;
    const int rows = cl->n_constraint_rows;
    const int32_t *is_ind;
    real Kdqd[MAXROWS];
    memset(Kdqd, 0, sizeof Kdqd);
    for (int i = 0; i < nsp; i++) w->qs[i] = y[i];

    /* K(q) first (does not need velocities) */
    if (cl->constraint_type == GRBDA_CONSTRAINT_LOOP_POSITION) {
        const int n_loops = ip[0];
        is_ind = ip + 1;
        const int32_t *lp = ip + 1 + cl->n_bodies;
        int row = 0;
        for (int l = 0; l < n_loops; l++) {
            const int np = lp[0];
            const int32_t *ps = lp + 1;
            const int ns = lp[1 + np];
            const int32_t *ss = lp + 2 + np;
            const int mask = lp[2 + np + ns];
            lp += 3 + np + ns;
            chain_t cp, cs;
            chain_eval(m, cl, ps, np, dp + 24 * l, w->qs, &cp);
            chain_eval(m, cl, ss, ns, dp + 24 * l + 12, w->qs, &cs);
            real K3[3 * MAXD];
            memset(K3, 0, sizeof K3);
            chain_accumulate(&cp, +1.0, NULL, K3, nsv, NULL);
            chain_accumulate(&cs, -1.0, NULL, K3, nsv, NULL);
            for (int ax = 0; ax < 3; ax++)
                if (mask & (1 << ax)) {
                    if (row >= rows) return GRBDA_ORACLE_EBADBLOB;
                    for (int j = 0; j < nsv; j++) w->K[row * nsv + j] = K3[ax * nsv + j];
                    w->phi[row] = cp.p[ax] - cs.p[ax];
                    row++;
                }
        }
        if (row != rows) return GRBDA_ORACLE_EBADBLOB;
    } else if (cl->constraint_type == GRBDA_CONSTRAINT_TRIG_POLY) {
        is_ind = ip;
        int rc = trigpoly_eval(ip + nsv, dp, nsp, rows, w->qs, NULL, w->K, NULL, w->phi);
        if (rc) return rc;
    } else
        return GRBDA_ORACLE_EUNSUPPORTED;

    /* G = P [I; -Kd^-1 Ki] (GenericJoint.cpp:71-83) */
    int ind[MAXD], dep[MAXD], ni = 0, nd = 0;
    for (int i = 0; i < nsv; i++) {
        if (is_ind[i]) ind[ni++] = i; else dep[nd++] = i;
    }
    if (ni != n || nd != rows) return GRBDA_ORACLE_EBADBLOB;
    real Kd[MAXROWS * MAXROWS], Ki[MAXROWS * MAXN];
    for (int r = 0; r < rows; r++) {
        for (int j = 0; j < nd; j++) Kd[r * nd + j] = w->K[r * nsv + dep[j]];
        for (int j = 0; j < ni; j++) Ki[r * ni + j] = w->K[r * nsv + ind[j]];
    }
    real Kd1[MAXROWS * MAXROWS];
    memcpy(Kd1, Kd, sizeof Kd);
    int rc = lu_solve(Kd1, Ki, nd, ni); /* Ki <- Kd^-1 Ki */
    if (rc) return rc;
    memset(w->G, 0, sizeof(real) * (size_t)(nsv * n));
    for (int j = 0; j < ni; j++) w->G[ind[j] * n + j] = 1.0;
    for (int r = 0; r < nd; r++)
        for (int j = 0; j < ni; j++) w->G[dep[r] * n + j] = -Ki[r * ni + j];
    /* qd_span = G yd */
    for (int i = 0; i < nsv; i++) {
        real s = 0;
        for (int j = 0; j < n; j++) s += w->G[i * n + j] * yd[j];
        w->qds[i] = s;
    }
    /* k = -Kdot qd (GenericJoint.cpp:57-64) */
    if (cl->constraint_type == GRBDA_CONSTRAINT_LOOP_POSITION) {
        const int n_loops = ip[0];
        const int32_t *lp = ip + 1 + cl->n_bodies;
        int row = 0;
        for (int l = 0; l < n_loops; l++) {
            const int np = lp[0];
            const int32_t *ps = lp + 1;
            const int ns = lp[1 + np];
            const int32_t *ss = lp + 2 + np;
            const int mask = lp[2 + np + ns];
            lp += 3 + np + ns;
            chain_t cp, cs;
            chain_eval(m, cl, ps, np, dp + 24 * l, w->qs, &cp);
            chain_eval(m, cl, ss, ns, dp + 24 * l + 12, w->qs, &cs);
            real K3[3 * MAXD], a3[3] = {0, 0, 0};
            memset(K3, 0, sizeof K3);
            chain_accumulate(&cp, +1.0, w->qds, K3, nsv, a3);
            chain_accumulate(&cs, -1.0, w->qds, K3, nsv, a3);
            for (int ax = 0; ax < 3; ax++)
                if (mask & (1 << ax)) Kdqd[row++] = a3[ax];
        }
    } else {
        rc = trigpoly_eval(ip + nsv, dp, nsp, rows, w->qs, w->qds, NULL, Kdqd, NULL);
        if (rc) return rc;
    }
    for (int r = 0; r < rows; r++) w->kb[r] = -Kdqd[r];
    /* g = P [0; Kd^-1 k] (GenericJoint.cpp:85-88) */
    real kk[MAXROWS];
    memcpy(kk, w->kb, sizeof kk);
    memcpy(Kd1, Kd, sizeof Kd);
    rc = lu_solve(Kd1, kk, nd, 1);
    if (rc) return rc;
    for (int r = 0; r < nd; r++) w->g[dep[r]] = kk[r];
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* ClusterTreeNode::updateKinematics -> ClusterJoints::*::updateKinematics +                   */
/* computeSpatialTransformFromParentToCurrentCluster.  Every explicit joint type of the        */
/* reference (Revolute, RevoluteWithRotor, RevolutePair(WithRotor), RevoluteTripleWithRotor)   */
/* is the Generic formula (src/Dynamics/ClusterJoints/GenericJoint.cpp:387-469) with a         */
/* Static constraint; Free is FreeJoint.cpp:28-46.                                             */
/* ------------------------------------------------------------------------------------------ */
static int cluster_kinematics(const model_t *m, int c, const real *q, const real *qd, cws_t *w)
{
    const grbda_desc_cluster *cl = &m->clusters[c];
    const int k = cl->n_bodies, n = cl->n_vel, nsv = cl->n_span_vel, dim = 6 * k;
    w->k = k; w->n = n; w->dim = dim; w->nsp = cl->n_span_pos; w->nsv = nsv;

    int rc = constraint_eval(m, c, q + cl->q_index, qd + cl->v_index, w);
    if (rc) return rc;

    /* single joints (include/grbda/Dynamics/Joints/Joint.h:61-68,95-98) */
    real S_span[MAXD * MAXD];
    memset(S_span, 0, sizeof(real) * (size_t)(dim * nsv));
    for (int i = 0; i < k; i++) {
        const grbda_desc_body *b = &m->bodies[cl->first_body + i];
        xf_t XJ, Xt;
        md_copy(Xt.E, b->Xtree_E, 9);
        md_copy(Xt.r, b->Xtree_r, 3);
        xf_identity(&XJ);
        const int so = span_vel_offset(m, cl, i), po = span_pos_offset(m, cl, i);
        if (b->joint_type == GRBDA_JOINT_FREE) {
            if (m->h->ori_repr == GRBDA_ORI_QUATERNION) quat_to_rot(w->qs + po + 3, XJ.E);
            else rpy_to_rot(w->qs + po + 3, XJ.E);
            for (int j = 0; j < 3; j++) XJ.r[j] = w->qs[po + j];
            w->XJ[i] = XJ; /* FreeJoint.cpp:45: Xup[0] = XJ, Xtree is not applied */
            for (int j = 0; j < 6; j++) S_span[(6 * i + j) * nsv + so + j] = 1.0;
        } else {
            coord_rotation(b->axis, w->qs[po], XJ.E);
            xf_mul(&XJ, &Xt, &w->XJ[i]);
            S_span[(6 * i + b->axis) * nsv + so] = 1.0; /* Spatial.h:296-316 */
        }
        /* connectivity (GenericJoint.cpp:471-485) */
        w->lam[i] = (b->parent >= cl->first_body && b->parent < cl->first_body + k)
                        ? b->parent - cl->first_body : -1;
        /* cluster ancestor sub-index (ClusterTreeModel.cpp:20-23) */
        int p = b->parent;
        while (p >= cl->first_body) p = m->bodies[p].parent;
        w->anc_sub[i] = p >= 0 ? m->bodies[p].sub_index : 0;
    }
    /* Xup (GenericJoint.cpp:453-469) */
    for (int i = 0; i < k; i++) {
        if (w->lam[i] >= 0) xf_mul(&w->XJ[i], &w->Xup[w->lam[i]], &w->Xup[i]);
        else w->Xup[i] = w->XJ[i];
    }
    /* X_intra (GenericJoint.cpp:406-424) */
    real X_intra[MAXD * MAXD], X_ring[MAXD * MAXD];
    memset(X_intra, 0, sizeof(real) * (size_t)(dim * dim));
    memset(X_ring, 0, sizeof(real) * (size_t)(dim * dim));
    for (int i = 0; i < dim; i++) X_intra[i * dim + i] = 1.0;
    for (int i = 0; i < k; i++) {
        int kk = i;
        int j = w->lam[i];
        while (j >= 0) {
            real prev[36], Xint[36], prod[36];
            for (int r = 0; r < 6; r++)
                for (int cc = 0; cc < 6; cc++) prev[r * 6 + cc] = X_intra[(6 * i + r) * dim + 6 * kk + cc];
            xf_matrix(&w->XJ[kk], Xint);
            mm(prev, Xint, prod, 6, 6, 6);
            for (int r = 0; r < 6; r++)
                for (int cc = 0; cc < 6; cc++) X_intra[(6 * i + r) * dim + 6 * j + cc] = prod[r * 6 + cc];
            kk = j;
            j = w->lam[j];
        }
    }
    /* S = X_intra S_span G ; vJ = X_intra S_span qd (GenericJoint.cpp:426-428) */
    real S_impl[MAXD * MAXD];
    mm(X_intra, S_span, S_impl, dim, dim, nsv);
    mm(S_impl, w->G, w->S, dim, nsv, n);
    mm(S_impl, w->qds, w->vJ, dim, nsv, 1);
    /* X_intra_ring (GenericJoint.cpp:430-447) */
    for (int i = 0; i < k; i++) {
        int j = w->lam[i];
        while (j >= 0) {
            real Xup[36], vp[6], vrel[6], cm[36], prod[36];
            for (int r = 0; r < 6; r++)
                for (int cc = 0; cc < 6; cc++) Xup[r * 6 + cc] = X_intra[(6 * i + r) * dim + 6 * j + cc];
            mm(Xup, w->vJ + 6 * j, vp, 6, 6, 1);
            for (int r = 0; r < 6; r++) vrel[r] = w->vJ[6 * i + r] - vp[r];
            crm_matrix(vrel, cm);
            mm(cm, Xup, prod, 6, 6, 6);
            for (int r = 0; r < 6; r++)
                for (int cc = 0; cc < 6; cc++) X_ring[(6 * i + r) * dim + 6 * j + cc] = -prod[r * 6 + cc];
            j = w->lam[j];
        }
    }
    /* cJ = X_ring S_span qd + S_impl g (GenericJoint.cpp:449-450) */
    real t1[MAXD], t2[MAXD], t3[MAXD];
    mm(S_span, w->qds, t1, dim, nsv, 1);
    mm(X_ring, t1, t2, dim, dim, 1);
    mm(S_impl, w->g, t3, dim, nsv, 1);
    for (int i = 0; i < dim; i++) w->cJ[i] = t2[i] + t3[i];
    /* block-diagonal I (ClusterTreeNode.cpp:17-23) */
    memset(w->I, 0, sizeof(real) * (size_t)(dim * dim));
    for (int i = 0; i < k; i++) {
        const grbda_desc_body *b = &m->bodies[cl->first_body + i];
        for (int r = 0; r < 6; r++)
            for (int cc = 0; cc < 6; cc++) w->I[(6 * i + r) * dim + 6 * i + cc] = b->inertia[r * 6 + cc];
    }
    return 0;
}

/* TreeModel::forwardKinematics (src/Dynamics/TreeModel.cpp:6-32) */
static int forward_kinematics(const model_t *m, const real *q, const real *qd, const real *fext,
                              cws_t *W)
{
    for (int c = 0; c < m->h->n_clusters; c++) {
        const grbda_desc_cluster *cl = &m->clusters[c];
        cws_t *w = &W[c];
        int rc = cluster_kinematics(m, c, q, qd, w);
        if (rc) return rc;
        for (int i = 0; i < w->k; i++) {
            if (cl->parent_cluster >= 0) {
                const cws_t *p = &W[cl->parent_cluster];
                real t[6];
                xf_motion(&w->Xup[i], p->v + 6 * w->anc_sub[i], t);
                for (int r = 0; r < 6; r++) w->v[6 * i + r] = t[r] + w->vJ[6 * i + r];
                xf_mul(&w->Xup[i], &p->Xa[w->anc_sub[i]], &w->Xa[i]);
            } else {
                for (int r = 0; r < 6; r++) w->v[6 * i + r] = w->vJ[6 * i + r];
                w->Xa[i] = w->Xup[i];
            }
            crm(w->v + 6 * i, w->vJ + 6 * i, w->avp + 6 * i);
        }
        /* TreeModel::setExternalForces (TreeModel.cpp:214-239) */
        w->has_ext = 0;
        if (fext) {
            for (int i = 0; i < w->k; i++)
                for (int r = 0; r < 6; r++) {
                    w->fext[6 * i + r] = fext[(size_t)(cl->first_body + i) * 6 + r];
                    if (w->fext[6 * i + r] != 0.0) w->has_ext = 1;
                }
        }
    }
    return 0;
}

/* GeneralizedTransform::inverseTransformSpatialInertia (SpatialTransforms.cpp:364-369,415-477):
 * M_out[p(a), p(b)] += X_a^T M[a,b] X_b */
static void add_inverse_transformed_inertia(const cws_t *w, const real *Ia, cws_t *p)
{
    real Xm[MAXK][36];
    for (int i = 0; i < w->k; i++) xf_matrix(&w->Xup[i], Xm[i]);
    for (int a = 0; a < w->k; a++)
        for (int b = 0; b < w->k; b++) {
            real blk[36], t[36], o[36];
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 6; c++) blk[r * 6 + c] = Ia[(6 * a + r) * w->dim + 6 * b + c];
            mm(blk, Xm[b], t, 6, 6, 6);
            mtm(Xm[a], t, o, 6, 6, 6);
            const int pa = w->anc_sub[a], pb = w->anc_sub[b];
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 6; c++) p->IA[(6 * pa + r) * p->dim + 6 * pb + c] += o[r * 6 + c];
        }
}

static void bias_force(const model_t *m, cws_t *w, real *out /* dim */)
{
    /* generalForceCrossProduct(v, I v) (Spatial.h:196-215) minus external forces
     * (ClusterTreeDynamics.cpp:95-105, SpatialTransforms.cpp:234-250) */
    (void)m;
    real Iv[MAXD];
    mm(w->I, w->v, Iv, w->dim, w->dim, 1);
    for (int i = 0; i < w->k; i++) crf(w->v + 6 * i, Iv + 6 * i, out + 6 * i);
    if (w->has_ext)
        for (int i = 0; i < w->k; i++) {
            real t[6];
            xf_force(&w->Xa[i], w->fext + 6 * i, t);
            for (int r = 0; r < 6; r++) out[6 * i + r] -= t[r];
        }
}

/* ClusterTreeModel::forwardDynamics + updateArticulatedBodies
 * (src/Dynamics/ClusterTreeDynamics.cpp:85-191) */
static int aba_one(const model_t *m, const real *q, const real *qd, const real *tau,
                   const real *fext, real *ydd, cws_t *W)
{
    const int nc = m->h->n_clusters;
    int rc = forward_kinematics(m, q, qd, fext, W);
    if (rc) return rc;
    for (int c = 0; c < nc; c++) memcpy(W[c].IA, W[c].I, sizeof(real) * (size_t)(W[c].dim * W[c].dim));
    /* updateArticulatedBodies backward pass (:171-188) */
    for (int c = nc - 1; c >= 0; c--) {
        cws_t *w = &W[c];
        const int dim = w->dim, n = w->n;
        real D[MAXN * MAXN], UT[MAXN * MAXD];
        mm(w->IA, w->S, w->U, dim, dim, n);
        mtm(w->S, w->U, D, dim, n, n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < dim; j++) UT[i * dim + j] = w->U[j * n + i];
        rc = lu_solve(D, UT, n, dim);
        if (rc) return rc;
        memcpy(w->DinvUT, UT, sizeof(real) * (size_t)(n * dim));
        if (m->clusters[c].parent_cluster >= 0) {
            real UD[MAXD * MAXD];
            mm(w->U, w->DinvUT, UD, dim, n, dim);
            for (int i = 0; i < dim * dim; i++) w->Ia[i] = w->IA[i] - UD[i];
            add_inverse_transformed_inertia(w, w->Ia, &W[m->clusters[c].parent_cluster]);
        }
    }
    /* bias forces (:94-105) */
    for (int c = 0; c < nc; c++) bias_force(m, &W[c], W[c].pA);
    /* backward pass (:108-129) */
    for (int c = nc - 1; c >= 0; c--) {
        cws_t *w = &W[c];
        const grbda_desc_cluster *cl = &m->clusters[c];
        const int dim = w->dim, n = w->n;
        real STp[MAXN];
        mtm(w->S, w->pA, STp, dim, n, 1);
        for (int i = 0; i < n; i++) w->u[i] = tau[cl->v_index + i] - STp[i];
        /* D_inv_u = D^-1 u: recompute D (the reference keeps the factorisation) */
        real D[MAXN * MAXN];
        mtm(w->S, w->U, D, dim, n, n);
        memcpy(w->Dinvu, w->u, sizeof(real) * (size_t)n);
        rc = lu_solve(D, w->Dinvu, n, 1);
        if (rc) return rc;
        if (cl->parent_cluster >= 0) {
            cws_t *p = &W[cl->parent_cluster];
            real cv[MAXD], t1[MAXD], t2[MAXD], pa[MAXD];
            for (int i = 0; i < dim; i++) cv[i] = w->cJ[i] + w->avp[i];
            mm(w->Ia, cv, t1, dim, dim, 1);
            mm(w->U, w->Dinvu, t2, dim, n, 1);
            for (int i = 0; i < dim; i++) pa[i] = w->pA[i] + t1[i] + t2[i];
            for (int i = 0; i < w->k; i++) {
                real t[6];
                xf_inv_force(&w->Xup[i], pa + 6 * i, t);
                for (int r = 0; r < 6; r++) p->pA[6 * w->anc_sub[i] + r] += t[r];
            }
        }
    }
    /* forward pass (:131-152) */
    for (int c = 0; c < nc; c++) {
        cws_t *w = &W[c];
        const grbda_desc_cluster *cl = &m->clusters[c];
        const int dim = w->dim, n = w->n;
        real at[MAXD], t[MAXN], Sq[MAXD];
        for (int i = 0; i < w->k; i++) {
            real ap[6], x[6];
            if (cl->parent_cluster >= 0)
                memcpy(ap, W[cl->parent_cluster].a + 6 * w->anc_sub[i], sizeof ap);
            else
                for (int r = 0; r < 6; r++) ap[r] = -m->h->gravity[r];
            xf_motion(&w->Xup[i], ap, x);
            for (int r = 0; r < 6; r++) at[6 * i + r] = x[r] + w->cJ[6 * i + r] + w->avp[6 * i + r];
        }
        mm(w->DinvUT, at, t, n, dim, 1);
        for (int i = 0; i < n; i++) ydd[cl->v_index + i] = w->Dinvu[i] - t[i];
        mm(w->S, ydd + cl->v_index, Sq, dim, n, 1);
        for (int i = 0; i < dim; i++) w->a[i] = at[i] + Sq[i];
    }
    return 0;
}

/* TreeModel::forwardAccelerationKinematics + recursiveNewtonEulerAlgorithm
 * (src/Dynamics/TreeModel.cpp:34-57,173-212) */
static int rnea_one(const model_t *m, const real *q, const real *qd, const real *ydd,
                    const real *fext, real *tau, cws_t *W)
{
    const int nc = m->h->n_clusters;
    int rc = forward_kinematics(m, q, qd, fext, W);
    if (rc) return rc;
    for (int c = 0; c < nc; c++) {
        cws_t *w = &W[c];
        const grbda_desc_cluster *cl = &m->clusters[c];
        real Sq[MAXD], Ia[MAXD], b[MAXD];
        mm(w->S, ydd + cl->v_index, Sq, w->dim, w->n, 1);
        for (int i = 0; i < w->k; i++) {
            real ap[6], x[6];
            if (cl->parent_cluster >= 0)
                memcpy(ap, W[cl->parent_cluster].a + 6 * w->anc_sub[i], sizeof ap);
            else
                for (int r = 0; r < 6; r++) ap[r] = -m->h->gravity[r];
            xf_motion(&w->Xup[i], ap, x);
            for (int r = 0; r < 6; r++)
                w->a[6 * i + r] = x[r] + Sq[6 * i + r] + w->cJ[6 * i + r] + w->avp[6 * i + r];
        }
        mm(w->I, w->a, Ia, w->dim, w->dim, 1);
        bias_force(m, w, b);
        for (int i = 0; i < w->dim; i++) w->f[i] = Ia[i] + b[i];
    }
    for (int c = nc - 1; c >= 0; c--) {
        cws_t *w = &W[c];
        const grbda_desc_cluster *cl = &m->clusters[c];
        mtm(w->S, w->f, tau + cl->v_index, w->dim, w->n, 1);
        if (cl->parent_cluster >= 0) {
            cws_t *p = &W[cl->parent_cluster];
            for (int i = 0; i < w->k; i++) {
                real t[6];
                xf_inv_force(&w->Xup[i], w->f + 6 * i, t);
                for (int r = 0; r < 6; r++) p->f[6 * w->anc_sub[i] + r] += t[r];
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Independent check: spanning tree CRBA + RNEA + Projection                                  */
/* (RigidBodyTreeDynamics.cpp:86-97; TreeModel.cpp:115-171 restated per body)                 */
/* ------------------------------------------------------------------------------------------ */
static int projection_one(const model_t *m, const real *q, const real *qd, const real *tau,
                          const real *fext, real *ydd, cws_t *W)
{
    const int nb = m->h->n_bodies, nc = m->h->n_clusters, nv = m->h->nv;
    /* constraints + single-joint transforms come from the same kinematics call; everything
     * below uses only XJ (single joint * Xtree), qds, G, g */
    int rc = forward_kinematics(m, q, qd, fext, W);
    if (rc) return rc;
    int *voff = (int *)malloc(sizeof(int) * (size_t)(nb + 1));
    int *ndof = (int *)malloc(sizeof(int) * (size_t)nb);
    if (!voff || !ndof) return GRBDA_ORACLE_ENOMEM;
    int ns = 0;
    for (int b = 0; b < nb; b++) {
        ndof[b] = m->bodies[b].joint_type == GRBDA_JOINT_FREE ? 6 : 1;
        voff[b] = ns;
        ns += ndof[b];
    }
    voff[nb] = ns;
    real *H = (real *)calloc((size_t)ns * ns, sizeof(real));
    real *C = (real *)calloc((size_t)ns, sizeof(real));
    real *Gf = (real *)calloc((size_t)ns * nv, sizeof(real));
    real *gf = (real *)calloc((size_t)ns, sizeof(real));
    real *Ic = (real *)malloc(sizeof(real) * 36 * (size_t)nb);
    real *vb = (real *)malloc(sizeof(real) * 6 * (size_t)nb);
    real *ab = (real *)malloc(sizeof(real) * 6 * (size_t)nb);
    real *fb = (real *)malloc(sizeof(real) * 6 * (size_t)nb);
    xf_t *Xb = (xf_t *)malloc(sizeof(xf_t) * (size_t)nb);
    xf_t *Xab = (xf_t *)malloc(sizeof(xf_t) * (size_t)nb);
    real *A = (real *)malloc(sizeof(real) * (size_t)nv * nv);
    real *rhs = (real *)malloc(sizeof(real) * (size_t)nv);
    real *HG = (real *)malloc(sizeof(real) * (size_t)ns * nv);
    real *tmp = (real *)malloc(sizeof(real) * (size_t)ns);
    if (!H || !C || !Gf || !gf || !Ic || !vb || !ab || !fb || !Xb || !Xab || !A || !rhs || !HG || !tmp)
        return GRBDA_ORACLE_ENOMEM;

    /* assemble block-diagonal G, stacked g, per-body joint velocity */
    real *qds_b = (real *)calloc((size_t)ns, sizeof(real));
    for (int c = 0; c < nc; c++) {
        const grbda_desc_cluster *cl = &m->clusters[c];
        const cws_t *w = &W[c];
        const int s0 = voff[cl->first_body];
        for (int i = 0; i < w->nsv; i++) {
            for (int j = 0; j < w->n; j++) Gf[(size_t)(s0 + i) * nv + cl->v_index + j] = w->G[i * w->n + j];
            gf[s0 + i] = w->g[i];
            qds_b[s0 + i] = w->qds[i];
        }
        for (int i = 0; i < w->k; i++) Xb[cl->first_body + i] = w->XJ[i];
    }
    /* RNEA on the spanning tree with qdd_span = 0 -> C; body velocities */
    for (int b = 0; b < nb; b++) {
        const grbda_desc_body *bd = &m->bodies[b];
        real vj[6] = {0, 0, 0, 0, 0, 0};
        if (bd->joint_type == GRBDA_JOINT_FREE) for (int r = 0; r < 6; r++) vj[r] = qds_b[voff[b] + r];
        else vj[bd->axis] = qds_b[voff[b]];
        real ap[6], vp[6] = {0, 0, 0, 0, 0, 0}, x[6], cr[6];
        if (bd->parent >= 0) {
            memcpy(ap, ab + 6 * bd->parent, sizeof ap);
            memcpy(vp, vb + 6 * bd->parent, sizeof vp);
            xf_mul(&Xb[b], &Xab[bd->parent], &Xab[b]);
        } else {
            for (int r = 0; r < 6; r++) ap[r] = -m->h->gravity[r];
            Xab[b] = Xb[b];
        }
        xf_motion(&Xb[b], vp, x);
        for (int r = 0; r < 6; r++) vb[6 * b + r] = x[r] + vj[r];
        crm(vb + 6 * b, vj, cr);
        xf_motion(&Xb[b], ap, x);
        for (int r = 0; r < 6; r++) ab[6 * b + r] = x[r] + cr[r];
        real Ia[6], Iv[6], cf[6];
        real Ib[36];
        md_copy(Ib, bd->inertia, 36);
        mm(Ib, ab + 6 * b, Ia, 6, 6, 1);
        mm(Ib, vb + 6 * b, Iv, 6, 6, 1);
        crf(vb + 6 * b, Iv, cf);
        for (int r = 0; r < 6; r++) fb[6 * b + r] = Ia[r] + cf[r];
        if (fext) {
            real t[6];
            xf_force(&Xab[b], fext + 6 * (size_t)b, t);
            for (int r = 0; r < 6; r++) fb[6 * b + r] -= t[r];
        }
        md_copy(Ic + 36 * b, bd->inertia, 36);
    }
    for (int b = nb - 1; b >= 0; b--) {
        const grbda_desc_body *bd = &m->bodies[b];
        if (bd->joint_type == GRBDA_JOINT_FREE) for (int r = 0; r < 6; r++) C[voff[b] + r] = fb[6 * b + r];
        else C[voff[b]] = fb[6 * b + bd->axis];
        if (bd->parent >= 0) {
            real t[6];
            xf_inv_force(&Xb[b], fb + 6 * b, t);
            for (int r = 0; r < 6; r++) fb[6 * bd->parent + r] += t[r];
        }
    }
    /* CRBA on the spanning tree */
    for (int b = nb - 1; b >= 0; b--) {
        const grbda_desc_body *bd = &m->bodies[b];
        if (bd->parent >= 0) {
            real Xm[36], t[36], o[36];
            xf_matrix(&Xb[b], Xm);
            mm(Ic + 36 * b, Xm, t, 6, 6, 6);
            mtm(Xm, t, o, 6, 6, 6);
            for (int i = 0; i < 36; i++) Ic[36 * bd->parent + i] += o[i];
        }
    }
    for (int b = 0; b < nb; b++) {
        const grbda_desc_body *bd = &m->bodies[b];
        for (int d = 0; d < ndof[b]; d++) {
            real s[6] = {0, 0, 0, 0, 0, 0}, F[6];
            s[bd->joint_type == GRBDA_JOINT_FREE ? d : bd->axis] = 1.0;
            mm(Ic + 36 * b, s, F, 6, 6, 1);
            /* own block */
            for (int e = 0; e < ndof[b]; e++) {
                const int ax = bd->joint_type == GRBDA_JOINT_FREE ? e : bd->axis;
                H[(size_t)(voff[b] + e) * ns + voff[b] + d] = F[ax];
            }
            int j = b;
            while (m->bodies[j].parent >= 0) {
                real t[6];
                xf_inv_force(&Xb[j], F, t);
                memcpy(F, t, sizeof F);
                j = m->bodies[j].parent;
                const grbda_desc_body *bj = &m->bodies[j];
                for (int e = 0; e < ndof[j]; e++) {
                    const int ax = bj->joint_type == GRBDA_JOINT_FREE ? e : bj->axis;
                    H[(size_t)(voff[j] + e) * ns + voff[b] + d] = F[ax];
                    H[(size_t)(voff[b] + d) * ns + voff[j] + e] = F[ax];
                }
            }
        }
    }
    /* A = G^T H G ; b = tau - G^T (C + H g) */
    mm(H, Gf, HG, ns, ns, nv);
    mtm(Gf, HG, A, ns, nv, nv);
    mm(H, gf, tmp, ns, ns, 1);
    for (int i = 0; i < ns; i++) tmp[i] += C[i];
    mtm(Gf, tmp, rhs, ns, nv, 1);
    for (int i = 0; i < nv; i++) rhs[i] = tau[i] - rhs[i];
    rc = lu_solve(A, rhs, nv, 1);
    if (!rc) memcpy(ydd, rhs, sizeof(real) * (size_t)nv);
    free(voff); free(ndof); free(H); free(C); free(Gf); free(gf); free(Ic); free(vb); free(ab);
    free(fb); free(Xb); free(Xab); free(A); free(rhs); free(HG); free(tmp); free(qds_b);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* batch drivers                                                                              */
/* ------------------------------------------------------------------------------------------ */
typedef int (*one_fn)(const model_t *, const real *, const real *, const real *, const real *,
                      real *, cws_t *);

static int run_batch(const void *blob, size_t bytes, const real *q, const real *qd, const real *x,
                     const real *fext, real *out, size_t B, one_fn fn)
{
    model_t m;
    int rc = parse_blob(blob, bytes, &m);
    if (rc) return rc;
    cws_t *W = (cws_t *)malloc(sizeof(cws_t) * (size_t)m.h->n_clusters);
    if (!W) return GRBDA_ORACLE_ENOMEM;
    const int nq = m.h->nq, nv = m.h->nv, nb = m.h->n_bodies;
    for (size_t s = 0; s < B && !rc; s++)
        rc = fn(&m, q + s * nq, qd + s * nv, x + s * nv, fext ? fext + s * nb * 6 : NULL, out + s * nv, W);
    free(W);
    return rc;
}

/* TreeNode::Xa_ of every body after TreeModel::forwardKinematics (TreeModel.cpp:6-32): out[B][n_bodies][12] */
int grbda_oracle_body_poses(const void *blob, size_t bytes, const real *q, real *out, size_t B)
{
    model_t m;
    int rc = parse_blob(blob, bytes, &m);
    if (rc) return rc;
    cws_t *W = (cws_t *)malloc(sizeof(cws_t) * (size_t)m.h->n_clusters);
    real *zero = (real *)calloc((size_t)m.h->nv, sizeof(real));
    if (!W || !zero) { free(W); free(zero); return GRBDA_ORACLE_ENOMEM; }
    const int nq = m.h->nq, nb = m.h->n_bodies;
    for (size_t s = 0; s < B && !rc; s++) {
        rc = forward_kinematics(&m, q + s * nq, zero, NULL, W);
        for (int c = 0; c < m.h->n_clusters && !rc; c++)
            for (int i = 0; i < W[c].k; i++) {
                real *o = out + (s * (size_t)nb + (size_t)(m.clusters[c].first_body + i)) * 12;
                memcpy(o, W[c].Xa[i].E, sizeof(real) * 9);
                memcpy(o + 9, W[c].Xa[i].r, sizeof(real) * 3);
            }
    }
    free(W);
    free(zero);
    return rc;
}

int grbda_oracle_forward_dynamics(const void *blob, size_t bytes, const real *q, const real *qd,
                                  const real *tau, const real *f_ext, real *ydd, size_t B)
{
    return run_batch(blob, bytes, q, qd, tau, f_ext, ydd, B, aba_one);
}
int grbda_oracle_inverse_dynamics(const void *blob, size_t bytes, const real *q, const real *qd,
                                  const real *ydd, const real *f_ext, real *tau, size_t B)
{
    return run_batch(blob, bytes, q, qd, ydd, f_ext, tau, B, rnea_one);
}
int grbda_oracle_forward_dynamics_projection(const void *blob, size_t bytes, const real *q,
                                             const real *qd, const real *tau,
                                             const real *f_ext, real *ydd, size_t B)
{
    return run_batch(blob, bytes, q, qd, tau, f_ext, ydd, B, projection_one);
}

typedef struct {
    const void *blob; size_t bytes; const real *q, *qd, *tau; real *ydd; size_t lo, hi; int nq, nv, rc;
} mt_arg_t;
static void *mt_worker(void *p)
{
    mt_arg_t *a = (mt_arg_t *)p;
    a->rc = grbda_oracle_forward_dynamics(a->blob, a->bytes, a->q + a->lo * a->nq, a->qd + a->lo * a->nv,
                                          a->tau + a->lo * a->nv, NULL, a->ydd + a->lo * a->nv,
                                          a->hi - a->lo);
    return NULL;
}
int grbda_oracle_forward_dynamics_mt(const void *blob, size_t bytes, const real *q,
                                     const real *qd, const real *tau, real *ydd, size_t B,
                                     int n_threads)
{
    model_t m;
    int rc = parse_blob(blob, bytes, &m);
    if (rc) return rc;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    mt_arg_t args[256];
    for (int t = 0; t < n_threads; t++) {
        mt_arg_t a = {blob, bytes, q, qd, tau, ydd, B * (size_t)t / n_threads,
                      B * (size_t)(t + 1) / n_threads, m.h->nq, m.h->nv, 0};
        args[t] = a;
        if (pthread_create(&th[t], NULL, mt_worker, &args[t])) return GRBDA_ORACLE_ENOMEM;
    }
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        if (args[t].rc) rc = args[t].rc;
    }
    return rc;
}

int grbda_oracle_cluster_constraint(const void *blob, size_t bytes, int cluster, const real *q,
                                    const real *qd, real *G, real *g, real *K, real *k,
                                    real *phi)
{
    model_t m;
    int rc = parse_blob(blob, bytes, &m);
    if (rc) return rc;
    if (cluster < 0 || cluster >= m.h->n_clusters) return GRBDA_ORACLE_EBADBLOB;
    cws_t *w = (cws_t *)malloc(sizeof(cws_t));
    if (!w) return GRBDA_ORACLE_ENOMEM;
    const grbda_desc_cluster *cl = &m.clusters[cluster];
    rc = constraint_eval(&m, cluster, q + cl->q_index, qd + cl->v_index, w);
    if (!rc) {
        const int nsv = cl->n_span_vel, n = cl->n_vel, rows = cl->n_constraint_rows;
        if (G) memcpy(G, w->G, sizeof(real) * (size_t)(nsv * n));
        if (g) memcpy(g, w->g, sizeof(real) * (size_t)nsv);
        if (K) for (int r = 0; r < rows; r++) memcpy(K + r * nsv, w->K + r * nsv, sizeof(real) * (size_t)nsv);
        if (k) memcpy(k, w->kb, sizeof(real) * (size_t)rows);
        if (phi) memcpy(phi, w->phi, sizeof(real) * (size_t)rows);
    }
    free(w);
    return rc;
}

/* Spanning state of every cluster, cluster after cluster: ClusterJoints::Base::toSpanningTreeState
 * (ClusterJoint.cpp:22-71) -- q_span = gamma(y) = G y for explicit constraints (LoopConstraint.cpp:38-52), the given
 * spanning positions for implicit ones, qd_span = G yd -- i.e. what the reference's tests hand to setState when
 * use_spanning_state is set (testRigidBodyDynamicsAlgos.cpp:45-72).  gmax[b] (may be NULL): largest |G| entry in the
 * dependent rows of the implicit clusters (G = P [1; -Kd^-1 Ki], GenericJoint.cpp:75-83); kcond[b] (may be NULL): largest
 * Frobenius condition number |Kd|_F |Kd^-1|_F of their dependent blocks. */
int grbda_oracle_spanning_state(const void *blob, size_t bytes, const real *q, const real *qd, real *q_span,
                                real *qd_span, real *gmax, real *kcond, size_t B)
{
    model_t m;
    int rc = parse_blob(blob, bytes, &m);
    if (rc) return rc;
    cws_t *w = (cws_t *)malloc(sizeof(cws_t));
    if (!w) return GRBDA_ORACLE_ENOMEM;
    int nsp = 0, nsv = 0;
    for (int c = 0; c < m.h->n_clusters; c++) { nsp += m.clusters[c].n_span_pos; nsv += m.clusters[c].n_span_vel; }
    for (size_t b = 0; b < B && !rc; b++) {
        const real *qb = q + b * (size_t)m.h->nq, *vb = qd + b * (size_t)m.h->nv;
        real *qo = q_span ? q_span + b * (size_t)nsp : NULL, *vo = qd_span ? qd_span + b * (size_t)nsv : NULL;
        real gm = 0, kc = 0;
        for (int c = 0; c < m.h->n_clusters && !rc; c++) {
            const grbda_desc_cluster *cl = &m.clusters[c];
            rc = constraint_eval(&m, c, qb + cl->q_index, vb + cl->v_index, w);
            if (rc) break;
            if (qo) { memcpy(qo, w->qs, sizeof(real) * (size_t)cl->n_span_pos); qo += cl->n_span_pos; }
            if (vo) { memcpy(vo, w->qds, sizeof(real) * (size_t)cl->n_span_vel); vo += cl->n_span_vel; }
            if (cl->constraint_type == GRBDA_CONSTRAINT_LOOP_POSITION || cl->constraint_type == GRBDA_CONSTRAINT_TRIG_POLY) {
                const int32_t *is_ind = m.ints + cl->int_offset + (cl->constraint_type == GRBDA_CONSTRAINT_LOOP_POSITION ? 1 : 0);
                for (int i = 0; i < cl->n_span_vel; i++)
                    if (!is_ind[i])
                        for (int a = 0; a < cl->n_vel; a++) {
                            const real x = fabs(w->G[i * cl->n_vel + a]);
                            if (x > gm || x != x) gm = x;
                        }
                /* Kd = K[:, dependent]; its inverse by LU on the identity */
                const int rows = cl->n_constraint_rows, nsv = cl->n_span_vel;
                real Kd[MAXROWS * MAXROWS], Ki[MAXROWS * MAXROWS], f1 = 0, f2 = 0;
                int dcol = 0;
                for (int i = 0; i < nsv; i++)
                    if (!is_ind[i]) {
                        for (int r = 0; r < rows; r++) Kd[r * rows + dcol] = w->K[r * nsv + i];
                        dcol++;
                    }
                for (int i = 0; i < rows * rows; i++) { f1 += Kd[i] * Kd[i]; Ki[i] = (i / rows == i % rows); }
                if (dcol != rows || lu_solve(Kd, Ki, rows, rows)) { kc = NAN; }
                else {
                    for (int i = 0; i < rows * rows; i++) f2 += Ki[i] * Ki[i];
                    const real cn = sqrt(f1 * f2);
                    if (cn > kc || cn != cn) kc = cn;
                }
            }
        }
        if (gmax) gmax[b] = gm;
        if (kcond) kcond[b] = kc;
    }
    free(w);
    return rc;
}

int grbda_oracle_project_positions(const void *blob, size_t bytes, real *q, size_t B,
                                   int max_iter, int *ok)
{
    model_t m;
    int rc = parse_blob(blob, bytes, &m);
    if (rc) return rc;
    cws_t *w = (cws_t *)malloc(sizeof(cws_t));
    if (!w) return GRBDA_ORACLE_ENOMEM;
    real zeros[MAXN] = {0};
    for (size_t s = 0; s < B; s++) {
        int good = 1;
        for (int c = 0; c < m.h->n_clusters; c++) {
            const grbda_desc_cluster *cl = &m.clusters[c];
            if (cl->constraint_type == GRBDA_CONSTRAINT_STATIC || cl->constraint_type == GRBDA_CONSTRAINT_FREE)
                continue;
            const int32_t *ip = m.ints + cl->int_offset;
            const int32_t *is_ind = cl->constraint_type == GRBDA_CONSTRAINT_LOOP_POSITION ? ip + 1 : ip;
            real *y = q + s * m.h->nq + cl->q_index;
            const int nsv = cl->n_span_vel, rows = cl->n_constraint_rows;
            int dep[MAXD], nd = 0;
            for (int i = 0; i < nsv; i++) if (!is_ind[i]) dep[nd++] = i;
            real nrm = 1e30;
            for (int it = 0; it <= max_iter; it++) {
                rc = constraint_eval(&m, c, y, zeros, w);
                if (rc == GRBDA_ORACLE_ESINGULAR) { rc = 0; break; }
                if (rc) { free(w); return rc; }
                nrm = 0;
                for (int r = 0; r < rows; r++) nrm += w->phi[r] * w->phi[r];
                nrm = sqrt(nrm);
                if (nrm < (sizeof(real) > 8 ? (real)1e-17 : (real)1e-12) || it == max_iter) break;
                real Kd[MAXROWS * MAXROWS], dq[MAXROWS];
                for (int r = 0; r < rows; r++) {
                    for (int j = 0; j < nd; j++) Kd[r * nd + j] = w->K[r * nsv + dep[j]];
                    dq[r] = -w->phi[r];
                }
                if (lu_solve(Kd, dq, nd, 1)) break;
                for (int j = 0; j < nd; j++) y[dep[j]] += dq[j];
            }
            if (!(nrm < 1e-8)) good = 0;
        }
        if (ok) ok[s] = good;
    }
    free(w);
    return 0;
}
