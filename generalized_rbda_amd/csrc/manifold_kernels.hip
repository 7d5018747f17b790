// manifold_kernels.hip -- analytic first-order derivatives of the cluster dynamics for models with IMPLICIT clusters
// (BASELINE config 5 names "four_bar / six_bar loop clusters"; Tello's differentials are the other user).
//
// The reference differentiates CasADi graphs through G(q), g(q, qd) of its GenericImplicit clusters
// (src/Dynamics/ClusterJoints/GenericJoint.cpp:57-90; yardstick UnitTests/testRigidBodyDynamicsAlgosDerivatives.cpp:271-383).
// Here the inverse dynamics in the independent coordinates y is written through the SPANNING tree,
//     tau(y, yd, ydd) = G^T tau_s(q_s, qd_s, qdd_s),   qd_s = G yd,   qdd_s = G ydd + g,   d q_s / d y = G,
// so, with A_q = d tau_s / d q_s, A_v = d tau_s / d qd_s, H_s = d tau_s / d qdd_s of the spanning tree (an explicit model:
// rnea_deriv_kernel on the spanning plan, plan.cpp make_spanning_blob) and G_a' = d G / d y_a along the manifold,
//     d tau / d y_a  = G^T [A_q G e_a + A_v (G_a' yd) + H_s (G_a' ydd + d g / d y_a)] + G_a'^T tau_s
//     d tau / d yd_a = G^T [A_v G e_a + H_s (d g / d yd_a)]
//     H              = G^T H_s G
// and d ydd / d (y, yd, tau) = -H^-1 (...), H^-1 by the same batched SPD solve as the explicit models.
//
// Derivatives of the constraint quantities (manifold_constraint_kernel, one state per lane).  With K = d phi / d q split into
// dependent / independent columns, K G = 0 and Kd g_dep = -kappa, kappa = D^2 phi [qd_s, qd_s] (= Kdot qd_s); along the
// direction d = G e_a of the manifold, K' = D K [d]:
//     G_a'(dep rows) = -Kd^-1 K' G,        d g_dep / d yd_a = -2 Kd^-1 K' qd_s,
//     d g_dep / d y_a = -Kd^-1 (kappa' + K'_d g_dep),   kappa' = D^3 phi [d, qd_s, qd_s] + 2 D^2 phi [qd_s, G_a' yd].
// K' and kappa' come from evaluating K and kappa in DUAL numbers (value + first-order part along d): the same templated code
// that computes K and kappa -- the point Jacobians / velocity-product accelerations of the two sub-chains of a URDF+ <loop>
// (ClusterTreeParsing.cpp:310-376), term-by-term differentiation of a trig-polynomial phi -- runs once with plain numbers and once per
// independent coordinate with duals.  No finite differences, no re-projection.
#include <hip/hip_runtime.h>

#include "devplan.h"

namespace grbda_hip {

#include "devmath.h"

// ---- dual numbers ------------------------------------------------------------------------------------------------
template <class T>
struct Du {
    T v, d;
    __device__ __forceinline__ Du() : v(0), d(0) {}
    __device__ __forceinline__ Du(T x) : v(x), d(0) {}
    __device__ __forceinline__ Du(T x, T y) : v(x), d(y) {}
};
template <class T> __device__ __forceinline__ Du<T> operator+(Du<T> a, Du<T> b) { return Du<T>(a.v + b.v, a.d + b.d); }
template <class T> __device__ __forceinline__ Du<T> operator-(Du<T> a, Du<T> b) { return Du<T>(a.v - b.v, a.d - b.d); }
template <class T> __device__ __forceinline__ Du<T> operator-(Du<T> a) { return Du<T>(-a.v, -a.d); }
template <class T> __device__ __forceinline__ Du<T> operator*(Du<T> a, Du<T> b) { return Du<T>(a.v * b.v, a.v * b.d + a.d * b.v); }
template <class T> __device__ __forceinline__ Du<T> operator*(Du<T> a, T b) { return Du<T>(a.v * b, a.d * b); }
template <class T> __device__ __forceinline__ Du<T> operator*(T a, Du<T> b) { return Du<T>(a * b.v, a * b.d); }
template <class T> __device__ __forceinline__ Du<T> &operator+=(Du<T> &a, Du<T> b) { a.v += b.v; a.d += b.d; return a; }
template <class T> __device__ __forceinline__ Du<T> &operator-=(Du<T> &a, Du<T> b) { a.v -= b.v; a.d -= b.d; return a; }
// sine and cosine of an angle given with its first-order part
template <class T> __device__ __forceinline__ void sc_of(T x, T &s, T &c) { sincos_precise(x, &s, &c); }
template <class T> __device__ __forceinline__ void sc_of(Du<T> x, Du<T> &s, Du<T> &c)
{
    T sn, cs;
    sincos_precise(x.v, &sn, &cs);
    s = Du<T>(sn, cs * x.d);
    c = Du<T>(cs, -sn * x.d);
}

// Array bounds of the per-lane work areas: KB bodies and KN independent coordinates per cluster are template parameters of
// everything below.  <kMaxClusterBodies, kMaxClusterDof> (8, 4) is what the structured kernels cover and what the derivative
// parts are compiled for; <kBigClusterBodies, kBigClusterDof> serves clusters beyond those limits (plan.h, HostPlan::big_clusters
// -- the reference's parallel-chain benchmark family, Benchmarking/src/pinocchioHelpers.cpp:355-410): forward / inverse dynamics
// and the mass matrix only, G kept in the coupling slab instead of registers, private arrays in scratch memory -- slow, correct.
// constraint rows of an implicit cluster: 3 in the structured kernels (registers, closed-form inverses of K_d), kMaxConstraintRows = 6 in the wide
// ones (runtime loops over private arrays, K_d by elimination with partial pivoting): clusters with 4 - 6 rows -- two planar loops that share
// bodies, a spatial loop beside a planar one; the reference inverts a K_d of any size, GenericJoint.cpp:57-90 -- take the spanning-tree route
// like the clusters beyond 8 bodies / 4 coordinates (plan.cpp)
template <int KB>
constexpr int kMRof = KB > kMaxClusterBodies ? kMaxConstraintRows : 3;
// rows of the coupling slab per body of an implicit cluster with n independent coordinates: everything the derivatives need,
// or G alone for the clusters beyond the structured kernels' limits (capi.cpp sizes crow / n_cpl_rows with the same rule)
template <int KB>
__device__ __forceinline__ int cpl_stride(int n)
{
    return KB > kMaxClusterBodies ? n : n * (4 + n);
}
// G (k x n) of the implicit cluster at hand: registers, or -- big clusters -- the slab rows it is handed on in anyway
template <class T, int KB, int KN, bool SLAB = (KB > kMaxClusterBodies)>
struct GStore {
    T g[SLAB ? 1 : KB][SLAB ? 1 : KN];
    T *cc;
    int stride;
    __device__ __forceinline__ T get(int i, int a) const
    {
        if constexpr (SLAB) return cc[(size_t)(i * stride + a) * kWave];
        else return g[i][a];
    }
    __device__ __forceinline__ void set(int i, int a, T v)
    {
        if constexpr (SLAB) cc[(size_t)(i * stride + a) * kWave] = v;
        else g[i][a] = v;
    }
};

template <class S>
__device__ __forceinline__ void cross_s(const S (&a)[3], const S (&b)[3], S (&o)[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// ---- URDF+ position loops: K (want_K) or kappa = Kdot qd (otherwise), in the scalar type S ------------------------------
// sn / cs: sine and cosine of the k spanning angles; qd: the k spanning rates
// wave-uniform index into a private array WITHOUT taking the array to scratch memory: compare-and-select over the (few) entries
template <class S, int N>
__device__ __forceinline__ S pick(const S (&a)[N], int idx)
{
    // (every entry read unconditionally, then selects on VALUES: "if (idx == j) r = a[j]" is folded into one load at a selected
    // address, which is exactly the dynamic index this avoids)
    S r = a[0];
#pragma unroll
    for (int j = 1; j < N; j++) {
        const S e = a[j];
        r = idx == j ? e : r;
    }
    return r;
}
template <class S, int N>
__device__ __forceinline__ void place(S (&a)[N], int idx, S v)
{
#pragma unroll
    for (int j = 0; j < N; j++) {
        const S e = a[j];
        a[j] = idx == j ? v : e;
    }
}

// (clusters beyond the structured limits: runtime loops over private arrays of 48 entries -- scratch memory, slow, correct)
template <class T, class S, int KB>
__device__ void loop_position_eval_dyn(cptr<T> consts, cptr<BodyRec> bodies, const ClusterRec &c, cptr<int32_t> loops, int n_loops,
                                       const S *sn, const S *cs, const S *qd, bool want_K, S (&K)[kMRof<KB>][KB], S (&kap)[kMRof<KB>], S *phi = nullptr)
{
    // phi (with want_K): the constraint values -- predecessor point minus successor point along the constrained axes
    cptr<int32_t> lp = loops;
    int row0 = 0;
    for (int l = 0; l < n_loops; l++) {
        const int np = lp[0], ns = lp[1 + np], mask = lp[2 + np + ns];
        cptr<T> org = consts + c.dofs + 24 * l;
        S acc[3] = {S(T(0)), S(T(0)), S(T(0))};
        for (int side = 0; side < 2; side++) {
            cptr<int32_t> subs = side == 0 ? lp + 1 : lp + 2 + np;
            const int len = side == 0 ? np : ns;
            const T sgn = side == 0 ? T(1) : T(-1);
            S E[9], r[3], A[KB][3], O[KB][3];
#pragma unroll
            for (int i = 0; i < 9; i++) E[i] = S(i % 4 == 0 ? T(1) : T(0));
#pragma unroll
            for (int i = 0; i < 3; i++) r[i] = S(T(0));
            for (int t = 0; t < len; t++) {
                const int sub = subs[t];
                const BodyRec b = load_rec(bodies + (c.first_body + sub));
                cptr<T> C = consts + b.cofs;
                S Eb[9], En[9];
                const S s = sn[sub], co = cs[sub];
#pragma unroll
                for (int j = 0; j < 3; j++) {  // Rz(q) Et (canonical joint axes, plan.cpp)
                    Eb[j] = co * C[j] + s * C[3 + j];
                    Eb[3 + j] = co * C[3 + j] - s * C[j];
                    Eb[6 + j] = S(C[6 + j]);
                }
#pragma unroll
                for (int i = 0; i < 3; i++) r[i] += E[i] * C[9] + E[3 + i] * C[10] + E[6 + i] * C[11];
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) En[3 * i + j] = Eb[3 * i] * E[j] + Eb[3 * i + 1] * E[3 + j] + Eb[3 * i + 2] * E[6 + j];
#pragma unroll
                for (int i = 0; i < 9; i++) E[i] = En[i];
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    A[t][i] = E[6 + i];
                    O[t][i] = r[i];
                }
            }
            S p[3];
            cptr<T> og = org + 12 * side;
#pragma unroll
            for (int i = 0; i < 3; i++) p[i] = r[i] + E[i] * og[9] + E[3 + i] * og[10] + E[6 + i] * og[11];
            if (want_K) {
                if (phi) {
#pragma unroll
                    for (int i = 0; i < 3; i++) acc[i] += sgn * p[i];
                }
                for (int t = 0; t < len; t++) {
                    const S a[3] = {A[t][0], A[t][1], A[t][2]}, d[3] = {p[0] - O[t][0], p[1] - O[t][1], p[2] - O[t][2]};
                    S J[3];
                    cross_s(a, d, J);
                    int row = row0;
                    for (int ax = 0; ax < 3; ax++)
                        if (mask & (1 << ax)) {
                            K[row][subs[t]] = sgn * J[ax];
                            row++;
                        }
                }
            } else {
                S w[3], al[3], ao[3], op[3];
#pragma unroll
                for (int i = 0; i < 3; i++) w[i] = al[i] = ao[i] = op[i] = S(T(0));
                for (int t = 0; t <= len; t++) {
                    S a[3], o[3], qd_t = S(T(0));
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        a[i] = t < len ? A[t][i] : S(T(0));
                        o[i] = t < len ? O[t][i] : p[i];
                    }
                    if (t < len) qd_t = qd[subs[t]];
                    const S d[3] = {o[0] - op[0], o[1] - op[1], o[2] - op[2]};
                    S wd[3], wwd[3], ad[3];
                    cross_s(w, d, wd);
                    cross_s(w, wd, wwd);
                    cross_s(al, d, ad);
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        ao[i] += ad[i] + wwd[i];
                        op[i] = o[i];
                    }
                    const S aq[3] = {a[0] * qd_t, a[1] * qd_t, a[2] * qd_t};
                    S waq[3];
                    cross_s(w, aq, waq);
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        al[i] += waq[i];
                        w[i] += aq[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; i++) acc[i] += sgn * ao[i];
            }
        }
        int row = row0;
        for (int ax = 0; ax < 3; ax++)
            if (mask & (1 << ax)) {
                if (!want_K) kap[row] = acc[ax];
                else if (phi) phi[row] = acc[ax];
                row++;
            }
        row0 = row;
        lp += 3 + np + ns;
    }
}

// the structured clusters (at most kMaxClusterBodies bodies): every loop unrolled to its bound, every array in registers
template <class T, class S, int KB>
__device__ __forceinline__ void loop_position_eval_reg(cptr<T> consts, cptr<BodyRec> bodies, const ClusterRec &c, cptr<int32_t> loops,
                                                       int n_loops, const S (&sn)[KB], const S (&cs)[KB], const S (&qd)[KB], bool want_K,
                                                       S (&K)[kMRof<KB>][KB], S (&kap)[kMRof<KB>], S *phi)
{
    cptr<int32_t> lp = loops;
    int row0 = 0;
    for (int l = 0; l < n_loops; l++) {
        const int np = lp[0], ns = lp[1 + np], mask = lp[2 + np + ns];
        cptr<T> org = consts + c.dofs + 24 * l;
        S acc[3] = {S(T(0)), S(T(0)), S(T(0))};
        for (int side = 0; side < 2; side++) {
            cptr<int32_t> subs = side == 0 ? lp + 1 : lp + 2 + np;
            const int len = side == 0 ? np : ns;
            const T sgn = side == 0 ? T(1) : T(-1);
            S E[9], r[3], A[KB][3], O[KB][3];
#pragma unroll
            for (int i = 0; i < 9; i++) E[i] = S(i % 4 == 0 ? T(1) : T(0));
#pragma unroll
            for (int i = 0; i < 3; i++) r[i] = S(T(0));
#pragma unroll
            for (int t = 0; t < KB; t++) {
#pragma unroll
                for (int i = 0; i < 3; i++) A[t][i] = O[t][i] = S(T(0));
                if (t < len) {
                    const int sub = subs[t];
                    const BodyRec b = load_rec(bodies + (c.first_body + sub));
                    cptr<T> C = consts + b.cofs;
                    S Eb[9], En[9];
                    const S s = pick(sn, sub), co = pick(cs, sub);
#pragma unroll
                    for (int j = 0; j < 3; j++) {  // Rz(q) Et (canonical joint axes, plan.cpp)
                        Eb[j] = co * C[j] + s * C[3 + j];
                        Eb[3 + j] = co * C[3 + j] - s * C[j];
                        Eb[6 + j] = S(C[6 + j]);
                    }
#pragma unroll
                    for (int i = 0; i < 3; i++) r[i] += E[i] * C[9] + E[3 + i] * C[10] + E[6 + i] * C[11];
#pragma unroll
                    for (int i = 0; i < 3; i++)
#pragma unroll
                        for (int j = 0; j < 3; j++) En[3 * i + j] = Eb[3 * i] * E[j] + Eb[3 * i + 1] * E[3 + j] + Eb[3 * i + 2] * E[6 + j];
#pragma unroll
                    for (int i = 0; i < 9; i++) E[i] = En[i];
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        A[t][i] = E[6 + i];
                        O[t][i] = r[i];
                    }
                }
            }
            S p[3];
            cptr<T> og = org + 12 * side;
#pragma unroll
            for (int i = 0; i < 3; i++) p[i] = r[i] + E[i] * og[9] + E[3 + i] * og[10] + E[6 + i] * og[11];
            if (want_K) {
                if (phi) {
#pragma unroll
                    for (int i = 0; i < 3; i++) acc[i] += sgn * p[i];
                }
#pragma unroll
                for (int t = 0; t < KB; t++) {
                    if (t >= len) continue;
                    const S a[3] = {A[t][0], A[t][1], A[t][2]}, d[3] = {p[0] - O[t][0], p[1] - O[t][1], p[2] - O[t][2]};
                    S J[3];
                    cross_s(a, d, J);
                    int row = row0;
                    const int sub = subs[t];
#pragma unroll
                    for (int ax = 0; ax < 3; ax++)
                        if (mask & (1 << ax)) {
#pragma unroll
                            for (int rr = 0; rr < kMRof<KB>; rr++) place(K[rr], rr == row ? sub : -1, sgn * J[ax]);
                            row++;
                        }
                }
            } else {
                S w[3], al[3], ao[3], op[3];
#pragma unroll
                for (int i = 0; i < 3; i++) w[i] = al[i] = ao[i] = op[i] = S(T(0));
#pragma unroll
                for (int t = 0; t <= KB; t++) {
                    if (t > len) continue;
                    S a[3], o[3], qd_t = S(T(0));
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        a[i] = S(T(0));
                        o[i] = p[i];
                    }
                    if constexpr (true) {
                        if (t < KB && t < len) {
#pragma unroll
                            for (int i = 0; i < 3; i++) {
                                a[i] = A[t < KB ? t : 0][i];
                                o[i] = O[t < KB ? t : 0][i];
                            }
                            qd_t = pick(qd, subs[t]);
                        }
                    }
                    const S d[3] = {o[0] - op[0], o[1] - op[1], o[2] - op[2]};
                    S wd[3], wwd[3], ad[3];
                    cross_s(w, d, wd);
                    cross_s(w, wd, wwd);
                    cross_s(al, d, ad);
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        ao[i] += ad[i] + wwd[i];
                        op[i] = o[i];
                    }
                    const S aq[3] = {a[0] * qd_t, a[1] * qd_t, a[2] * qd_t};
                    S waq[3];
                    cross_s(w, aq, waq);
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        al[i] += waq[i];
                        w[i] += aq[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; i++) acc[i] += sgn * ao[i];
            }
        }
        int row = row0;
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
            if (mask & (1 << ax)) {
                if (!want_K) place(kap, row, acc[ax]);
                else if (phi) phi[row] = acc[ax];
                row++;
            }
        row0 = row;
        lp += 3 + np + ns;
    }
}

template <class T, class S, int KB>
__device__ __forceinline__ void loop_position_eval(cptr<T> consts, cptr<BodyRec> bodies, const ClusterRec &c, cptr<int32_t> loops, int n_loops,
                                                   const S (&sn)[KB], const S (&cs)[KB], const S (&qd)[KB], bool want_K, S (&K)[kMRof<KB>][KB],
                                                   S (&kap)[kMRof<KB>], S *phi = nullptr)
{
    if constexpr (KB > kMaxClusterBodies) loop_position_eval_dyn<T, S, KB>(consts, bodies, c, loops, n_loops, sn, cs, qd, want_K, K, kap, phi);
    else loop_position_eval_reg<T, S, KB>(consts, bodies, c, loops, n_loops, sn, cs, qd, want_K, K, kap, phi);
}

// ---- trig-polynomial phi (plan.cpp: ints [n_args, per row: n_terms, per term: n_factors, (type, argument)...], constants
// [per distinct argument w[k], b][per term coef]) ---------------------------------------------------------------------------
// The DISTINCT arguments a = w . q + b are evaluated once per call -- the Tello hip differential has 4 in the 30 factors of its two
// rows, and sincos_precise is what an evaluation costs -- and parked in LDS ([slot][lane]: the factors pick them by a wave-uniform
// index that a private array would take to scratch memory).  tl: this lane's column of the work area (kTrigLdsSlots rows of T), or
// nullptr / more than kTrigArgsLds arguments: every factor evaluates its own.  fresh = false: [a, sin, cos] of the previous call
// with the same q are still there (the kappa pass after the K pass), only the rates a' = w . qd are new.
constexpr int kTrigArgsLds = 8;
constexpr int kTrigLdsSlots = kTrigArgsLds * 4 * 2;  // [a, a', sin, cos] per argument, two numbers each when they are duals
template <class T>
__device__ __forceinline__ void tl_put(T *tl, int slot, T v) { tl[(size_t)(2 * slot) * kWave] = v; }
template <class T>
__device__ __forceinline__ void tl_put(T *tl, int slot, Du<T> v)
{
    tl[(size_t)(2 * slot) * kWave] = v.v;
    tl[(size_t)(2 * slot + 1) * kWave] = v.d;
}
template <class T>
__device__ __forceinline__ void tl_get(const T *tl, int slot, T &v) { v = tl[(size_t)(2 * slot) * kWave]; }
template <class T>
__device__ __forceinline__ void tl_get(const T *tl, int slot, Du<T> &v)
{
    v.v = tl[(size_t)(2 * slot) * kWave];
    v.d = tl[(size_t)(2 * slot + 1) * kWave];
}

template <class T, class S, int KB>
__device__ __forceinline__ void trig_poly_eval_s(cptr<T> consts, const ClusterRec &c, cptr<int32_t> prog, const S (&q)[KB], const S (&qd)[KB],
                                                 bool want_K, S (&K)[kMRof<KB>][KB], S (&kap)[kMRof<KB>], T *tl = nullptr, bool fresh = true)
{
    const int k = c.k;
    cptr<int32_t> ip = prog;
    const int n_args = *ip++;
    cptr<T> ap = consts + c.dofs;
    cptr<T> cp = ap + n_args * (k + 1);
    const bool cached = tl != nullptr && n_args <= kTrigArgsLds;
    if (cached) {
        for (int arg = 0; arg < n_args; arg++) {
            cptr<T> w = ap + arg * (k + 1);
            if (fresh) {
                S a = S(w[k]);
#pragma unroll
                for (int j = 0; j < KB; j++)
                    if (j < k) a += q[j] * w[j];
                S sn, cs;
                sc_of(a, sn, cs);
                tl_put(tl, 4 * arg, a);
                tl_put(tl, 4 * arg + 2, sn);
                tl_put(tl, 4 * arg + 3, cs);
            }
            if (!want_K) {
                S adot = S(T(0));
#pragma unroll
                for (int j = 0; j < KB; j++)
                    if (j < k) adot += qd[j] * w[j];
                tl_put(tl, 4 * arg + 1, adot);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < kMRof<KB>; r++) {
        if (r >= c.rows) break;
        const int nt = *ip++;
        S Krow[KB], kd = S(T(0));
#pragma unroll
        for (int j = 0; j < KB; j++) Krow[j] = S(T(0));
        for (int t = 0; t < nt; t++) {
            const int nf = *ip++;
            const T coef = *cp++;
            S f0[4], f1[4], f2[4], ad[4];
            cptr<T> wv[4];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                f0[f] = S(T(1)); f1[f] = S(T(0)); f2[f] = S(T(0)); ad[f] = S(T(0));
                wv[f] = ap;
                if (f < nf) {
                    const int type = ip[0], arg = ip[1];
                    ip += 2;
                    cptr<T> w = ap + arg * (k + 1);
                    wv[f] = w;
                    S a, adot = S(T(0)), sn, cs;
                    if (cached) {
                        tl_get(tl, 4 * arg, a);
                        if (!want_K) tl_get(tl, 4 * arg + 1, adot);
                        tl_get(tl, 4 * arg + 2, sn);
                        tl_get(tl, 4 * arg + 3, cs);
                    } else {
                        a = S(w[k]);
#pragma unroll
                        for (int j = 0; j < KB; j++)
                            if (j < k) {
                                a += q[j] * w[j];
                                adot += qd[j] * w[j];
                            }
                        sc_of(a, sn, cs);
                    }
                    ad[f] = adot;
                    if (type == 1) { f0[f] = sn; f1[f] = cs; f2[f] = -sn; }
                    else if (type == 2) { f0[f] = cs; f1[f] = -sn; f2[f] = -cs; }
                    else { f0[f] = a; f1[f] = S(T(1)); f2[f] = S(T(0)); }
                }
            }
#pragma unroll
            for (int f = 0; f < 4; f++) {
                if (f >= nf) continue;
                S others = S(coef);
#pragma unroll
                for (int h = 0; h < 4; h++)
                    if (h != f) others = others * f0[h];
                if (want_K) {
                    const S of1 = others * f1[f];
#pragma unroll
                    for (int j = 0; j < KB; j++)
                        if (j < k) Krow[j] += of1 * wv[f][j];
                } else {
                    kd += others * f2[f] * ad[f] * ad[f];
#pragma unroll
                    for (int h = 0; h < 4; h++)
                        if (h != f && h < nf) {
                            S rest = S(coef);
#pragma unroll
                            for (int m2 = 0; m2 < 4; m2++)
                                if (m2 != f && m2 != h) rest = rest * f0[m2];
                            kd += rest * f1[f] * ad[f] * f1[h] * ad[h];
                        }
                }
            }
        }
        if (want_K) {
#pragma unroll
            for (int j = 0; j < KB; j++)
                if (j < k) K[r][j] = Krow[j];
        } else {
            kap[r] = kd;
        }
    }
}

// K or kappa of an implicit cluster at the spanning state (q, qd), scalar type S
template <class T, class S, int KB>
__device__ __forceinline__ void constraint_eval(cptr<T> consts, cptr<BodyRec> bodies, cptr<int32_t> cints, const ClusterRec &c,
                                                const S (&q)[KB], const S (&qd)[KB], bool want_K, S (&K)[kMRof<KB>][KB], S (&kap)[kMRof<KB>],
                                                T *tl = nullptr, bool fresh = true)
{
    cptr<int32_t> ip = cints + c.iofs;
    const int hdr0 = ip[0], n_ind = ip[1];
    cptr<int32_t> payload = ip + 3 + n_ind + c.rows;
    if (c.cons_type == 0) {
        S sn[KB], cs[KB];
        if constexpr (KB > kMaxClusterBodies) {
            for (int j = 0; j < c.k; j++) sc_of(q[j], sn[j], cs[j]);
        } else {
#pragma unroll
            for (int j = 0; j < KB; j++) {
                sn[j] = S(T(0));
                cs[j] = S(T(1));
                if (j < c.k) sc_of(q[j], sn[j], cs[j]);
            }
        }
        loop_position_eval<T, S, KB>(consts, bodies, c, payload, hdr0, sn, cs, qd, want_K, K, kap);
    } else {
        trig_poly_eval_s<T, S, KB>(consts, c, payload, q, qd, want_K, K, kap, tl, fresh);
    }
}

template <class T, int M>
__device__ __forceinline__ void inv_rows(int R, const T (&A)[M][M], T (&Ai)[M][M])
{
    for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++) Ai[i][j] = 0;
    if (R == 1) {
        Ai[0][0] = T(1) / A[0][0];
    } else if (R == 2) {
        const T id = T(1) / (A[0][0] * A[1][1] - A[0][1] * A[1][0]);
        Ai[0][0] = A[1][1] * id; Ai[0][1] = -A[0][1] * id;
        Ai[1][0] = -A[1][0] * id; Ai[1][1] = A[0][0] * id;
    } else if (R == 3) {
        const T c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1];
        const T c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2];
        const T c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
        const T id = T(1) / (A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02);
        Ai[0][0] = c00 * id; Ai[1][0] = c01 * id; Ai[2][0] = c02 * id;
        Ai[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * id;
        Ai[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * id;
        Ai[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * id;
        Ai[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * id;
        Ai[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * id;
        Ai[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * id;
    } else if constexpr (M > 3) {
        // 4 .. M rows: Gauss-Jordan elimination with partial pivoting on [A | 1] (the wide kernels only: runtime loops over private arrays).
        // A singular K_d leaves Inf / NaN behind, as the closed forms do.
        T W[M][M];
        for (int i = 0; i < R; i++)
            for (int j = 0; j < R; j++) {
                W[i][j] = A[i][j];
                Ai[i][j] = i == j ? T(1) : T(0);
            }
        for (int c = 0; c < R; c++) {
            int piv = c;
            T best = W[c][c] < 0 ? -W[c][c] : W[c][c];
            for (int i = c + 1; i < R; i++) {
                const T a = W[i][c] < 0 ? -W[i][c] : W[i][c];
                if (a > best) { best = a; piv = i; }
            }
            if (piv != c)
                for (int j = 0; j < R; j++) {
                    const T t1 = W[c][j]; W[c][j] = W[piv][j]; W[piv][j] = t1;
                    const T t2 = Ai[c][j]; Ai[c][j] = Ai[piv][j]; Ai[piv][j] = t2;
                }
            const T ip = T(1) / W[c][c];
            for (int j = 0; j < R; j++) {
                W[c][j] *= ip;
                Ai[c][j] *= ip;
            }
            for (int i = 0; i < R; i++) {
                if (i == c) continue;
                const T f = W[i][c];
                for (int j = 0; j < R; j++) {
                    W[i][j] -= f * W[c][j];
                    Ai[i][j] -= f * Ai[c][j];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One implicit cluster of at most kMaxClusterBodies bodies / kMaxClusterDof independent coordinates in kernel 1: spanning state, G, g
// and -- want_d -- their first-order parts along every independent coordinate (the formulas at the top of this file).  Every loop runs
// to its compile-time bound under a wave-uniform guard and the dependent / independent coordinate numbers go through pick / place,
// so that K, G and the dual work areas stay in registers (the runtime loops this replaces kept 1.2 KB per lane in scratch memory
// and every multiply-add of the small solves was a dependent scratch round trip).
// ---------------------------------------------------------------------------------------------------------------
template <class T, int KB, int KN>
__device__ __forceinline__ void manifold_implicit_cluster(cptr<T> consts, cptr<BodyRec> bodies, cptr<int32_t> cints, cptr<int32_t> span_q,
                                                          cptr<int32_t> span_v, const ClusterRec &cr, const T *qs, const T *qds, const T *ydds,
                                                          T *oq, T *ov, T *oa, T *cc, bool live, int want_d, T *tl)
{
    const int k = cr.k, n = cr.n, rows = cr.rows;
    cptr<int32_t> ip = cints + cr.iofs;
    const int n_ind = ip[1];
    int ind[KN], dep[kMRof<KB>];
#pragma unroll
    for (int a = 0; a < KN; a++) ind[a] = a < n ? ip[2 + a] : -1;
#pragma unroll
    for (int r = 0; r < kMRof<KB>; r++) dep[r] = r < rows ? ip[3 + n_ind + r] : -1;
    T qv[KB], yd[KN], yddv[KN];
#pragma unroll
    for (int j = 0; j < KB; j++) qv[j] = j < k ? qs[cr.q_index + j] : T(0);
#pragma unroll
    for (int a = 0; a < KN; a++) {
        yd[a] = a < n ? qds[cr.v_index + a] : T(0);
        yddv[a] = (a < n && ydds) ? ydds[cr.v_index + a] : T(0);
    }
    T K[kMRof<KB>][KB], kap[kMRof<KB>], zero[KB];
#pragma unroll
    for (int r = 0; r < kMRof<KB>; r++) {
        kap[r] = 0;
#pragma unroll
        for (int j = 0; j < KB; j++) K[r][j] = 0;
    }
#pragma unroll
    for (int j = 0; j < KB; j++) zero[j] = 0;
    constraint_eval<T, T, KB>(consts, bodies, cints, cr, qv, zero, true, K, kap, tl, true);
    T Kd[kMRof<KB>][kMRof<KB>], Kdi[kMRof<KB>][kMRof<KB>], qdv[KB], gv[KB], G[KB][KN];
#pragma unroll
    for (int r = 0; r < kMRof<KB>; r++)
#pragma unroll
        for (int j = 0; j < kMRof<KB>; j++) Kd[r][j] = (r < rows && j < rows) ? pick(K[r], dep[j]) : T(r == j);
    inv_rows(rows, Kd, Kdi);
#pragma unroll
    for (int i = 0; i < KB; i++) {
        gv[i] = 0;
#pragma unroll
        for (int a = 0; a < KN; a++) G[i][a] = (a < n && i == ind[a]) ? T(1) : T(0);
    }
#pragma unroll
    for (int r = 0; r < kMRof<KB>; r++)
#pragma unroll
        for (int a = 0; a < KN; a++) {
            if (r >= rows || a >= n) continue;
            T sum = 0;
#pragma unroll
            for (int j = 0; j < kMRof<KB>; j++)
                if (j < rows) sum += Kdi[r][j] * pick(K[j], ind[a]);
#pragma unroll
            for (int i = 0; i < KB; i++) G[i][a] = i == dep[r] ? -sum : G[i][a];
        }
#pragma unroll
    for (int i = 0; i < KB; i++) {
        T sum = 0;
#pragma unroll
        for (int a = 0; a < KN; a++) sum += G[i][a] * yd[a];
        qdv[i] = sum;
    }
    constraint_eval<T, T, KB>(consts, bodies, cints, cr, qv, qdv, false, K, kap, tl, false);  // (K is not touched: want_K false)
#pragma unroll
    for (int r = 0; r < kMRof<KB>; r++) {
        if (r >= rows) continue;
        T sum = 0;
#pragma unroll
        for (int j = 0; j < kMRof<KB>; j++)
            if (j < rows) sum += Kdi[r][j] * kap[j];
        place(gv, dep[r], -sum);
    }
    const int stride = cpl_stride<kMaxClusterBodies>(n);
#pragma unroll
    for (int i = 0; i < KB; i++) {
        if (i >= k) continue;
        if (live) {
            oq[span_q[cr.first_body + i]] = qv[i];
            ov[span_v[cr.first_body + i]] = qdv[i];
            if (oa) {
                T sum = gv[i];
#pragma unroll
                for (int a = 0; a < KN; a++) sum += G[i][a] * yddv[a];
                oa[span_v[cr.first_body + i]] = sum;
            }
        }
#pragma unroll
        for (int a = 0; a < KN; a++)
            if (a < n) cc[(size_t)(i * stride + a) * kWave] = G[i][a];
    }
    if (!want_d) return;
    // ---- first-order parts along every independent coordinate ----
    for (int i = 0; i < k; i++)
        for (int j = n; j < stride; j++) cc[(size_t)(i * stride + j) * kWave] = 0;
    for (int a = 0; a < n; a++) {
        Du<T> qD[KB], qdD[KB], KD[kMRof<KB>][KB], kapD[kMRof<KB>];
#pragma unroll
        for (int j = 0; j < KB; j++) {
            qD[j] = Du<T>(qv[j], pick(G[j], a));
            qdD[j] = Du<T>(T(0));
        }
#pragma unroll
        for (int r = 0; r < kMRof<KB>; r++) {
            kapD[r] = Du<T>(T(0));
#pragma unroll
            for (int j = 0; j < KB; j++) KD[r][j] = Du<T>(T(0));
        }
        constraint_eval<T, Du<T>, KB>(consts, bodies, cints, cr, qD, qdD, true, KD, kapD, tl, true);
        // G' (dependent rows) = -Kd^-1 K' G ;  d g / d yd_a = -2 Kd^-1 K' qd_s
        T Gp[kMRof<KB>][KN], KpQd[kMRof<KB>], qdp[KB], KpG[kMRof<KB>][KN];
#pragma unroll
        for (int r = 0; r < kMRof<KB>; r++) {
            T sum = 0;
#pragma unroll
            for (int j = 0; j < KB; j++) sum += KD[r][j].d * qdv[j];
            KpQd[r] = sum;
#pragma unroll
            for (int b2 = 0; b2 < KN; b2++) {
                T kg = 0;  // (K' G)[r][b2]
#pragma unroll
                for (int j = 0; j < KB; j++) kg += KD[r][j].d * G[j][b2];
                KpG[r][b2] = kg;
            }
        }
#pragma unroll
        for (int j = 0; j < KB; j++) qdp[j] = 0;
#pragma unroll
        for (int r = 0; r < kMRof<KB>; r++)
#pragma unroll
            for (int b2 = 0; b2 < KN; b2++) {
                Gp[r][b2] = 0;
                if (r >= rows || b2 >= n) continue;
                T sum = 0;
#pragma unroll
                for (int r2 = 0; r2 < kMRof<KB>; r2++)
                    if (r2 < rows) sum += Kdi[r][r2] * KpG[r2][b2];
                Gp[r][b2] = -sum;
#pragma unroll
                for (int i = 0; i < KB; i++) qdp[i] += i == dep[r] ? -sum * yd[b2] : T(0);
            }
#pragma unroll
        for (int j = 0; j < KB; j++) qdD[j] = Du<T>(qdv[j], qdp[j]);
        constraint_eval<T, Du<T>, KB>(consts, bodies, cints, cr, qD, qdD, false, KD, kapD, tl, false);  // (KD is not touched)
        T gdep[kMRof<KB>];
#pragma unroll
        for (int j = 0; j < kMRof<KB>; j++) gdep[j] = j < rows ? pick(gv, dep[j]) : T(0);
#pragma unroll
        for (int r = 0; r < kMRof<KB>; r++) {
            if (r >= rows) continue;
            const int i = dep[r];
            T gy = 0, gyd = 0, ay = 0, by = 0;
#pragma unroll
            for (int r2 = 0; r2 < kMRof<KB>; r2++) {
                if (r2 >= rows) continue;
                T kdg = 0;  // (K'_d g_dep)[r2]
#pragma unroll
                for (int j = 0; j < kMRof<KB>; j++)
                    if (j < rows) kdg += pick(KD[r2], dep[j]).d * gdep[j];
                gy += Kdi[r][r2] * (kapD[r2].d + kdg);
                gyd += Kdi[r][r2] * KpQd[r2];
            }
            gy = -gy;
            gyd = T(-2) * gyd;
#pragma unroll
            for (int b2 = 0; b2 < KN; b2++) {
                if (b2 >= n) continue;
                ay += Gp[r][b2] * yd[b2];
                by += Gp[r][b2] * yddv[b2];
                cc[(size_t)(i * stride + 4 * n + a * n + b2) * kWave] = Gp[r][b2];
            }
            cc[(size_t)(i * stride + n + a) * kWave] = ay;
            cc[(size_t)(i * stride + 2 * n + a) * kWave] = by + gy;
            cc[(size_t)(i * stride + 3 * n + a) * kWave] = gyd;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel 1: spanning state and coupling of every cluster, one state per lane.
//   q_s [B][nq_s], qd_s / qdd_s [B][nv_s]: the state of the spanning model (row-major: what its kernels take)
//   cpl [tile][row][lane]: per implicit cluster, from row crow[c], per body i (stride n (4 + n)):
//       [G row n][(G_a' yd)_i, a < n][(G_a' ydd + d g / d y_a)_i][(d g / d yd_a)_i][G_a'[i][b], a-major]
// span_q / span_v: first spanning position / velocity index of every body; want_d = 0: G rows only (mass matrix)
// ---------------------------------------------------------------------------------------------------------------
template <class T, int KB, int KN>
__global__ __launch_bounds__(kWave, 1) void manifold_constraint_kernel(DevPlan<T> DP, int n_clusters, const int32_t *__restrict__ span_q_,
                                                                     const int32_t *__restrict__ span_v_, const int32_t *__restrict__ crow_,
                                                                     int nq_s, int nv_s, int n_cpl_rows, int want_d,
                                                                     const T *__restrict__ q, const T *__restrict__ qd,
                                                                     const T *__restrict__ ydd, T *__restrict__ q_s, T *__restrict__ qd_s,
                                                                     T *__restrict__ qdd_s, T *__restrict__ cpl, size_t B)
{
    cptr<ClusterRec> clusters = (cptr<ClusterRec>)DP.clusters;
    cptr<BodyRec> bodies = (cptr<BodyRec>)DP.bodies;
    cptr<T> consts = (cptr<T>)DP.consts;
    cptr<int32_t> cints = (cptr<int32_t>)DP.cints;
    cptr<int32_t> span_q = (cptr<int32_t>)span_q_, span_v = (cptr<int32_t>)span_v_, crow = (cptr<int32_t>)crow_;
    const int lane = threadIdx.x, nq = DP.nq, nv = DP.nv;
    // (one wavefront per workgroup: trig_poly_eval_s.  DYNAMIC LDS, asked for only by plans that have a trig-polynomial cluster --
    // launch_manifold_constraint: a static array cost every instantiation 16 / 32 KB per workgroup, loop-position models included)
    T *tl = reinterpret_cast<T *>(grbda_smem) + lane;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r0 = tile * kWave + lane;
        const bool live = r0 < B;
        const size_t st = live ? r0 : B - 1;
        const T *qs = q + st * (size_t)nq, *qds = qd + st * (size_t)nv, *ydds = ydd ? ydd + st * (size_t)nv : nullptr;
        T *oq = q_s + st * (size_t)nq_s, *ov = qd_s + st * (size_t)nv_s, *oa = qdd_s ? qdd_s + st * (size_t)nv_s : nullptr;
        T *cp = cpl + (tile * (size_t)n_cpl_rows) * kWave + lane;
        for (int c = 0; c < n_clusters; c++) {
            const ClusterRec cr = load_rec(clusters + c);
            const int sq0 = span_q[cr.first_body], sv0 = span_v[cr.first_body];
            if (cr.kind == CK_FREE) {
                const int npos = DP.ori_repr == 0 ? 7 : 6;
                if (live) {
                    for (int j = 0; j < npos; j++) oq[sq0 + j] = qs[cr.q_index + j];
                    for (int j = 0; j < 6; j++) {
                        ov[sv0 + j] = qds[cr.v_index + j];
                        if (oa) oa[sv0 + j] = ydds ? ydds[cr.v_index + j] : T(0);
                    }
                }
                continue;
            }
            const int k = cr.k, n = cr.n;
            if (cr.kind == CK_STATIC) {
                for (int i = 0; i < k; i++) {
                    const BodyRec b = load_rec(bodies + (cr.first_body + i));
                    cptr<T> G = consts + b.cofs + kBodyConstFixed;
                    T a = 0, v = 0, w = 0;
                    for (int j = 0; j < n; j++) {
                        a += G[j] * qs[cr.q_index + j];
                        v += G[j] * qds[cr.v_index + j];
                        if (ydds) w += G[j] * ydds[cr.v_index + j];
                        // (wide variant: the projection and the apply kernels read every cluster's G from the slab -- coalesced vector
                        // loads that pipeline, where table lookups on the scalar unit would serialise their inner loops)
                        if constexpr (KB > kMaxClusterBodies) cp[(size_t)(crow[c] + i * n + j) * kWave] = G[j];
                    }
                    if (live) {
                        oq[span_q[cr.first_body + i]] = a;
                        ov[span_v[cr.first_body + i]] = v;
                        if (oa) oa[span_v[cr.first_body + i]] = w;
                    }
                }
                continue;
            }
            // ---- implicit cluster ----
            if constexpr (KB <= kMaxClusterBodies) {
                manifold_implicit_cluster<T, KB, KN>(consts, bodies, cints, span_q, span_v, cr, qs, qds, ydds, oq, ov, oa, cp + (size_t)crow[c] * kWave,
                                                     live, want_d, tl);
                continue;
            }
            // (clusters beyond the structured limits: runtime loops, G in the coupling slab, no derivative parts)
            cptr<int32_t> ip = cints + cr.iofs;
            const int n_ind = ip[1], rows = cr.rows;
            cptr<int32_t> ind = ip + 2, dep = ip + 3 + n_ind;
            T qv[KB], yd[KN], yddv[KN];
            for (int j = 0; j < KB; j++) qv[j] = j < k ? qs[cr.q_index + j] : T(0);
            for (int a = 0; a < KN; a++) {
                yd[a] = a < n ? qds[cr.v_index + a] : T(0);
                yddv[a] = (a < n && ydds) ? ydds[cr.v_index + a] : T(0);
            }
            T K[kMRof<KB>][KB], kap[kMRof<KB>], zero[KB];
            for (int r = 0; r < kMRof<KB>; r++) {
                kap[r] = 0;
                for (int j = 0; j < KB; j++) K[r][j] = 0;
            }
            for (int j = 0; j < KB; j++) zero[j] = 0;
            constraint_eval<T, T, KB>(consts, bodies, cints, cr, qv, zero, true, K, kap, tl, true);
            T Kd[kMRof<KB>][kMRof<KB>], Kdi[kMRof<KB>][kMRof<KB>], qdv[KB], gv[KB];
            const int stride = cpl_stride<KB>(n);
            T *cc = cp + (size_t)crow[c] * kWave;
            GStore<T, KB, KN> G;
            G.cc = cc;
            G.stride = stride;
            for (int r = 0; r < kMRof<KB>; r++)
                for (int j = 0; j < kMRof<KB>; j++) Kd[r][j] = (r < rows && j < rows) ? K[r][dep[j]] : T(r == j);
            inv_rows(rows, Kd, Kdi);
            for (int i = 0; i < k; i++) {
                gv[i] = 0;
                for (int a = 0; a < n; a++) G.set(i, a, T(0));
            }
            for (int a = 0; a < n; a++) G.set(ind[a], a, T(1));
            for (int r = 0; r < rows; r++)
                for (int a = 0; a < n; a++) {
                    T s = 0;
                    for (int j = 0; j < rows; j++) s += Kdi[r][j] * K[j][ind[a]];
                    G.set(dep[r], a, -s);
                }
            for (int i = 0; i < k; i++) {
                T s = 0;
                for (int a = 0; a < n; a++) s += G.get(i, a) * yd[a];
                qdv[i] = s;
            }
            constraint_eval<T, T, KB>(consts, bodies, cints, cr, qv, qdv, false, K, kap, tl, false);  // (K is not touched: want_K false)
            for (int r = 0; r < rows; r++) {
                T s = 0;
                for (int j = 0; j < rows; j++) s += Kdi[r][j] * kap[j];
                gv[dep[r]] = -s;
            }
            if (live) {
                for (int i = 0; i < k; i++) {
                    oq[span_q[cr.first_body + i]] = qv[i];
                    ov[span_v[cr.first_body + i]] = qdv[i];
                    if (oa) {
                        T s = gv[i];
                        for (int a = 0; a < n; a++) s += G.get(i, a) * yddv[a];
                        oa[span_v[cr.first_body + i]] = s;
                    }
                }
            }
            if constexpr (KB <= kMaxClusterBodies) {
                for (int i = 0; i < k; i++)
                    for (int a = 0; a < n; a++) cc[(size_t)(i * stride + a) * kWave] = G.g[i][a];
            }
            if (!want_d || KB > kMaxClusterBodies) continue;
            // ---- first-order parts along every independent coordinate ----
            for (int i = 0; i < k; i++)
                for (int j = n; j < stride; j++) cc[(size_t)(i * stride + j) * kWave] = 0;
            for (int a = 0; a < n; a++) {
                Du<T> qD[KB], qdD[KB], KD[kMRof<KB>][KB], kapD[kMRof<KB>];
                for (int j = 0; j < KB; j++) {
                    qD[j] = Du<T>(qv[j], j < k ? G.get(j, a) : T(0));
                    qdD[j] = Du<T>(T(0));
                }
                for (int r = 0; r < kMRof<KB>; r++) {
                    kapD[r] = Du<T>(T(0));
                    for (int j = 0; j < KB; j++) KD[r][j] = Du<T>(T(0));
                }
                constraint_eval<T, Du<T>, KB>(consts, bodies, cints, cr, qD, qdD, true, KD, kapD, tl, true);
                // G' (dependent rows) = -Kd^-1 K' G ;  d g / d yd_a = -2 Kd^-1 K' qd_s
                T Gp[kMRof<KB>][KN], KpQd[kMRof<KB>], qdp[KB];
                for (int r = 0; r < rows; r++) {
                    T s = 0;
                    for (int j = 0; j < k; j++) s += KD[r][j].d * qdv[j];
                    KpQd[r] = s;
                }
                for (int j = 0; j < KB; j++) qdp[j] = 0;
                for (int r = 0; r < rows; r++)
                    for (int b2 = 0; b2 < n; b2++) {
                        T s = 0;
                        for (int r2 = 0; r2 < rows; r2++) {
                            T kg = 0;  // (K' G)[r2][b2]
                            for (int j = 0; j < k; j++) kg += KD[r2][j].d * G.get(j, b2);
                            s += Kdi[r][r2] * kg;
                        }
                        Gp[r][b2] = -s;
                        qdp[dep[r]] += -s * yd[b2];
                    }
                for (int j = 0; j < KB; j++) qdD[j] = Du<T>(j < k ? qdv[j] : T(0), qdp[j]);
                constraint_eval<T, Du<T>, KB>(consts, bodies, cints, cr, qD, qdD, false, KD, kapD, tl, false);  // (KD is not touched)
                for (int r = 0; r < rows; r++) {
                    const int i = dep[r];
                    T gy = 0, gyd = 0, ay = 0, by = 0;
                    for (int r2 = 0; r2 < rows; r2++) {
                        T kdg = 0;  // (K'_d g_dep)[r2]
                        for (int j = 0; j < rows; j++) kdg += KD[r2][dep[j]].d * gv[dep[j]];
                        gy += Kdi[r][r2] * (kapD[r2].d + kdg);
                        gyd += Kdi[r][r2] * KpQd[r2];
                    }
                    gy = -gy;
                    gyd = T(-2) * gyd;
                    for (int b2 = 0; b2 < n; b2++) {
                        ay += Gp[r][b2] * yd[b2];
                        by += Gp[r][b2] * yddv[b2];
                        cc[(size_t)(i * stride + 4 * n + a * n + b2) * kWave] = Gp[r][b2];
                    }
                    cc[(size_t)(i * stride + n + a) * kWave] = ay;
                    cc[(size_t)(i * stride + 2 * n + a) * kWave] = by + gy;
                    cc[(size_t)(i * stride + 3 * n + a) * kWave] = gyd;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel 2: projection onto the independent coordinates, one state per lane.
//   Aq, Av: d tau_s / d q_s, d tau_s / d qd_s of the spanning model, Hs: its joint-space inertia -- rnea_deriv_kernel's packed
//   layout interleaved by 64: [tile][entry][lane]; tau_s [B][nv_s]; rel_s: DerivProgram::related of the spanning model
//   outputs Dq, Dqd, H in the layout spd_solve reads for the model's own nv (packed runs / packed lower rows, interleaved by IL)
// mode 0: all three; mode 1: H only (Aq, Av, tau_s unused)
// ---------------------------------------------------------------------------------------------------------------
template <class T, int IL, int KB, int KN>
__global__ __launch_bounds__(kWave, 1) void manifold_project_kernel(DevPlan<T> DP, int n_clusters, const int32_t *__restrict__ span_v_,
                                                                  const int32_t *__restrict__ crow_, const uint64_t *__restrict__ rel_,
                                                                  const uint64_t *__restrict__ rel_s_, int nv_s, int n_cpl_rows, int mode,
                                                                  const T *__restrict__ Aq, const T *__restrict__ Av,
                                                                  const T *__restrict__ Hs, const T *__restrict__ tau_s,
                                                                  const T *__restrict__ cpl, T *__restrict__ Dq, T *__restrict__ Dqd,
                                                                  T *__restrict__ H, size_t B)
{
    cptr<ClusterRec> clusters = (cptr<ClusterRec>)DP.clusters;
    cptr<BodyRec> bodies = (cptr<BodyRec>)DP.bodies;
    cptr<T> consts = (cptr<T>)DP.consts;
    cptr<int32_t> span_v = (cptr<int32_t>)span_v_, crow = (cptr<int32_t>)crow_;
    cptr<uint64_t> rel = (cptr<uint64_t>)rel_, rel_s = (cptr<uint64_t>)rel_s_;
    const int lane = threadIdx.x, nv = DP.nv;
    const size_t nn_s = (size_t)nv_s * nv_s, nn = (size_t)nv * nv;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r0 = tile * kWave + lane;
        const bool live = r0 < B;
        const size_t st = live ? r0 : B - 1;
        const T *aq = Aq ? Aq + tile * nn_s * kWave + lane : nullptr, *av = Av ? Av + tile * nn_s * kWave + lane : nullptr;
        const T *hs = Hs + tile * nn_s * kWave + lane;
        const T *ts = tau_s ? tau_s + st * (size_t)nv_s : nullptr;
        const T *cp = cpl + (tile * (size_t)n_cpl_rows) * kWave + lane;
        const size_t grp = st / IL, sub = st % IL;
        T *Dqs = Dq ? Dq + grp * nn * IL + sub : nullptr, *Dqds = Dqd ? Dqd + grp * nn * IL + sub : nullptr;
        T *Hout = H + grp * nn * IL + sub;
        auto packed = [](int r, int c) -> size_t { return c <= r ? (size_t)(r * r + c) : (size_t)(c * c + c + 1 + r); };
        auto sym = [](int r, int c) -> size_t { return r >= c ? (size_t)(r * (r + 1) / 2 + c) : (size_t)(c * (c + 1) / 2 + r); };
        // (gridDim.y > 1: the column clusters are dealt out over the workgroups of a tile -- batches with fewer tiles than wavefront slots)
        for (int cJ = blockIdx.y; cJ < n_clusters; cJ += gridDim.y) {
            const ClusterRec J = load_rec(clusters + cJ);
            const int nJ = J.kind == CK_FREE ? 6 : J.n, kJ = J.kind == CK_FREE ? 6 : J.k;
            const int strideJ = J.kind == CK_LOOP ? cpl_stride<KB>(J.n) : 0;
            const T *cJp = cp + (size_t)(J.kind == CK_LOOP ? crow[cJ] : 0) * kWave;
            if (J.kind == CK_FREE) {
                // The floating base's columns: G e_a is a unit vector, so a column of the projected matrices is a column of the spanning
                // ones contracted with G_I^T -- and every cluster is related to the base.  All FC columns at once: each entry of A_q / A_v / H_s
                // in the base's columns is read ONCE (column by column it was the same rows of G_I against one entry at a time, six passes).
#ifdef GRBDA_EXP_FC
                constexpr int FC = GRBDA_EXP_FC;
#else
                constexpr int FC = 3;  // (six at once: 197 registers in fp32, two wavefronts per SIMD instead of three, 7 % slower)
#endif
                const int sv0 = span_v[J.first_body];
                for (int a0 = 0; a0 < 6; a0 += FC) {
                    for (int cI = 0; cI < n_clusters; cI++) {
                        const ClusterRec I = load_rec(clusters + cI);
                        if (!((rel[I.v_index] >> J.v_index) & 1)) continue;
                        const int nI = I.kind == CK_FREE ? 6 : I.n, kI = I.kind == CK_FREE ? 6 : I.k;
                        const int strideI = I.kind == CK_LOOP ? cpl_stride<KB>(I.n) : 0;
                        const T *cIp = cp + (size_t)(I.kind == CK_LOOP ? crow[cI] : 0) * kWave;
                        T oq[KN][FC], ov[KN][FC], oh[KN][FC];
#pragma unroll
                        for (int b2 = 0; b2 < KN; b2++)
#pragma unroll
                            for (int a = 0; a < FC; a++) oq[b2][a] = ov[b2][a] = oh[b2][a] = 0;
                        for (int ri = 0; ri < kI; ri++) {
                            const int r = I.kind == CK_FREE ? span_v[I.first_body] + ri : span_v[I.first_body + ri];
                            const uint64_t rr = rel_s[r];
                            T x[FC], y[FC], h[FC];
#pragma unroll
                            for (int a = 0; a < FC; a++) {
                                const int sv = sv0 + a0 + a;
                                const bool on = (rr >> sv) & 1;
                                h[a] = on ? hs[sym(r, sv) * kWave] : T(0);
                                x[a] = (on && mode == 0) ? aq[packed(r, sv) * kWave] : T(0);
                                y[a] = (on && mode == 0) ? av[packed(r, sv) * kWave] : T(0);
                            }
                            if (I.kind == CK_FREE) {  // (the base against itself: G_I is the identity too)
                                if (live) {
#pragma unroll
                                    for (int a = 0; a < FC; a++) {
                                        const int vI = I.v_index + ri, vJ = J.v_index + a0 + a;
                                        if (mode == 0) {
                                            Dqs[packed(vI, vJ) * IL] = x[a];
                                            Dqds[packed(vI, vJ) * IL] = y[a];
                                        }
                                        if (vJ <= vI) Hout[sym(vI, vJ) * IL] = h[a];
                                    }
                                }
                                continue;
                            }
                            cptr<T> gS = consts + (I.kind == CK_STATIC ? load_rec(bodies + (I.first_body + ri)).cofs + kBodyConstFixed : 0);
#pragma unroll
                            for (int b2 = 0; b2 < KN; b2++) {
                                if (b2 >= nI) continue;
                                const T g = I.kind == CK_STATIC ? gS[b2] : cIp[(size_t)(ri * strideI + b2) * kWave];
#pragma unroll
                                for (int a = 0; a < FC; a++) {
                                    oq[b2][a] += g * x[a];
                                    ov[b2][a] += g * y[a];
                                    oh[b2][a] += g * h[a];
                                }
                            }
                        }
                        if (live && I.kind != CK_FREE) {
#pragma unroll
                            for (int b2 = 0; b2 < KN; b2++) {
                                if (b2 >= nI) continue;
#pragma unroll
                                for (int a = 0; a < FC; a++) {
                                    const int vI = I.v_index + b2, vJ = J.v_index + a0 + a;
                                    if (mode == 0) {
                                        Dqs[packed(vI, vJ) * IL] = oq[b2][a];
                                        Dqds[packed(vI, vJ) * IL] = ov[b2][a];
                                    }
                                    if (vJ <= vI) Hout[sym(vI, vJ) * IL] = oh[b2][a];
                                }
                            }
                        }
                    }
                }
                continue;
            }
            // (J is a static or an implicit cluster here: the base's columns went above.)  NC columns at a time: the entries of A_q / A_v / H_s of
            // a cluster pair are read once per NC columns of J
#ifdef GRBDA_EXP_NC
            constexpr int NC = GRBDA_EXP_NC;
#else
            constexpr int NC = sizeof(T) == 4 ? 2 : 1;  // (fp64 with two: 234 registers + scratch, one wavefront per SIMD)
#endif
            for (int a0 = 0; a0 < nJ; a0 += NC) {
                // column data over the spanning coordinates of cluster J: G e_a, G_a' yd, G_a' ydd + dg/dy_a, dg/dyd_a
                T gJ[KB][NC], ayJ[KB][NC], byJ[KB][NC], bvJ[KB][NC];
                int svJ[KB];
                for (int s = 0; s < KB; s++) {
#pragma unroll
                    for (int u = 0; u < NC; u++) gJ[s][u] = ayJ[s][u] = byJ[s][u] = bvJ[s][u] = 0;
                    svJ[s] = 0;
                    if (s >= kJ) continue;
                    svJ[s] = span_v[J.first_body + s];
#pragma unroll
                    for (int u = 0; u < NC; u++) {
                        const int a = a0 + u;
                        if (a >= nJ) continue;
                        if (J.kind == CK_STATIC) {
                            gJ[s][u] = consts[load_rec(bodies + (J.first_body + s)).cofs + kBodyConstFixed + a];
                        } else {
                            gJ[s][u] = cJp[(size_t)(s * strideJ + a) * kWave];
                            if (mode == 0) {
                                ayJ[s][u] = cJp[(size_t)(s * strideJ + J.n + a) * kWave];
                                byJ[s][u] = cJp[(size_t)(s * strideJ + 2 * J.n + a) * kWave];
                                bvJ[s][u] = cJp[(size_t)(s * strideJ + 3 * J.n + a) * kWave];
                            }
                        }
                    }
                }
                for (int cI = 0; cI < n_clusters; cI++) {
                    const ClusterRec I = load_rec(clusters + cI);
                    if (!((rel[I.v_index] >> J.v_index) & 1)) continue;  // clusters on different branches: structural zeros
                    const int nI = I.kind == CK_FREE ? 6 : I.n, kI = I.kind == CK_FREE ? 6 : I.k;
                    const int strideI = I.kind == CK_LOOP ? cpl_stride<KB>(I.n) : 0;
                    const T *cIp = cp + (size_t)(I.kind == CK_LOOP ? crow[cI] : 0) * kWave;
                    T oq[KN + 2][NC], ov[KN + 2][NC], oh[KN + 2][NC];
                    for (int b2 = 0; b2 < KN + 2; b2++)
#pragma unroll
                        for (int u = 0; u < NC; u++) oq[b2][u] = ov[b2][u] = oh[b2][u] = 0;
                    for (int ri = 0; ri < kI; ri++) {
                        const int r = I.kind == CK_FREE ? span_v[I.first_body] + ri : span_v[I.first_body + ri];
                        const uint64_t rr = rel_s[r];
                        T colq[NC], colv[NC], colh[NC];
#pragma unroll
                        for (int u = 0; u < NC; u++) colq[u] = colv[u] = colh[u] = 0;
                        for (int s = 0; s < kJ; s++) {
                            const int sv = svJ[s];
                            if (!((rr >> sv) & 1)) continue;
                            const T h = hs[sym(r, sv) * kWave];
                            T x = 0, y = 0;
                            if (mode == 0) {
                                x = aq[packed(r, sv) * kWave];
                                y = av[packed(r, sv) * kWave];
                            }
#pragma unroll
                            for (int u = 0; u < NC; u++) {
                                colh[u] += h * gJ[s][u];
                                if (mode == 0) {
                                    colq[u] += x * gJ[s][u] + y * ayJ[s][u] + h * byJ[s][u];
                                    colv[u] += y * gJ[s][u] + h * bvJ[s][u];
                                }
                            }
                        }
                        for (int b2 = 0; b2 < nI; b2++) {
                            T g;
                            if (I.kind == CK_FREE) g = ri == b2 ? T(1) : T(0);
                            else if (I.kind == CK_STATIC) g = consts[load_rec(bodies + (I.first_body + ri)).cofs + kBodyConstFixed + b2];
                            else g = cIp[(size_t)(ri * strideI + b2) * kWave];
#pragma unroll
                            for (int u = 0; u < NC; u++) {
                                oq[b2][u] += g * colq[u];
                                ov[b2][u] += g * colv[u];
                                oh[b2][u] += g * colh[u];
                            }
                        }
                        // (d G^T / d y_a) tau_s: own cluster only, dependent bodies carry the rows of G_a'
                        if (mode == 0 && cI == cJ && I.kind == CK_LOOP) {
                            const T tr = ts[r];
                            for (int b2 = 0; b2 < nI; b2++)
#pragma unroll
                                for (int u = 0; u < NC; u++)
                                    if (a0 + u < nJ) oq[b2][u] += cIp[(size_t)(ri * strideI + 4 * I.n + (a0 + u) * I.n + b2) * kWave] * tr;
                        }
                    }
                    if (live) {
                        for (int b2 = 0; b2 < nI; b2++) {
                            const int vI = I.v_index + b2;
#pragma unroll
                            for (int u = 0; u < NC; u++) {
                                if (a0 + u >= nJ) continue;
                                const int vJ = J.v_index + a0 + u;
                                if (mode == 0) {
                                    Dqs[packed(vI, vJ) * IL] = oq[b2][u];
                                    Dqds[packed(vI, vJ) * IL] = ov[b2][u];
                                }
                                if (vJ <= vI) Hout[sym(vI, vJ) * IL] = oh[b2][u];
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel 2w: H = G^T H_s G for plans with clusters beyond the structured limits (mode 1 of kernel 2, state-major packed rows).
// The general kernel's column work areas would be private arrays of 48 entries here -- scratch memory, and a dependent
// read-modify-write of scratch per multiply-add (measured: 16.5 ms per 65 536 states of the reference's 16-body parallel chain,
// 87 % of that model's forward dynamics).  Per column a of cluster J and row cluster I this kernel first forms (H_s G e_a) on the
// bodies of I with scalar accumulators and parks the k_I values in LDS ([body][lane]), then contracts them with the rows of G_I:
// every accumulator a register, every load a coalesced row of the tile-interleaved H_s or of the coupling slab, G of explicit
// clusters from the constant tables on the scalar unit.
// ---------------------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char manifold_smem[];
template <class T>
__global__ __launch_bounds__(kWave, 4) void manifold_project_wide_kernel(DevPlan<T> DP, int n_clusters, const int32_t *__restrict__ span_v_,
                                                                       const int32_t *__restrict__ crow_, const uint64_t *__restrict__ rel_,
                                                                       const uint64_t *__restrict__ rel_s_, const int32_t *__restrict__ relt_,
                                                                       const int32_t *__restrict__ relt_s_, int nv_s, int n_cpl_rows,
                                                                       const T *__restrict__ Hs, const T *__restrict__ cpl,
                                                                       T *__restrict__ H, size_t B)
{
    // relt / relt_s (plans with more than 64 velocities): nv x nv and nv_s x nv_s tables instead of the one-word masks
    cptr<int32_t> relt = (cptr<int32_t>)relt_, relt_s = (cptr<int32_t>)relt_s_;
    const bool tables = relt_ != nullptr;
    cptr<ClusterRec> clusters = (cptr<ClusterRec>)DP.clusters;
    cptr<int32_t> span_v = (cptr<int32_t>)span_v_, crow = (cptr<int32_t>)crow_;
    cptr<uint64_t> rel = (cptr<uint64_t>)rel_, rel_s = (cptr<uint64_t>)rel_s_;
    const int lane = threadIdx.x, nv = DP.nv;
    const size_t nn_s = (size_t)nv_s * nv_s, nn = (size_t)nv * nv;
    T *park = reinterpret_cast<T *>(manifold_smem) + lane;  // park[i * kWave]: (H_s G e_a) on body i of the row cluster
    const size_t n_tiles = (B + kWave - 1) / kWave;
    auto sym = [](int r, int c) -> size_t { return r >= c ? (size_t)(r * (r + 1) / 2 + c) : (size_t)(c * (c + 1) / 2 + r); };
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r0 = tile * kWave + lane;
        const bool live = r0 < B;
        const size_t st = live ? r0 : B - 1;
        const T *hs = Hs + tile * nn_s * kWave + lane;
        const T *cp = cpl + (tile * (size_t)n_cpl_rows) * kWave + lane;
        T *Hout = H + st * nn;
        // entry (i, a) of a cluster's G: the unit matrix of a free base, or a row of the coupling slab (explicit clusters too in wide plans)
        auto Gval = [&](const ClusterRec &C, int c0, int i, int a) -> T {
            if (C.kind == CK_FREE) return i == a ? T(1) : T(0);
            return cp[(size_t)(c0 + i * C.n + a) * kWave];
        };
        for (int cJ = 0; cJ < n_clusters; cJ++) {
            const ClusterRec J = load_rec(clusters + cJ);
            const int nJ = J.kind == CK_FREE ? 6 : J.n, kJ = J.kind == CK_FREE ? 6 : J.k;
            const int svJ = span_v[J.first_body], rowJ = J.kind == CK_FREE ? 0 : crow[cJ];  // (a cluster's bodies are consecutive spanning coordinates)
            for (int a = 0; a < nJ; a++) {
                const int vJ = J.v_index + a;
                for (int cI = cJ; cI < n_clusters; cI++) {  // (the lower triangle: rows at or after the column's cluster)
                    const ClusterRec I = load_rec(clusters + cI);
                    if (tables ? relt[I.v_index * nv + J.v_index] == 0 : !((rel[I.v_index] >> J.v_index) & 1)) continue;  // different branches: structural zeros
                    const int nI = I.kind == CK_FREE ? 6 : I.n, kI = I.kind == CK_FREE ? 6 : I.k;
                    const int svI = span_v[I.first_body], rowI = I.kind == CK_FREE ? 0 : crow[cI];
                    for (int ri = 0; ri < kI; ri++) {
                        const int r = svI + ri;
                        const uint64_t rr = tables ? 0 : rel_s[r];
                        T acc = 0;
#pragma unroll 8
                        for (int s = 0; s < kJ; s++) {
                            const int sv = svJ + s;
                            // (the load is unconditional -- a structural zero of H_s is allocated, never written, and discarded by the
                            // select -- so that the unrolled loop keeps eight pairs of loads in flight instead of branching per entry)
                            const T hv = hs[sym(r, sv) * kWave];
                            const bool on = tables ? relt_s[r * nv_s + sv] != 0 : ((rr >> sv) & 1) != 0;
                            const T h = on ? hv : T(0);
                            acc += h * Gval(J, rowJ, s, a);
                        }
                        park[ri * kWave] = acc;
                    }
                    for (int b2 = 0; b2 < nI; b2++) {
                        const int vI = I.v_index + b2;
                        if (vJ > vI) continue;
                        T acc = 0;
#pragma unroll 8
                        for (int ri = 0; ri < kI; ri++) acc += Gval(I, rowI, ri, b2) * park[ri * kWave];
                        if (live) Hout[sym(vI, vJ)] = acc;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel 3: between the independent coordinates and the spanning tree, one state per lane (capi.cpp, projection_run).
//   mode 0 (inverse dynamics):  out = G^T x_s                                   x_s = tau_s(q_s, G yd, G ydd + g)
//   mode 1 (forward dynamics):  out = Hinv (tau - G^T x_s)                      x_s = C_s + H_s g,  Hinv = (G^T H_s G)^-1 [B][nv][nv]
//   mode 2:                      out = tau - G^T x_s
// ---------------------------------------------------------------------------------------------------------------
template <class T, int KB>
__global__ __launch_bounds__(kWave, 1) void manifold_apply_kernel(DevPlan<T> DP, int n_clusters, const int32_t *__restrict__ span_v_,
                                                                const int32_t *__restrict__ crow_, int nv_s, int n_cpl_rows, int mode,
                                                                const T *__restrict__ x_s, const T *__restrict__ tau,
                                                                const T *__restrict__ Hinv, const T *__restrict__ cpl, T *__restrict__ out,
                                                                size_t B)
{
    cptr<ClusterRec> clusters = (cptr<ClusterRec>)DP.clusters;
    cptr<BodyRec> bodies = (cptr<BodyRec>)DP.bodies;
    cptr<T> consts = (cptr<T>)DP.consts;
    cptr<int32_t> span_v = (cptr<int32_t>)span_v_, crow = (cptr<int32_t>)crow_;
    const int lane = threadIdx.x, nv = DP.nv;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r0 = tile * kWave + lane;
        const bool live = r0 < B;
        const size_t st = live ? r0 : B - 1;
        const T *xs = x_s + st * (size_t)nv_s;
        const T *cp = cpl + (tile * (size_t)n_cpl_rows) * kWave + lane;
        T rhs[2 * kWave];  // (up to 128 velocities on the wide route)
        for (int c = 0; c < n_clusters; c++) {
            const ClusterRec cr = load_rec(clusters + c);
            const int n = cr.kind == CK_FREE ? 6 : cr.n, k = cr.kind == CK_FREE ? 6 : cr.k;
            const int stride = cr.kind == CK_LOOP ? cpl_stride<KB>(cr.n) : 0;
            const T *cc = cp + (size_t)(cr.kind == CK_LOOP ? crow[c] : 0) * kWave;
            for (int a = 0; a < n; a++) {
                T s = 0;
                for (int i = 0; i < k; i++) {
                    T g;
                    int sv;
                    if (cr.kind == CK_FREE) { g = i == a ? T(1) : T(0); sv = span_v[cr.first_body] + i; }
                    else if (cr.kind == CK_STATIC) {
                        if constexpr (KB > kMaxClusterBodies) g = cp[(size_t)(crow[c] + i * cr.n + a) * kWave];  // (wide plans: G rows in the slab)
                        else g = consts[load_rec(bodies + (cr.first_body + i)).cofs + kBodyConstFixed + a];
                        sv = span_v[cr.first_body + i];
                    }
                    else { g = cc[(size_t)(i * stride + a) * kWave]; sv = span_v[cr.first_body + i]; }
                    s += g * xs[sv];
                }
                rhs[cr.v_index + a] = mode == 0 ? s : tau[st * (size_t)nv + cr.v_index + a] - s;
            }
        }
        if (!live) continue;
        T *o = out + st * (size_t)nv;
        if (mode == 0 || mode == 2) {  // (mode 2: tau - G^T x_s, the right-hand side of the wide route's own solve)
            for (int i = 0; i < nv; i++) o[i] = rhs[i];
        } else {
            const T *Hi = Hinv + st * (size_t)nv * nv;
            for (int i = 0; i < nv; i++) {
                T s = 0;
                for (int j = 0; j < nv; j++) s += Hi[(size_t)i * nv + j] * rhs[j];
                o[i] = s;
            }
        }
    }
}
// `big`: the plan has clusters beyond kMaxClusterBodies / kMaxClusterDof (HostPlan::big_clusters): the wide variants, G-only slab
template <class T>
hipError_t launch_manifold_apply(const DevPlan<T> &P, int n_clusters, const int32_t *span_v, const int32_t *crow, int nv_s, int n_cpl_rows, int mode,
                                 const T *x_s, const T *tau, const T *Hinv, const T *cpl, T *out, size_t B, int grid, hipStream_t stream, bool big)
{
    if (big)
        hipLaunchKernelGGL((manifold_apply_kernel<T, kBigClusterBodies>), dim3(grid), dim3(kWave), 0, stream, P, n_clusters, span_v, crow, nv_s,
                           n_cpl_rows, mode, x_s, tau, Hinv, cpl, out, B);
    else
        hipLaunchKernelGGL((manifold_apply_kernel<T, kMaxClusterBodies>), dim3(grid), dim3(kWave), 0, stream, P, n_clusters, span_v, crow, nv_s,
                           n_cpl_rows, mode, x_s, tau, Hinv, cpl, out, B);
    return hipGetLastError();
}
template hipError_t launch_manifold_apply<float>(const DevPlan<float> &, int, const int32_t *, const int32_t *, int, int, int, const float *,
                                                 const float *, const float *, const float *, float *, size_t, int, hipStream_t, bool);
template hipError_t launch_manifold_apply<double>(const DevPlan<double> &, int, const int32_t *, const int32_t *, int, int, int, const double *,
                                                  const double *, const double *, const double *, double *, size_t, int, hipStream_t, bool);

template <class T>
hipError_t launch_manifold_constraint(const DevPlan<T> &P, int n_clusters, const int32_t *span_q, const int32_t *span_v, const int32_t *crow,
                                      int nq_s, int nv_s, int n_cpl_rows, int want_d, const T *q, const T *qd, const T *ydd, T *q_s, T *qd_s,
                                      T *qdd_s, T *cpl, size_t B, int grid, hipStream_t stream, int shape, bool trig)
{
    // the sine / cosine cache of trig-polynomial constraints (kTrigLdsSlots rows per wavefront); the grid is cut to what the LDS holds
    const size_t lds = trig ? static_cast<size_t>(kTrigLdsSlots) * kWave * sizeof(T) : 0;
    int n_cu = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    }
    auto fit = [&](int g) {
        const size_t cap = static_cast<size_t>(n_cu) * lds_workgroups_per_cu(lds);
        return static_cast<size_t>(g) > cap ? static_cast<int>(cap) : g;
    };
    // shape 0: clusters of up to kMaxClusterBodies bodies / kMaxClusterDof independent coordinates; 1: beyond (runtime loops, no derivative
    // parts); 2: every implicit cluster has at most 4 bodies and 2 independent coordinates (the Tello differentials): the same code with
    // half the unrolled work areas
    const bool big = shape == 1;
    if (shape == 2) {
        // (fp32: 216 registers, two wavefronts per SIMD -- the caller's grid is for one)
        const size_t n_tiles = (B + kWave - 1) / kWave;
        if (sizeof(T) == 4 && static_cast<size_t>(grid) * 2 <= n_tiles) grid *= 2;
        grid = fit(grid);
        hipLaunchKernelGGL((manifold_constraint_kernel<T, 4, 2>), dim3(grid), dim3(kWave), lds, stream, P, n_clusters, span_q, span_v, crow, nq_s, nv_s,
                           n_cpl_rows, want_d, q, qd, ydd, q_s, qd_s, qdd_s, cpl, B);
        return hipGetLastError();
    }
    if (big) {
        if (want_d) return hipErrorInvalidValue;  // (the wide variant carries no derivative parts)
        hipLaunchKernelGGL((manifold_constraint_kernel<T, kBigClusterBodies, kBigClusterDof>), dim3(fit(grid)), dim3(kWave), lds, stream, P, n_clusters,
                           span_q, span_v, crow, nq_s, nv_s, n_cpl_rows, want_d, q, qd, ydd, q_s, qd_s, qdd_s, cpl, B);
    } else {
        hipLaunchKernelGGL((manifold_constraint_kernel<T, kMaxClusterBodies, kMaxClusterDof>), dim3(fit(grid)), dim3(kWave), lds, stream, P, n_clusters,
                           span_q, span_v, crow, nq_s, nv_s, n_cpl_rows, want_d, q, qd, ydd, q_s, qd_s, qdd_s, cpl, B);
    }
    return hipGetLastError();
}
template <class T>
hipError_t launch_manifold_project_wide(const DevPlan<T> &P, int n_clusters, const int32_t *span_v, const int32_t *crow, const uint64_t *rel,
                                        const uint64_t *rel_s, const int32_t *relt, const int32_t *relt_s, int nv_s, int n_cpl_rows, const T *Hs,
                                        const T *cpl, T *H, size_t B, int grid, hipStream_t stream)
{
    // (bound by the latency of its coalesced loads -- the tile's H_s and coupling rows come from HBM: as many wavefronts as the LDS holds,
    // up to four per SIMD; the caller's grid is for one per SIMD)
    const size_t lds = kBigClusterBodies * kWave * sizeof(T);
    size_t per_cu = lds_workgroups_per_cu(lds);
    if (per_cu > 16) per_cu = 16;
    // (the caller's grid is four wavefronts per CU, or the number of tiles when that is smaller: round UP to whole CUs -- grid / 4 was zero
    // for batches of fewer than four tiles, and every tile of such a batch ran on one wavefront)
    size_t g = (static_cast<size_t>(grid) + 3) / 4 * per_cu;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    if (g > n_tiles) g = n_tiles;
    if (g < 1) g = 1;
    hipLaunchKernelGGL((manifold_project_wide_kernel<T>), dim3(static_cast<unsigned>(g)), dim3(kWave), lds, stream, P, n_clusters, span_v, crow, rel,
                       rel_s, relt, relt_s, nv_s, n_cpl_rows, Hs, cpl, H, B);
    return hipGetLastError();
}
template hipError_t launch_manifold_project_wide<float>(const DevPlan<float> &, int, const int32_t *, const int32_t *, const uint64_t *,
                                                        const uint64_t *, const int32_t *, const int32_t *, int, int, const float *, const float *,
                                                        float *, size_t, int, hipStream_t);
template hipError_t launch_manifold_project_wide<double>(const DevPlan<double> &, int, const int32_t *, const int32_t *, const uint64_t *,
                                                         const uint64_t *, const int32_t *, const int32_t *, int, int, const double *,
                                                         const double *, double *, size_t, int, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// Kernel 5: Newton projection of the dependent spanning positions of URDF+ position-loop clusters onto phi(q) = 0, in place, for plans
// with clusters beyond the structured limits (the structured plans use kernels.hip's project_kernel): GenericJoint.cpp:289-385 --
// q_dep <- q_dep - K_d^-1 phi until |phi|_2 < tol or max_iter steps; ok[b] = every cluster converged.  One state per lane.
// ---------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(kWave, 1) void manifold_newton_kernel(DevPlan<T> DP, int n_clusters, T *__restrict__ q, int32_t *__restrict__ ok,
                                                                 size_t B, int max_iter, T tol)
{
    constexpr int KB = kBigClusterBodies;
    cptr<ClusterRec> clusters = (cptr<ClusterRec>)DP.clusters;
    cptr<BodyRec> bodies = (cptr<BodyRec>)DP.bodies;
    cptr<T> consts = (cptr<T>)DP.consts;
    cptr<int32_t> cints = (cptr<int32_t>)DP.cints;
    const int lane = threadIdx.x, nq = DP.nq;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r0 = tile * kWave + lane;
        const bool live = r0 < B;
        T *qs = q + (live ? r0 : B - 1) * (size_t)nq;
        bool good = true;
        for (int c = 0; c < n_clusters; c++) {
            const ClusterRec cr = load_rec(clusters + c);
            if (cr.kind != CK_LOOP) continue;
            if (cr.cons_type != 0) { good = false; continue; }  // (trig-polynomial constraints: structured plans only)
            cptr<int32_t> ip = cints + cr.iofs;
            const int hdr0 = ip[0], n_ind = ip[1], rows = cr.rows, k = cr.k;
            cptr<int32_t> dep = ip + 3 + n_ind;
            cptr<int32_t> payload = ip + 3 + n_ind + rows;
            T qv[KB], sn[KB], cs[KB], zero[KB];
            for (int j = 0; j < KB; j++) {
                qv[j] = j < k ? qs[cr.q_index + j] : T(0);
                zero[j] = 0;
            }
            T nrm = T(1e30);
            for (int it = 0; it <= max_iter; it++) {
                T K[kMRof<KB>][KB], kap[kMRof<KB>], phi[kMRof<KB>];
                for (int r = 0; r < kMRof<KB>; r++) {
                    kap[r] = phi[r] = 0;
                    for (int j = 0; j < KB; j++) K[r][j] = 0;
                }
                for (int j = 0; j < k; j++) sincos_precise(qv[j], &sn[j], &cs[j]);
                loop_position_eval<T, T, KB>(consts, bodies, cr, payload, hdr0, sn, cs, zero, true, K, kap, phi);
                nrm = 0;
                for (int r = 0; r < rows; r++) nrm += phi[r] * phi[r];
                nrm = (T)__builtin_sqrt((double)nrm);
                // (all lanes iterate together: a lane that has converged keeps its coordinates)
                const bool done = nrm < tol;
                if (__builtin_amdgcn_ballot_w64(!done) == 0 || it == max_iter) break;
                T Kd[kMRof<KB>][kMRof<KB>], Kdi[kMRof<KB>][kMRof<KB>];
                for (int r = 0; r < kMRof<KB>; r++)
                    for (int j = 0; j < kMRof<KB>; j++) Kd[r][j] = (r < rows && j < rows) ? K[r][dep[j]] : T(r == j);
                inv_rows(rows, Kd, Kdi);
                for (int r = 0; r < rows; r++) {
                    T s = 0;
                    for (int j = 0; j < rows; j++) s += Kdi[r][j] * phi[j];
                    if (!done) qv[dep[r]] -= s;
                }
            }
            good = good && (nrm < tol);
            if (live)
                for (int r = 0; r < rows; r++) qs[cr.q_index + dep[r]] = qv[dep[r]];
        }
        if (ok && live) ok[r0] = good ? 1 : 0;
    }
}
template <class T>
hipError_t launch_manifold_newton(const DevPlan<T> &P, int n_clusters, T *q, int32_t *ok, size_t B, int max_iter, T tol, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL((manifold_newton_kernel<T>), dim3(grid), dim3(kWave), 0, stream, P, n_clusters, q, ok, B, max_iter, tol);
    return hipGetLastError();
}
template hipError_t launch_manifold_newton<float>(const DevPlan<float> &, int, float *, int32_t *, size_t, int, float, int, hipStream_t);
template hipError_t launch_manifold_newton<double>(const DevPlan<double> &, int, double *, int32_t *, size_t, int, double, int, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// Kernel 6: state input in the reference's conventions for plans with clusters beyond the structured limits -- the rules of kernels.hip's
// state_kernel (ClusterJoints::Base::toSpanningTreeState, ClusterJoint.cpp:22-71; LoopConstraint.cpp:15-26): per cluster the caller's
// positions / velocities are independent or SPANNING coordinates (flags F); explicit clusters: y = G+ q_span, yd = G+ qd_span, spanning
// velocities valid iff |K qd_span| < tol; implicit (position-loop) clusters: spanning positions, valid iff |phi(q)| < tol, spanning
// velocities valid iff |K qd_span| < tol and yd = their independent entries.  status[b] = 0 or (1 position / 2 velocity) + 256 cluster of
// the first failure; cond[b] = (max |Kd^-1 Ki|, max |Kd|_F |Kd^-1|_F) over the implicit clusters.  One state per lane.
// ---------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(kWave, 1) void manifold_state_kernel(DevPlan<T> DP, int n_clusters, StateFlags F, const T *__restrict__ q_in,
                                                                const T *__restrict__ qd_in, int in_nq, int in_nv, T *__restrict__ q_out,
                                                                T *__restrict__ qd_out, int32_t *__restrict__ status, T *__restrict__ cond,
                                                                size_t B, T tol)
{
    constexpr int KB = kBigClusterBodies;
    cptr<ClusterRec> clusters = (cptr<ClusterRec>)DP.clusters;
    cptr<BodyRec> bodies = (cptr<BodyRec>)DP.bodies;
    cptr<T> consts = (cptr<T>)DP.consts;
    cptr<int32_t> cints = (cptr<int32_t>)DP.cints;
    const int lane = threadIdx.x, nq = DP.nq, nv = DP.nv;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r0 = tile * kWave + lane;
        const bool write = r0 < B;
        const size_t row = write ? r0 : B - 1;
        const T *qi = q_in + row * (size_t)in_nq;
        const T *vi = qd_in ? qd_in + row * (size_t)in_nv : nullptr;
        T *qo = q_out ? q_out + row * (size_t)nq : nullptr;
        T *vo = qd_out ? qd_out + row * (size_t)nv : nullptr;
        int st = 0;
        T gm = 0, kc = 0;
        for (int ci = 0; ci < n_clusters; ci++) {
            const ClusterRec c = load_rec(clusters + ci);
            const bool ps = (F.pos[ci >> 6] >> (ci & 63)) & 1, vs = (F.vel[ci >> 6] >> (ci & 63)) & 1;
            T *qoc = qo ? qo + c.q_index : nullptr, *voc = vo ? vo + c.v_index : nullptr;
            if (c.kind == CK_FREE) {
                const int npos = DP.ori_repr == 0 ? 7 : 6;
                if (qoc && write)
                    for (int j = 0; j < npos; j++) qoc[j] = qi[j];
                if (vi && voc && write)
                    for (int j = 0; j < 6; j++) voc[j] = vi[j];
                qi += npos;
                if (vi) vi += 6;
            } else if (c.kind == CK_LOOP) {
                const int k = c.k, n = c.n, rows = c.rows;
                cptr<int32_t> ip = cints + c.iofs;
                const int hdr0 = ip[0], n_ind = ip[1];
                cptr<int32_t> ind = ip + 2, dep = ip + 3 + n_ind, payload = ip + 3 + n_ind + rows;
                T sn[KB], cs[KB], zero[KB], K[kMRof<KB>][KB], kap[kMRof<KB>], phi[kMRof<KB>];
                for (int j = 0; j < KB; j++) zero[j] = 0;
                for (int r = 0; r < kMRof<KB>; r++) {
                    kap[r] = phi[r] = 0;
                    for (int j = 0; j < KB; j++) K[r][j] = 0;
                }
                for (int i = 0; i < k; i++) {
                    const T v = qi[i];
                    sincos_precise(v, &sn[i], &cs[i]);
                    if (qoc && write) qoc[i] = v;
                }
                if (c.cons_type == 0) loop_position_eval<T, T, KB>(consts, bodies, c, payload, hdr0, sn, cs, zero, true, K, kap, phi);
                else if (st == 0) st = 1 + 256 * ci;  // (trig-polynomial constraints: structured plans only)
                T n2 = 0;
                for (int r = 0; r < rows; r++) n2 += phi[r] * phi[r];
                if (!((T)__builtin_sqrt((double)n2) < tol) && st == 0) st = 1 + 256 * ci;
                if (vi) {
                    if (vs) {
                        T s2 = 0;
                        for (int r = 0; r < rows; r++) {
                            T sacc = 0;
                            for (int i = 0; i < k; i++) sacc += K[r][i] * vi[i];
                            s2 += sacc * sacc;
                        }
                        if (!((T)__builtin_sqrt((double)s2) < tol) && st == 0) st = 2 + 256 * ci;
                    }
                    for (int a = 0; a < n; a++) {
                        const T v = vi[vs ? ind[a] : a];
                        if (voc && write) voc[a] = v;
                    }
                }
                T Kd[kMRof<KB>][kMRof<KB>], Kdi[kMRof<KB>][kMRof<KB>];
                for (int r = 0; r < kMRof<KB>; r++)
                    for (int j = 0; j < kMRof<KB>; j++) Kd[r][j] = (r < rows && j < rows) ? K[r][dep[j]] : T(r == j);
                inv_rows(rows, Kd, Kdi);
                T f1 = 0, f2 = 0;
                for (int r = 0; r < rows; r++)
                    for (int j = 0; j < rows; j++) {
                        f1 += Kd[r][j] * Kd[r][j];
                        f2 += Kdi[r][j] * Kdi[r][j];
                    }
                const T cn = (T)__builtin_sqrt((double)(f1 * f2));
                kc = (cn > kc || cn != cn) ? cn : kc;
                for (int r = 0; r < rows; r++)
                    for (int a = 0; a < n; a++) {
                        T x = 0;
                        for (int j = 0; j < rows; j++) x += Kdi[r][j] * K[j][ind[a]];
                        x = x < 0 ? -x : x;
                        gm = (x > gm || x != x) ? x : gm;  // (NaN of a singular Kd sticks)
                    }
                qi += k;
                if (vi) vi += vs ? k : n;
            } else {  // explicit: consts[dofs] = K (rows x k), then G+ (n x k)  (plan.cpp)
                cptr<T> Kc = consts + c.dofs;
                cptr<T> Gp = Kc + c.rows * c.k;
                for (int a = 0; a < c.n; a++) {
                    T y = 0;
                    if (ps) {
                        for (int i = 0; i < c.k; i++) y += Gp[a * c.k + i] * qi[i];
                    } else {
                        y = qi[a];
                    }
                    if (qoc && write) qoc[a] = y;
                }
                qi += ps ? c.k : c.n;
                if (vi) {
                    if (vs) {
                        T s2 = 0;
                        for (int rr = 0; rr < c.rows; rr++) {
                            T sacc = 0;
                            for (int i = 0; i < c.k; i++) sacc += Kc[rr * c.k + i] * vi[i];
                            s2 += sacc * sacc;
                        }
                        if (!((T)__builtin_sqrt((double)s2) < tol) && st == 0) st = 2 + 256 * ci;
                    }
                    for (int a = 0; a < c.n; a++) {
                        T y = 0;
                        if (vs) {
                            for (int i = 0; i < c.k; i++) y += Gp[a * c.k + i] * vi[i];
                        } else {
                            y = vi[a];
                        }
                        if (voc && write) voc[a] = y;
                    }
                    vi += vs ? c.k : c.n;
                }
            }
        }
        if (status && write) status[r0] = st;
        if (cond && write) {
            cond[2 * r0] = gm;
            cond[2 * r0 + 1] = kc;
        }
    }
}
template <class T>
hipError_t launch_manifold_state(const DevPlan<T> &P, int n_clusters, const StateFlags &F, const T *q_in, const T *qd_in, int in_nq, int in_nv,
                                 T *q_out, T *qd_out, int32_t *status, T *cond, size_t B, T tol, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL((manifold_state_kernel<T>), dim3(grid), dim3(kWave), 0, stream, P, n_clusters, F, q_in, qd_in, in_nq, in_nv, q_out, qd_out,
                       status, cond, B, tol);
    return hipGetLastError();
}
template hipError_t launch_manifold_state<float>(const DevPlan<float> &, int, const StateFlags &, const float *, const float *, int, int, float *,
                                                 float *, int32_t *, float *, size_t, float, int, hipStream_t);
template hipError_t launch_manifold_state<double>(const DevPlan<double> &, int, const StateFlags &, const double *, const double *, int, int,
                                                  double *, double *, int32_t *, double *, size_t, double, int, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// Kernel 4: y = H^-1 b for up to 128 velocities, one state per WORKGROUP of four wavefronts (the wide route beyond the 64 coordinates of
// deriv_kernels.hip's row-per-lane solves).  H: packed lower rows [B][nv^2] as kernel 2w leaves them, structural zeros (never
// written) masked by the nv x nv table; the factor L (H = L L^T) replaces it in LDS, then two substitutions on the one right-hand side.
// Slow and plain: a column at a time, barriers between the steps.
// ---------------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(4 * kWave) void spd_wide_solve_kernel(const T *__restrict__ H, const int32_t *__restrict__ relt, const T *__restrict__ rhs,
                                                                 T *__restrict__ out, int nv, size_t B, unsigned long long *bad_count)
{
    T *L = reinterpret_cast<T *>(manifold_smem);          // packed lower rows: L[i (i + 1) / 2 + j]
    T *x = L + (size_t)nv * (nv + 1) / 2;                 // right-hand side, then the solution
    const int tid = threadIdx.x, nt = blockDim.x;
    const size_t nn = (size_t)nv * nv;
    for (size_t s = blockIdx.x; s < B; s += gridDim.x) {
        __syncthreads();
        for (int e = tid; e < nv * (nv + 1) / 2; e += nt) {
            // entry e = (i, j): i = the largest row with i (i + 1) / 2 <= e
            int i = (int)((__builtin_sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
            while ((i + 1) * (i + 2) / 2 <= e) i++;
            while (i * (i + 1) / 2 > e) i--;
            const int j = e - i * (i + 1) / 2;
            L[e] = relt[i * nv + j] ? H[s * nn + e] : T(0);
        }
        for (int i = tid; i < nv; i += nt) x[i] = rhs[s * (size_t)nv + i];
        __syncthreads();
        bool bad = false;
        for (int k = 0; k < nv; k++) {
            const int kk = k * (k + 1) / 2 + k;
            const T d = L[kk];
            bad = bad || !(d > T(0));
            const T r = T(1) / __builtin_sqrt((double)d);
            __syncthreads();  // every thread has read the pivot
            if (tid == 0) L[kk] = (T)__builtin_sqrt((double)d);
            for (int i = k + 1 + tid; i < nv; i += nt) L[i * (i + 1) / 2 + k] *= r;
            __syncthreads();
            for (int i = k + 1 + (tid >> 4); i < nv; i += nt >> 4) {
                const T lik = L[i * (i + 1) / 2 + k];
                for (int j = k + 1 + (tid & 15); j <= i; j += 16) L[i * (i + 1) / 2 + j] -= lik * L[j * (j + 1) / 2 + k];
            }
            __syncthreads();
        }
        if (bad && tid == 0 && bad_count) atomicAdd(bad_count, 1ull);
        for (int k = 0; k < nv; k++) {  // L y = b
            if (tid == 0) x[k] /= L[k * (k + 1) / 2 + k];
            __syncthreads();
            const T yk = x[k];
            for (int i = k + 1 + tid; i < nv; i += nt) x[i] -= L[i * (i + 1) / 2 + k] * yk;
            __syncthreads();
        }
        for (int k = nv - 1; k >= 0; k--) {  // L^T x = y
            if (tid == 0) x[k] /= L[k * (k + 1) / 2 + k];
            __syncthreads();
            const T xk = x[k];
            for (int i = tid; i < k; i += nt) x[i] -= L[k * (k + 1) / 2 + i] * xk;
            __syncthreads();
        }
        for (int i = tid; i < nv; i += nt) out[s * (size_t)nv + i] = x[i];
    }
}
template <class T>
hipError_t launch_spd_wide_solve(const T *H, const int32_t *relt, const T *rhs, T *out, int nv, size_t B, int n_cu, hipStream_t stream,
                                 unsigned long long *bad_count)
{
    const size_t lds = ((size_t)nv * (nv + 1) / 2 + nv) * sizeof(T);
    size_t per_cu = lds_workgroups_per_cu(lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) return hipErrorInvalidValue;
    size_t g = static_cast<size_t>(n_cu) * per_cu;
    if (g > B) g = B;
    const void *fn = reinterpret_cast<const void *>(&spd_wide_solve_kernel<T>);
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((spd_wide_solve_kernel<T>), dim3(static_cast<unsigned>(g)), dim3(4 * kWave), lds, stream, H, relt, rhs, out, nv, B,
                       bad_count);
    return hipGetLastError();
}
template hipError_t launch_spd_wide_solve<float>(const float *, const int32_t *, const float *, float *, int, size_t, int, hipStream_t,
                                                 unsigned long long *);
template hipError_t launch_spd_wide_solve<double>(const double *, const int32_t *, const double *, double *, int, size_t, int, hipStream_t,
                                                  unsigned long long *);

template <class T>
hipError_t launch_manifold_project(const DevPlan<T> &P, int n_clusters, const int32_t *span_v, const int32_t *crow, const uint64_t *rel,
                                   const uint64_t *rel_s, int nv_s, int n_cpl_rows, int mode, const T *Aq, const T *Av, const T *Hs,
                                   const T *tau_s, const T *cpl, T *Dq, T *Dqd, T *H, size_t B, int grid, hipStream_t stream, int interleave,
                                   bool big)
{
    if (big) {
        if (mode != 1 || interleave != 1) return hipErrorInvalidValue;  // (H only, state-major)
        return launch_manifold_project_wide<T>(P, n_clusters, span_v, crow, rel, rel_s, nullptr, nullptr, nv_s, n_cpl_rows, Hs, cpl, H, B, grid, stream);
    }
    // (101 registers in fp32, 189 in fp64, and every load a coalesced row that comes from HBM: four / two wavefronts per SIMD hide what one
    // cannot -- the caller's grid is for one per SIMD)
    size_t g = static_cast<size_t>(grid) * (sizeof(T) == 4 ? 4 : 2);
    const size_t n_tiles = (B + kWave - 1) / kWave;
    // fewer than two tiles per wavefront slot: the column clusters of a tile go to workgroups of their own (TelloWithArms, 262 144 states:
    // 4 096 tiles on 4 096 slots, every wavefront a chain of ~60 k dependent loads)
    const unsigned gy = (n_tiles < 2 * g && n_clusters > 1) ? static_cast<unsigned>(n_clusters) : 1u;
    if (g > n_tiles) g = n_tiles;
    if (g < 1) g = 1;
    const dim3 gr(static_cast<unsigned>(g), gy);
    if (interleave == kDerivGroup) {
        hipLaunchKernelGGL((manifold_project_kernel<T, kDerivGroup, kMaxClusterBodies, kMaxClusterDof>), gr, dim3(kWave), 0, stream, P,
                           n_clusters, span_v, crow, rel, rel_s, nv_s, n_cpl_rows, mode, Aq, Av, Hs, tau_s, cpl, Dq, Dqd, H, B);
    } else {
        hipLaunchKernelGGL((manifold_project_kernel<T, 1, kMaxClusterBodies, kMaxClusterDof>), gr, dim3(kWave), 0, stream, P, n_clusters,
                           span_v, crow, rel, rel_s, nv_s, n_cpl_rows, mode, Aq, Av, Hs, tau_s, cpl, Dq, Dqd, H, B);
    }
    return hipGetLastError();
}
template hipError_t launch_manifold_constraint<float>(const DevPlan<float> &, int, const int32_t *, const int32_t *, const int32_t *, int, int, int,
                                                      int, const float *, const float *, const float *, float *, float *, float *, float *,
                                                      size_t, int, hipStream_t, int, bool);
template hipError_t launch_manifold_constraint<double>(const DevPlan<double> &, int, const int32_t *, const int32_t *, const int32_t *, int, int,
                                                       int, int, const double *, const double *, const double *, double *, double *, double *,
                                                       double *, size_t, int, hipStream_t, int, bool);
template hipError_t launch_manifold_project<float>(const DevPlan<float> &, int, const int32_t *, const int32_t *, const uint64_t *,
                                                   const uint64_t *, int, int, int, const float *, const float *, const float *, const float *,
                                                   const float *, float *, float *, float *, size_t, int, hipStream_t, int, bool);
template hipError_t launch_manifold_project<double>(const DevPlan<double> &, int, const int32_t *, const int32_t *, const uint64_t *,
                                                    const uint64_t *, int, int, int, const double *, const double *, const double *,
                                                    const double *, const double *, double *, double *, double *, size_t, int, hipStream_t, int,
                                                    bool);

}  // namespace grbda_hip
