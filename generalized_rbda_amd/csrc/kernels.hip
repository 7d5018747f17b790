// kernels.hip -- batched cluster-ABA and cluster-RNEA for gfx950 (MI355X, CDNA4).
//
// Execution model
//   * one robot STATE per lane, 64 states per wavefront, one wavefront per workgroup;
//   * every lane evaluates the SAME model, so the model "program" (plan.h: steps, body and
//     cluster records, Xtree / inertia / G constants) is wave-uniform: it is read through the
//     scalar unit (s_load) and feeds the VALU as SGPR operands, while the 64 lanes run the
//     per-state spatial arithmetic with no divergence and no cross-lane traffic;
//   * per-state intermediates are "slots" laid out [slot][lane]: LDS for the first
//     n_lds_slots (conflict-free: lane == bank), a per-wave global slab for the rest
//     (one fully coalesced 256/512-byte row per access);
//   * a wave walks the batch with a grid stride, so the scratch slab stays hot in L2.
//
// Algorithm (structured restatement of the reference's dense cluster formulation)
//   The reference builds dense 6k x 6k cluster quantities (ClusterTreeDynamics.cpp:157-191).
//   Here the same recursion is carried per BODY of the spanning tree, in "relative in-cluster
//   coordinates" (S = X_intra * S_span * G, GenericJoint.cpp:426): with T = X_intra,
//     D  = S^T IA S      = G^T (Shat^T (T^T IA T) Shat) G = G^T Hc G
//     F  = Xup^T IA S    = sum_j fc_j G_j.          (6 x n, parent-body frame)
//     u' = tau - S^T(pA + IA c) = tau - G^T b
//   where Hc (k x k) is the in-cluster joint-space inertia of the composite articulated
//   inertias, fc_j the force at the parent body per unit spanning acceleration of joint j and
//   b_j the in-cluster bias torque.  The projected inertia / bias handed to the parent body are
//     IA_p += sum_roots X^T IAc X - F D^-1 F^T,   pA_p += sum_roots X^T t + F D^-1 u'
//   which equals Xup^T (IA - U D^-1 U^T) Xup and Xup^T (pA + Ia c + U D^-1 u)
//   (ClusterTreeDynamics.cpp:120-127,181-187) whenever all in-cluster roots hang off one parent
//   body (checked by the plan compiler).
#include <hip/hip_runtime.h>

#include "devplan.h"

namespace grbda_hip {

// optional in-kernel cycle accounting (build with -DGRBDA_PROFILE, see tools/prof_run.py; never in
// the shipped library): s_memtime deltas per phase, summed over waves into grbda_prof[]
// wavefronts per SIMD the f32 ABA kernel is register-allocated for
#ifndef GRBDA_ABA32_WAVES
#define GRBDA_ABA32_WAVES 2
#endif


#ifdef GRBDA_PROFILE
__device__ unsigned long long grbda_prof[32];
#define PROF_T0() unsigned long long prof_t = __builtin_amdgcn_s_memtime()
#define PROF_ARGS , unsigned long long (&prof_acc)[24], unsigned long long &prof_t
#define PROF_PASS , prof_acc, prof_t
#define PROF_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PROF_SYNCV() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define PROF_ADD(i)                                                    \
    do {                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();  \
        prof_acc[i] += now_ - prof_t;                                  \
        prof_t = now_;                                                 \
    } while (0)
#else
#define PROF_T0()
#define PROF_ADD(i)
#define PROF_ARGS
#define PROF_PASS
#define PROF_SYNC()
#define PROF_SYNCV()
#endif

#include "devmath.h"

// body record of the fast kernels: the plan compiler re-expresses every revolute body in a frame whose
// z axis is the joint axis (plan.cpp, "canonical joint axes"), so the axis is a compile-time constant
// there and every axis-dependent selection folds away.  The general kernels keep the run-time axis.
template <bool GEN>
__device__ __forceinline__ BodyRec load_body(cptr<BodyRec> p)
{
    BodyRec b = load_rec(p);
    if constexpr (!GEN) b.axis = 2;
    return b;
}

template <class T>
struct Tables {
    cptr<Step> steps;
    cptr<ClusterRec> clusters;
    cptr<BodyRec> bodies;
    cptr<T> consts;
    cptr<int32_t> cints;
    cptr<int32_t> acc_k;
    int n_steps, nq, nv, ori_repr;
    T a_root[6];
};
template <class T>
__device__ __forceinline__ Tables<T> make_tables(const DevPlan<T> &P)
{
    Tables<T> t;
    t.steps = (cptr<Step>)P.steps;
    t.clusters = (cptr<ClusterRec>)P.clusters;
    t.bodies = (cptr<BodyRec>)P.bodies;
    t.consts = (cptr<T>)P.consts;
    t.cints = (cptr<int32_t>)P.cints;
    t.acc_k = (cptr<int32_t>)P.acc_k;
    t.n_steps = P.n_steps;
    t.nq = P.nq;
    t.nv = P.nv;
    t.ori_repr = P.ori_repr;
#pragma unroll
    for (int i = 0; i < 6; i++) t.a_root[i] = P.a_root[i];
    return t;
}

// ---------------------------------------------------------------------------------------------
// slot store: an object is either wholly in LDS or wholly in the wave's global slab (plan.cpp)
// ---------------------------------------------------------------------------------------------

// SPLIT policy (fast kernels, layouts with Layout::split): the [K | y0] blocks -- written once by the backward sweep,
// read once by the acceleration sweep, 7 scalars per DoF and far too many for LDS -- and the backward accumulators of
// branching bodies (27 scalars, touched once per child limb) ALWAYS live in the wave's global slab, every other
// object ALWAYS in LDS.  No access then tests kSlotGlobal: the scalar branch, the
// duplicated LDS / global code behind it and the control-flow scaffolding around both disappear from the kernel.
template <class T, bool SPLIT = false>
struct Slots {
    T *glb;  // wave's global slab (uniform: the per-lane part of an address is a 32-bit offset, which
             // keeps the 64-bit address arithmetic to one vector add per object)
    int lane;
    // lanes of this tile whose D = S^T IA S had a pivot that is not positive (chain_kernels.hip, ChainMem::bad: the same counting)
    mutable unsigned long long bad = 0;
    __device__ __forceinline__ void pivot(T d) const { bad |= __builtin_amdgcn_ballot_w64(!(d > T(0))); }
    __device__ __forceinline__ void flush_bad(unsigned long long *counter, int rows_valid) const
    {
        const unsigned long long live = rows_valid >= kWave ? ~0ull : ((1ull << rows_valid) - 1ull);
        const unsigned long long b = bad & live;
        if (b && counter && lane == 0) atomicAdd(counter, (unsigned long long)__builtin_popcountll(b));
        bad = 0;
    }

    __device__ __forceinline__ T lds_get(int s) const { return reinterpret_cast<T *>(grbda_smem)[s * kWave + lane]; }
    __device__ __forceinline__ void lds_put(int s, T x) const { reinterpret_cast<T *>(grbda_smem)[s * kWave + lane] = x; }

    template <int N>
    __device__ __forceinline__ void ld_glb(int s, T (&x)[N]) const
    {
        const T *p = glb + (size_t)(unsigned)((s & ~kSlotGlobal) * kWave + lane);
#pragma unroll
        for (int i = 0; i < N; i++) x[i] = p[i * kWave];
    }
    template <int N>
    __device__ __forceinline__ void st_glb(int s, const T (&x)[N]) const
    {
        T *p = glb + (size_t)(unsigned)((s & ~kSlotGlobal) * kWave + lane);
#pragma unroll
        for (int i = 0; i < N; i++) p[i * kWave] = x[i];
    }
    template <int N>
    __device__ __forceinline__ void ld(int s, T (&x)[N]) const
    {
        if (!SPLIT && (s & kSlotGlobal)) {
            ld_glb(s, x);
        } else {
#pragma unroll
            for (int i = 0; i < N; i++) x[i] = lds_get(s + i);
        }
    }
    template <int N>
    __device__ __forceinline__ void st(int s, const T (&x)[N]) const
    {
        if (!SPLIT && (s & kSlotGlobal)) {
            st_glb(s, x);
        } else {
#pragma unroll
            for (int i = 0; i < N; i++) lds_put(s + i, x[i]);
        }
    }
    // [K | y0] blocks
    template <int N>
    __device__ __forceinline__ void ldK(int s, T (&x)[N]) const
    {
        if constexpr (SPLIT) ld_glb(s, x);
        else ld(s, x);
    }
    template <int N>
    __device__ __forceinline__ void stK(int s, const T (&x)[N]) const
    {
        if constexpr (SPLIT) st_glb(s, x);
        else st(s, x);
    }
    __device__ __forceinline__ T ldK1(int s) const
    {
        T x[1];
        ldK(s, x);
        return x[0];
    }
    __device__ __forceinline__ void stK1(int s, T v) const
    {
        const T x[1] = {v};
        stK(s, x);
    }
    __device__ __forceinline__ T ld1(int s) const
    {
        T x[1];
        ld(s, x);
        return x[0];
    }
    __device__ __forceinline__ void st1(int s, T v) const
    {
        const T x[1] = {v};
        st(s, x);
    }
    // the same for objects of the global class ([K | y0] blocks, backward accumulators)
    template <int N>
    __device__ __forceinline__ void accK(int s, const T (&x)[N], int first) const
    {
        if (first) {
            stK(s, x);
        } else {
            T y[N];
            ldK(s, y);
#pragma unroll
            for (int i = 0; i < N; i++) y[i] += x[i];
            stK(s, y);
        }
    }
    // x is stored when first != 0, accumulated otherwise
    template <int N>
    __device__ __forceinline__ void acc(int s, const T (&x)[N], int first) const
    {
        if (first) {
            st(s, x);
        } else {
            T y[N];
            ld(s, y);
#pragma unroll
            for (int i = 0; i < N; i++) y[i] += x[i];
            st(s, y);
        }
    }
};

// ---------------------------------------------------------------------------------------------
// per-lane view of one state of the batch
// ---------------------------------------------------------------------------------------------
template <class T>
struct Lane {
    // the tile's inputs, transposed once per tile into coordinate-major rows of the wave's global
    // slab (row j of q at in_q[j * 64], one coalesced 64-element row per coordinate)
    const T *in_q, *in_qd, *in_x;  // x: tau (ABA) or ydd (RNEA); all already offset by the lane
    // results (ydd of the ABA, tau of the RNEA) go to the x rows of the slab -- coordinate j's input is dead
    // by the time its result exists -- and leave the tile through one coalesced transposition
    // (write_outputs): a direct store would scatter every coordinate over 64 cache lines.
    T *out_rows;
    __device__ __forceinline__ void put(int j, T v) const { out_rows[(size_t)j * kWave] = v; }
    const T *fext;                 // this state's [n_bodies][6] world-frame external forces, or nullptr
    bool active;
    __device__ __forceinline__ T q(int j) const { return in_q[(size_t)j * kWave]; }
    __device__ __forceinline__ T qd(int j) const { return in_qd[(size_t)j * kWave]; }
    __device__ __forceinline__ T x(int j) const { return in_x[(size_t)j * kWave]; }
    // coordinate a of the current step's cluster
    int lane;
    __device__ __forceinline__ T cy(const ClusterRec &c, int a) const { return q(c.q_index + a); }
    __device__ __forceinline__ T cyd(const ClusterRec &c, int a) const { return qd(c.v_index + a); }
    __device__ __forceinline__ T cx(const ClusterRec &c, int a) const { return x(c.v_index + a); }
};

template <class T>
struct Carry {
    T IA[21];
    T psi[6];
};


// spanning joint value of body i: row i of G times the independent cluster coordinates
// (LoopConstraint::Static::gamma, LoopConstraint.cpp:49-52; ClusterJoint.cpp:55-58)
template <class T, int N>
__device__ __forceinline__ T gdot(cptr<T> G, const T (&y)[N])
{
    T s = 0;
#pragma unroll
    for (int a = 0; a < N; a++) s += G[a] * y[a];
    return s;
}

// ---------------------------------------------------------------------------------------------
// implicit position-loop constraints (LoopConstraint::GenericImplicit built by
// ClusterTreeParsing.cpp:310-376; the reference evaluates CasADi functions K, G, k, g per state,
// GenericJoint.cpp:117-129).  Here K = d phi / d q comes from the geometric Jacobian of the two
// sub-chains NCA -> predecessor / successor, k = -Kdot qd from their velocity-product
// accelerations, G = P [1; -Kd^-1 Ki], g = P [0; Kd^-1 k] (GenericJoint.cpp:57-90).
// Per-step scratch block (slots, see plan.h): G rows k*(N+1) | K rows*k | qd_span k | q_span k | chain 6k
// ---------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void cross3(const T (&a)[3], const T (&b)[3], T (&o)[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// inverse of an R x R matrix held in a 3x3 array, R in {1,2,3} (wave-uniform), closed form
template <class T>
__device__ __forceinline__ void inv_small(int R, const T (&A)[3][3], T (&Ai)[3][3])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Ai[i][j] = 0;
    if (R == 1) {
        Ai[0][0] = T(1) / A[0][0];
    } else if (R == 2) {
        const T id = T(1) / (A[0][0] * A[1][1] - A[0][1] * A[1][0]);
        Ai[0][0] = A[1][1] * id; Ai[0][1] = -A[0][1] * id;
        Ai[1][0] = -A[1][0] * id; Ai[1][1] = A[0][0] * id;
    } else {
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
    }
}

// Scratch of an implicit cluster.  The constraint is evaluated ONCE per evaluation, in the first sweep that
// visits the cluster; what the later sweeps need -- [G rows k*(n+1), last column g][qd_span k][q_span k] --
// lives in a block kept until the last sweep (base), the work space of the evaluation itself --
// [K rows*k][chain 6k] -- in a block that only lives during that step (tmp).
template <int N>
struct ImpLayout {
    int G, K, qds, qs, chain;
    __device__ __forceinline__ ImpLayout(int base, int tmp, int k, int rows)
    {
        G = base;
        qds = G + k * (N + 1);
        qs = qds + k;
        K = tmp;
        chain = K + rows * k;
    }
};

// walk one sub-chain with positions only: joint axes a_t and origins o_t (NCA coordinates) go to the
// chain scratch, the constraint point p is returned
template <class T, class SL>
__device__ __forceinline__ void loop_chain_positions(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                                     cptr<int32_t> subs, int len, cptr<T> origin, int qs_slot,
                                                     int chain_slot, T (&p)[3])
{
    T E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, r[3] = {0, 0, 0};
    for (int t = 0; t < len; t++) {
        const int sub = subs[t];
        const BodyRec b = load_rec(P.bodies + (c.first_body + sub));
        cptr<T> C = P.consts + b.cofs;
        T sc[2], Eb[9], En[9];
        sincos_precise(S.ld1(qs_slot + sub), &sc[0], &sc[1]);
        build_E(b.axis, sc[0], sc[1], C, Eb);
        // X_new = (Eb, r_tree) * (E, r):  E_new = Eb E,  r_new = r + E^T r_tree
#pragma unroll
        for (int i = 0; i < 3; i++) r[i] += E[i] * C[9] + E[3 + i] * C[10] + E[6 + i] * C[11];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) En[3 * i + j] = Eb[3 * i] * E[j] + Eb[3 * i + 1] * E[3 + j] + Eb[3 * i + 2] * E[6 + j];
#pragma unroll
        for (int i = 0; i < 9; i++) E[i] = En[i];
        // joint axis in NCA coordinates = row `axis` of E; joint origin = r
        T ao[6];
        if (b.axis == 0) { ao[0] = E[0]; ao[1] = E[1]; ao[2] = E[2]; }
        else if (b.axis == 1) { ao[0] = E[3]; ao[1] = E[4]; ao[2] = E[5]; }
        else { ao[0] = E[6]; ao[1] = E[7]; ao[2] = E[8]; }
        ao[3] = r[0]; ao[4] = r[1]; ao[5] = r[2];
        S.st(chain_slot + 6 * t, ao);
    }
    // constraint frame origin: p = r + E^T r_origin (the origin's own rotation does not move the point)
#pragma unroll
    for (int i = 0; i < 3; i++) p[i] = r[i] + E[i] * origin[9] + E[3 + i] * origin[10] + E[6 + i] * origin[11];
}

// ---- K(q) and Kdot*qd of URDF+ position loops --------------------------------------------------
template <class T, int N, class SL>
__device__ __forceinline__ void loop_position_K(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                                const ImpLayout<N> &lay, cptr<int32_t> loops, int n_loops,
                                                T *phi = nullptr /* [3]: p_pred - p_succ on the masked axes */)
{
    if (phi) phi[0] = phi[1] = phi[2] = 0;
    const int k = c.k;
    cptr<int32_t> lp = loops;
    int row0 = 0;
    for (int l = 0; l < n_loops; l++) {
        const int np = lp[0], ns = lp[1 + np], mask = lp[2 + np + ns];
        cptr<T> org = P.consts + c.dofs + 24 * l;
        for (int side = 0; side < 2; side++) {
            cptr<int32_t> subs = side == 0 ? lp + 1 : lp + 2 + np;
            const int len = side == 0 ? np : ns;
            T p[3];
            loop_chain_positions(P, S, c, subs, len, org + 12 * side, lay.qs, lay.chain, p);
            const T sgn = side == 0 ? T(1) : T(-1);
            if (phi) {
                int row = row0;
#pragma unroll
                for (int ax = 0; ax < 3; ax++)
                    if (mask & (1 << ax)) {
                        if (row == 0) phi[0] += sgn * p[ax];
                        else if (row == 1) phi[1] += sgn * p[ax];
                        else phi[2] += sgn * p[ax];
                        row++;
                    }
            }
            for (int t = 0; t < len; t++) {
                T ao[6];
                S.ld(lay.chain + 6 * t, ao);
                const T a[3] = {ao[0], ao[1], ao[2]}, d[3] = {p[0] - ao[3], p[1] - ao[4], p[2] - ao[5]};
                T J[3];
                cross3(a, d, J);
                int row = row0;
                const int sub = subs[t];
#pragma unroll
                for (int ax = 0; ax < 3; ax++)
                    if (mask & (1 << ax)) {
                        S.st1(lay.K + row * k + sub, sgn * J[ax]);
                        row++;
                    }
            }
        }
        row0 += ((mask >> 0) & 1) + ((mask >> 1) & 1) + ((mask >> 2) & 1);
        lp += 3 + np + ns;
    }
}

// velocity-product acceleration of the constraint points: (Kdot qd)[row]
template <class T, int N, class SL>
__device__ __forceinline__ void loop_position_Kdqd(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                                   const ImpLayout<N> &lay, cptr<int32_t> loops, int n_loops, T (&kdq)[3])
{
    cptr<int32_t> lp = loops;
    int row0 = 0;
    for (int l = 0; l < n_loops; l++) {
        const int np = lp[0], ns = lp[1 + np], mask = lp[2 + np + ns];
        cptr<T> org = P.consts + c.dofs + 24 * l;
        T acc[3] = {0, 0, 0};
        for (int side = 0; side < 2; side++) {
            cptr<int32_t> subs = side == 0 ? lp + 1 : lp + 2 + np;
            const int len = side == 0 ? np : ns;
            T p[3];
            loop_chain_positions(P, S, c, subs, len, org + 12 * side, lay.qs, lay.chain, p);
            T w[3] = {0, 0, 0}, al[3] = {0, 0, 0}, ao_[3] = {0, 0, 0}, op[3] = {0, 0, 0};
            for (int t = 0; t <= len; t++) {
                T a[3] = {0, 0, 0}, o[3];
                T qd_t = 0;
                if (t < len) {
                    T rec[6];
                    S.ld(lay.chain + 6 * t, rec);
                    a[0] = rec[0]; a[1] = rec[1]; a[2] = rec[2];
                    o[0] = rec[3]; o[1] = rec[4]; o[2] = rec[5];
                    qd_t = S.ld1(lay.qds + subs[t]);
                } else {
                    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
                }
                // the point o is carried rigidly by the previous frame (w, al at origin op)
                const T d[3] = {o[0] - op[0], o[1] - op[1], o[2] - op[2]};
                T wd[3], wwd[3], ad[3];
                cross3(w, d, wd);
                cross3(w, wd, wwd);
                cross3(al, d, ad);
#pragma unroll
                for (int i = 0; i < 3; i++) { ao_[i] += ad[i] + wwd[i]; op[i] = o[i]; }
                // then the joint adds its own rate about a (qdd = 0)
                const T aq[3] = {a[0] * qd_t, a[1] * qd_t, a[2] * qd_t};
                T waq[3];
                cross3(w, aq, waq);
#pragma unroll
                for (int i = 0; i < 3; i++) { al[i] += waq[i]; w[i] += aq[i]; }
            }
            const T sgn = side == 0 ? T(1) : T(-1);
#pragma unroll
            for (int i = 0; i < 3; i++) acc[i] += sgn * ao_[i];
        }
        int row = row0;
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
            if (mask & (1 << ax)) {
                if (row == 0) kdq[0] = acc[ax];
                else if (row == 1) kdq[1] = acc[ax];
                else kdq[2] = acc[ax];
                row++;
            }
        row0 = row;
        lp += 3 + np + ns;
    }
}

// ---- trig-polynomial constraints: phi_r = sum_t coef * prod_f F_f(w_f . q + b_f), F in {id, sin, cos} ----
// (the hand-written phi lambdas of the Tello differentials, src/Robots/Tello.cpp:139-163,237-261; the
// reference differentiates them with CasADi, here analytically).  want_K: K rows to scratch;
// otherwise the second directional derivative along qd_span goes to kdq.
template <class T, int N, class SL>
__device__ __forceinline__ void trig_poly_eval(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                               const ImpLayout<N> &lay, cptr<int32_t> prog, bool want_K, T (&kdq)[3],
                                               T *phi = nullptr /* [3]: constraint values, with want_K */)
{
    const int k = c.k;
    cptr<int32_t> ip = prog;
    const int n_args = *ip++;
    cptr<T> ap = P.consts + c.dofs;          // per distinct argument: w[k], b
    cptr<T> cp = ap + n_args * (k + 1);      // per term: coef
    // spanning positions / velocities once into registers (k <= 8, static indexing)
    T qv[kMaxClusterBodies];
#pragma unroll
    for (int j = 0; j < kMaxClusterBodies; j++)
        qv[j] = j < k ? S.ld1((want_K ? lay.qs : lay.qds) + j) : T(0);
    // every distinct argument once: first pass a = w.q + b with its sine and cosine, second pass w.qd
    // (work space [a, sin a, cos a, w.qd] per argument, in place of the position loops' chain scratch)
    for (int i = 0; i < n_args; i++) {
        cptr<T> w = ap + i * (k + 1);
        T a = want_K ? w[k] : T(0);
#pragma unroll
        for (int j = 0; j < kMaxClusterBodies; j++)
            if (j < k) a += w[j] * qv[j];
        if (want_K) {
            T sn, cs;
            sincos_precise(a, &sn, &cs);
            const T v3[3] = {a, sn, cs};
            S.st(lay.chain + 4 * i, v3);
        } else {
            S.st1(lay.chain + 4 * i + 3, a);
        }
    }
    for (int r = 0; r < c.rows; r++) {
        const int nt = *ip++;
        T Krow[kMaxClusterBodies];
#pragma unroll
        for (int j = 0; j < kMaxClusterBodies; j++) Krow[j] = 0;
        T kd = 0, ph = 0;
        for (int t = 0; t < nt; t++) {
            const int nf = *ip++;
            const T coef = *cp++;
            T f0[4], f1[4], f2[4], ad[4];
            cptr<T> wv[4];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                f0[f] = 1; f1[f] = 0; f2[f] = 0; ad[f] = 0;
                wv[f] = ap;
                if (f < nf) {
                    const int type = ip[0], arg = ip[1];
                    ip += 2;
                    wv[f] = ap + arg * (k + 1);
                    T v4[4];
                    S.ld(lay.chain + 4 * arg, v4);  // a, sin a, cos a, w.qd (the last one only in the second pass)
                    ad[f] = want_K ? T(0) : v4[3];
                    if (type == 1) { f0[f] = v4[1]; f1[f] = v4[2]; f2[f] = -v4[1]; }
                    else if (type == 2) { f0[f] = v4[2]; f1[f] = -v4[1]; f2[f] = -v4[2]; }
                    else { f0[f] = v4[0]; f1[f] = 1; f2[f] = 0; }
                }
            }
            if (phi) ph += coef * f0[0] * f0[1] * f0[2] * f0[3];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                if (f < nf) {
                    T others = coef;
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        if (g != f) others *= f0[g];
                    if (want_K) {
#pragma unroll
                        for (int j = 0; j < kMaxClusterBodies; j++)
                            if (j < k) Krow[j] += others * f1[f] * wv[f][j];
                    } else {
                        kd += others * f2[f] * ad[f] * ad[f];
#pragma unroll
                        for (int g = 0; g < 4; g++)
                            if (g != f && g < nf) {
                                T rest = coef;
#pragma unroll
                                for (int h = 0; h < 4; h++)
                                    if (h != f && h != g) rest *= f0[h];
                                kd += rest * f1[f] * ad[f] * f1[g] * ad[g];
                            }
                    }
                }
            }
        }
        if (want_K) {
#pragma unroll
            for (int j = 0; j < kMaxClusterBodies; j++)
                if (j < k) S.st1(lay.K + r * k + j, Krow[j]);
            if (phi) {
                if (r == 0) phi[0] = ph;
                else if (r == 1) phi[1] = ph;
                else phi[2] = ph;
            }
        } else {
            if (r == 0) kdq[0] = kd;
            else if (r == 1) kdq[1] = kd;
            else kdq[2] = kd;
        }
    }
}

// G rows, g, spanning positions / velocities of an implicit cluster into the scratch block
template <class T, int N, class SL>
__device__ __noinline__ void eval_loop_constraint(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                                  const Lane<T> &L, const T (&yd)[N], bool want_bias)
{
    const ImpLayout<N> lay(c.slot_imp_fwd, c.slot_imp_bwd, c.k, c.rows);
    const int k = c.k, rows = c.rows;
    cptr<int32_t> ip = P.cints + c.iofs;
    const int hdr0 = ip[0];  // number of loops (position loops)
    const int n_ind = ip[1];
    cptr<int32_t> ind = ip + 2;
    cptr<int32_t> dep = ip + 3 + n_ind;
    cptr<int32_t> payload = dep + rows;
    T kdq[3] = {0, 0, 0};
#ifdef GRBDA_EXP_NO_CONSTRAINT
    return;
#endif

    for (int i = 0; i < k; i++) S.st1(lay.qs + i, L.q(c.q_index + i));
    for (int i = 0; i < rows * k; i++) S.st1(lay.K + i, T(0));

    // ---- K(q) ------------------------------------------------------------------------------------
    if (c.cons_type == 0) loop_position_K<T, N>(P, S, c, lay, payload, hdr0);
#ifndef GRBDA_EXP_NO_TRIG
    else trig_poly_eval<T, N>(P, S, c, lay, payload, true, kdq);
#endif

    // ---- G = P [1; -Kd^-1 Ki] ------------------------------------------------------------------
    T Kd[3][3], Kdi[3][3], X[3][N];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) Kd[r][j] = (r < rows && j < rows) ? S.ld1(lay.K + r * k + dep[j]) : T(r == j);
    inv_small(rows, Kd, Kdi);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int a = 0; a < N; a++) {
            T s = 0;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (r < rows && j < rows) s += Kdi[r][j] * S.ld1(lay.K + j * k + ind[a]);
            X[r][a] = s;
        }
#pragma unroll
    for (int a = 0; a < N; a++) {
        T row[N + 1];
#pragma unroll
        for (int bb = 0; bb <= N; bb++) row[bb] = bb == a ? T(1) : T(0);
        S.st(lay.G + ind[a] * (N + 1), row);
    }
#pragma unroll
    for (int r = 0; r < 3; r++)
        if (r < rows) {
            T row[N + 1];
#pragma unroll
            for (int a = 0; a < N; a++) row[a] = -X[r][a];
            row[N] = 0;
            S.st(lay.G + dep[r] * (N + 1), row);
        }
    // spanning velocities qd_span = G yd
    for (int i = 0; i < k; i++) {
        T row[N + 1];
        S.ld(lay.G + i * (N + 1), row);
        T s = 0;
#pragma unroll
        for (int a = 0; a < N; a++) s += row[a] * yd[a];
        S.st1(lay.qds + i, s);
    }
    if (!want_bias) return;

    // ---- k = -Kdot qd ; g = P [0; Kd^-1 k] ----------------------------------------------------------
    if (c.cons_type == 0) loop_position_Kdqd<T, N>(P, S, c, lay, payload, hdr0, kdq);
#ifndef GRBDA_EXP_NO_TRIG
    else trig_poly_eval<T, N>(P, S, c, lay, payload, false, kdq);
#endif
#pragma unroll
    for (int r = 0; r < 3; r++)
        if (r < rows) S.st1(lay.G + dep[r] * (N + 1) + N, -(Kdi[r][0] * kdq[0] + Kdi[r][1] * kdq[1] + Kdi[r][2] * kdq[2]));
}

// coupling of body i of a revolute cluster: spanning angle, row of G, bias g_i.
// Explicit clusters: constants (LoopConstraint::Static, LoopConstraint.cpp:38-52), kept on the
// scalar side (ConstRow: the row stays in SGPRs / the constant cache);
// implicit clusters: per state, from the step's scratch block (eval_loop_constraint).
template <class T>
struct ConstRow {
    cptr<T> p;
    __device__ __forceinline__ T operator[](int a) const { return p[a]; }
};
template <class T, int N>
struct RegRow {
    T v[N];
    __device__ __forceinline__ T operator[](int a) const { return v[a]; }
};
template <class T, int N, bool LOOP>
struct RowSel {
    using type = ConstRow<T>;
};
template <class T, int N>
struct RowSel<T, N, true> {
    using type = RegRow<T, N>;
};

template <class T, int N, bool LOOP, class SL>
__device__ __forceinline__ void body_coupling(const Tables<T> &P, const SL &S, const ClusterRec &c, int imp_base,
                                              int i, cptr<T> C, const T (&y)[N], T &qi,
                                              typename RowSel<T, N, LOOP>::type &Gr, T &gi)
{
    if constexpr (LOOP) {
        const ImpLayout<N> lay(imp_base, 0, c.k, c.rows);
        T row[N + 1];
        S.ld(lay.G + i * (N + 1), row);
#pragma unroll
        for (int a = 0; a < N; a++) Gr.v[a] = row[a];
        gi = row[N];
        qi = S.ld1(lay.qs + i);
    } else {
        Gr.p = C + kBodyConstFixed;
        T s = 0;
#pragma unroll
        for (int a = 0; a < N; a++) s += Gr.p[a] * y[a];
        qi = s;
        gi = 0;
    }
}
template <class T, int N, class Row>
__device__ __forceinline__ T rdot(const Row &G, const T (&y)[N])
{
    T s = 0;
#pragma unroll
    for (int a = 0; a < N; a++) s += G[a] * y[a];
    return s;
}

// ---------------------------------------------------------------------------------------------
// external forces (TreeModel::setExternalForces, TreeModel.cpp:214-239): forces are given in world
// coordinates, the bias force gets  -Xa.transformForceVector(f_ext)  (ClusterTreeDynamics.cpp:101-105,
// SpatialTransforms.cpp:62-71,234-250) with Xa the absolute transform world -> body (TreeNode::Xa_,
// TreeModel.cpp:20-27).  Only the general kernel variant carries this code.
// ---------------------------------------------------------------------------------------------
// Xa = X * Xa_parent:  E = E_i E_p,  r = r_p + E_p^T r_i   (SpatialTransforms.cpp:149-157)
template <class T, class R3, class SL>
__device__ __forceinline__ void compose_absolute(const SL &S, int parent_slot_Xa, const T (&E)[9], R3 r, T (&Xa)[12])
{
    if (parent_slot_Xa >= 0) {
        T Xp[12];
        S.ld(parent_slot_Xa, Xp);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Xa[3 * i + j] = E[3 * i] * Xp[j] + E[3 * i + 1] * Xp[3 + j] + E[3 * i + 2] * Xp[6 + j];
#pragma unroll
        for (int i = 0; i < 3; i++) Xa[9 + i] = Xp[9 + i] + Xp[i] * r[0] + Xp[3 + i] * r[1] + Xp[6 + i] * r[2];
    } else {
#pragma unroll
        for (int i = 0; i < 9; i++) Xa[i] = E[i];
#pragma unroll
        for (int i = 0; i < 3; i++) Xa[9 + i] = r[i];
    }
}
// f -= Xa.transformForceVector(f_ext[body])
template <class T>
__device__ __forceinline__ void subtract_external_force(const Lane<T> &L, int body, const T (&Xa)[12], T (&f)[6])
{
    T w[6];
#pragma unroll
    for (int j = 0; j < 6; j++) w[j] = L.fext[(size_t)body * 6 + j];
    const T t0 = w[0] - (Xa[10] * w[5] - Xa[11] * w[4]);
    const T t1 = w[1] - (Xa[11] * w[3] - Xa[9] * w[5]);
    const T t2 = w[2] - (Xa[9] * w[4] - Xa[10] * w[3]);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        f[i] -= Xa[3 * i] * t0 + Xa[3 * i + 1] * t1 + Xa[3 * i + 2] * t2;
        f[3 + i] -= Xa[3 * i] * w[3] + Xa[3 * i + 1] * w[4] + Xa[3 * i + 2] * w[5];
    }
}

// kinematics of one revolute body: joint transform and spatial velocity.  Bodies with children
// were handled by the forward sweep (sin/cos and v are in their slots); leaf bodies are evaluated
// here from the parent's stored velocity, so they never occupy a slot.
template <class T, class SL>
__device__ __forceinline__ void body_kinematics(const Tables<T> &P, const SL &S, const BodyRec &b, cptr<T> C,
                                                T qi, T qdi, T (&sc)[2], T (&E)[9], T (&v)[6])
{
    if (b.has_child) {
        S.ld(b.slot_sc, sc);
        S.ld(b.slot_v, v);
        build_E(b.axis, sc[0], sc[1], C, E);
    } else {
        if (b.axisym) {
            // rotor: every quantity handed to the parent is independent of the joint angle; use q = 0
            sc[0] = 0;
            sc[1] = 1;
        } else {
            sincos_t(qi, &sc[0], &sc[1]);
        }
        build_E(b.axis, sc[0], sc[1], C, E);
        if (b.parent >= 0) {
            T vp[6];
            S.ld(b.parent_slot_v, vp);
            xmotion(E, C + 9, vp, v);
        } else {
#pragma unroll
            for (int j = 0; j < 6; j++) v[j] = 0;
        }
        add_axis(v, b.axis, qdi);
    }
}

// ---------------------------------------------------------------------------------------------
// ABA sweep 1: ClusterTreeNode::updateKinematics + TreeModel::forwardKinematics
// (ClusterTreeNode.cpp:26-31, TreeModel.cpp:6-32) -- only bodies that have children
// ---------------------------------------------------------------------------------------------
template <class T, int N, bool LOOP, bool GEN, class SL>
__device__ __forceinline__ void aba_fwd_static(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                               const Lane<T> &L)
{
    T y[N], yd[N];
#pragma unroll
    for (int a = 0; a < N; a++) {
        y[a] = L.cy(c, a);
        yd[a] = L.cyd(c, a);
    }
    const int imp = c.slot_imp_fwd;
    if constexpr (LOOP) eval_loop_constraint<T, N>(P, S, c, L, yd, true);  // kept for the later sweeps
    for (int i = 0; i < c.k; i++) {
        if (!((c.child_mask >> i) & 1)) continue;  // leaf bodies are evaluated inside the backward step
        const BodyRec b = load_body<GEN>(P.bodies + (c.first_body + i));
        cptr<T> C = P.consts + b.cofs;
        T qi, gi;
        typename RowSel<T, N, LOOP>::type Gr;
        body_coupling<T, N, LOOP>(P, S, c, imp, i, C, y, qi, Gr, gi);
        T sc[2], E[9], v[6];
        sincos_t(qi, &sc[0], &sc[1]);
        S.st(b.slot_sc, sc);
        build_E(b.axis, sc[0], sc[1], C, E);
        if (b.parent >= 0) {
            T vp[6];
            S.ld(b.parent_slot_v, vp);
            xmotion(E, C + 9, vp, v);
        } else {
#pragma unroll
            for (int j = 0; j < 6; j++) v[j] = 0;
        }
        add_axis(v, b.axis, rdot<T, N>(Gr, yd));
        S.st(b.slot_v, v);
        if constexpr (GEN) {
            if (L.fext) {
                T Xa[12];
                compose_absolute(S, b.parent_slot_Xa, E, C + 9, Xa);
                S.st(b.slot_Xa, Xa);
            }
        }
    }
}

// Free cluster (FreeJoint.cpp:28-46, Joint.h:61-68): Xup = XJ = (R(ori), position), v = yd.
// a' = Xup * (-gravity); needs only E because -gravity has no angular part in general? No:
// the general formula is kept.
// Xa of the floating base: (R(orientation), position)
template <class T>
__device__ __forceinline__ void free_absolute(const Tables<T> &P, const ClusterRec &c, const Lane<T> &L, T (&Xa)[12])
{
    T o[4], E[9];
    const int nori = P.ori_repr == 0 ? 4 : 3;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = j < nori ? L.q(c.q_index + 3 + j) : T(0);
    free_rotation(P.ori_repr, o, E);
#pragma unroll
    for (int j = 0; j < 9; j++) Xa[j] = E[j];
#pragma unroll
    for (int j = 0; j < 3; j++) Xa[9 + j] = L.q(c.q_index + j);
}

template <class T>
__device__ __forceinline__ void free_base_accel(const Tables<T> &P, const ClusterRec &c, const Lane<T> &L, T (&ag)[6])
{
    T o[4], E[9], r[3], g[6];
    const int nori = P.ori_repr == 0 ? 4 : 3;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = j < nori ? L.q(c.q_index + 3 + j) : T(0);
    free_rotation(P.ori_repr, o, E);
#pragma unroll
    for (int j = 0; j < 3; j++) r[j] = L.q(c.q_index + j);
#pragma unroll
    for (int j = 0; j < 6; j++) g[j] = P.a_root[j];
    xmotion(E, r, g, ag);
}

// ---------------------------------------------------------------------------------------------
// ABA sweep 2 (fused 2a + 2b): updateArticulatedBodies + bias back-propagation
// (ClusterTreeDynamics.cpp:94-129,157-191; ClusterTreeNode.cpp:33-37)
// ---------------------------------------------------------------------------------------------
template <class T, int N, bool LOOP, bool GEN, class SL>
__device__ __forceinline__ void aba_bwd_static(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                               const Lane<T> &L, Carry<T> &carry PROF_ARGS)
{
    T y[N], yd[N], u[N], F[6][N], D[N][N];
    // contribution of this cluster to its parent body when it is handed over in registers
    T pIA[21], ppsi[6];
#pragma unroll
    for (int j = 0; j < 21; j++) pIA[j] = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) ppsi[j] = 0;
#pragma unroll
    for (int a = 0; a < N; a++) {
        y[a] = L.cy(c, a);
        yd[a] = L.cyd(c, a);
        u[a] = L.cx(c, a);
#pragma unroll
        for (int r = 0; r < 6; r++) F[r][a] = 0;
#pragma unroll
        for (int bb = 0; bb < N; bb++) D[a][bb] = 0;
    }

    const int imp = c.slot_imp_fwd;  // evaluated by the forward step

    // in-cluster bias acceleration (the cJ part of GenericJoint.cpp:430-450), chained clusters only
    if (c.chained) {
        for (int i = 0; i < c.k; i++) {
            const BodyRec b = load_body<GEN>(P.bodies + (c.first_body + i));
            cptr<T> C = P.consts + b.cofs;
            T qi, gi;
            typename RowSel<T, N, LOOP>::type Gr;
            body_coupling<T, N, LOOP>(P, S, c, imp, i, C, y, qi, Gr, gi);
            const T qdi = rdot<T, N>(Gr, yd);
            T sc[2], E[9], v[6], ccl[6];
            body_kinematics<T>(P, S, b, C, qi, qdi, sc, E, v);
            vxaxis(b.axis, v, qdi, ccl);
            add_axis(ccl, b.axis, gi);  // S_implicit * g (GenericJoint.cpp:449-450)
            if (b.lam >= 0) {
                T cp[6], t[6];
                const BodyRec bl = load_body<GEN>(P.bodies + (b.lam));
                S.ld(bl.slot_ccl, cp);
                xmotion(E, C + 9, cp, t);
#pragma unroll
                for (int j = 0; j < 6; j++) ccl[j] += t[j];
            }
            S.st(b.slot_ccl, ccl);
        }
    }

    for (int i = c.k - 1; i >= 0; i--) {
        PROF_ADD(5);  // (previous body) joint-space terms + push up the in-cluster chain
        const BodyRec b = load_body<GEN>(P.bodies + (c.first_body + i));
        PROF_SYNC();
        PROF_ADD(6);  // body record round trip
        cptr<T> C = P.consts + b.cofs;
        cptr<T> Ic = C + 12;
        T qi, gi;
        typename RowSel<T, N, LOOP>::type G;
        body_coupling<T, N, LOOP>(P, S, c, imp, i, C, y, qi, G, gi);
        const T qdi = rdot<T, N>(G, yd);
        T sc[2], E[9], v[6];
        body_kinematics<T>(P, S, b, C, qi, qdi, sc, E, v);
        PROF_SYNC();
        PROF_ADD(7);  // constants + kinematics (LDS v / parent v)
        T chat[6];
        vxaxis(b.axis, v, qdi, chat);
        add_axis(chat, b.axis, gi);

        // composite articulated inertia and bias of this body
        T IA[21], psi[6];
        {
            T Iv[6];
            symv_c(Ic, v, Iv);
            crf(v, Iv, psi);  // pA = v x* (I v), ClusterTreeDynamics.cpp:95-98
        }
        if constexpr (GEN) {
            if (L.fext) {
                T Xa[12];
                if (b.has_child) S.ld(b.slot_Xa, Xa);
                else compose_absolute(S, b.parent_slot_Xa, E, C + 9, Xa);
                subtract_external_force(L, c.first_body + i, Xa, psi);
            }
        }
        // own inertia, plus the constant X0^T I X0 of axisymmetric leaf children (rotors) when there are any
        cptr<T> Ib = b.xofs >= 0 ? P.consts + b.xofs : Ic;
        if (b.carry_in) {
#pragma unroll
            for (int j = 0; j < 21; j++) IA[j] = Ib[j] + carry.IA[j];
#pragma unroll
            for (int j = 0; j < 6; j++) psi[j] += carry.psi[j];
        } else if (b.has_child) {
            T acc[21], pacc[6];
            S.ldK(b.slot_IA, acc);
            S.ldK(b.slot_psi, pacc);
#pragma unroll
            for (int j = 0; j < 21; j++) IA[j] = Ib[j] + acc[j];
#pragma unroll
            for (int j = 0; j < 6; j++) psi[j] += pacc[j];
        } else {
#pragma unroll
            for (int j = 0; j < 21; j++) IA[j] = Ib[j];
        }

        PROF_SYNC();
        PROF_ADD(8);  // own bias + accumulator loads
        T h[6];
        column(IA, b.axis, h);
        const T d = pick(h, b.axis);
        T bj = pick(psi, b.axis);
        if (c.chained) {
            T ccl[6];
            S.ld(b.slot_ccl, ccl);
#pragma unroll
            for (int j = 0; j < 6; j++) bj += h[j] * ccl[j];
        } else {
#pragma unroll
            for (int j = 0; j < 6; j++) bj += h[j] * chat[j];
        }

        // hand composite inertia and bias to the tree parent (in-cluster or parent cluster)
        if (b.parent >= 0) {
            T t[6], Ic_c[6], tp[6], Bc[21];
            symv(IA, chat, Ic_c);
#pragma unroll
            for (int j = 0; j < 6; j++) t[j] = psi[j] + Ic_c[j];
            xforce_inv(E, C + 9, t, tp);
            if (b.axisym) {
                // X0^T I X0 is already part of the parent's constants (xofs)
                if (c.carry_out && b.lam < 0) {
#pragma unroll
                    for (int j = 0; j < 6; j++) ppsi[j] += tp[j];
                } else {
                    S.accK(b.parent_slot_psi, tp, b.acc_first);
                }
            } else {
                congruence(E, C + 9, IA, Bc);
                if (c.carry_out && b.lam < 0) {
#pragma unroll
                    for (int j = 0; j < 6; j++) ppsi[j] += tp[j];
#pragma unroll
                    for (int j = 0; j < 21; j++) pIA[j] += Bc[j];
                } else {
                    S.accK(b.parent_slot_psi, tp, b.acc_first);
                    S.accK(b.parent_slot_IA, Bc, b.acc_first_IA);
                }
            }
        }

        PROF_SYNC();
        PROF_ADD(9);  // IA*c, force transform, congruence, hand-over / accumulate
        // joint-space terms: D += d G^T G, u -= G^T b, push h up the in-cluster chain
#pragma unroll
        for (int a = 0; a < N; a++) {
            u[a] -= G[a] * bj;
#pragma unroll
            for (int bb = 0; bb < N; bb++) D[a][bb] += d * G[a] * G[bb];
        }
        T f[6];
        xforce_inv(E, C + 9, h, f);
        int l = b.lam;
        while (l >= 0) {
            const BodyRec bl = load_body<GEN>(P.bodies + (l));
            cptr<T> Cl = P.consts + bl.cofs;
            T ql, gl;
            typename RowSel<T, N, LOOP>::type Gl;
            body_coupling<T, N, LOOP>(P, S, c, imp, l - c.first_body, Cl, y, ql, Gl, gl);
            const T Hc = pick(f, bl.axis);
#pragma unroll
            for (int a = 0; a < N; a++)
#pragma unroll
                for (int bb = 0; bb < N; bb++) D[a][bb] += Hc * (Gl[a] * G[bb] + G[a] * Gl[bb]);
            T scl[2], El[9], f2[6];
            S.ld(bl.slot_sc, scl);
            build_E(bl.axis, scl[0], scl[1], Cl, El);
            xforce_inv(El, Cl + 9, f, f2);
#pragma unroll
            for (int j = 0; j < 6; j++) f[j] = f2[j];
            l = bl.lam;
        }
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int a = 0; a < N; a++) F[r][a] += f[r] * G[a];
    }

    PROF_ADD(10);
    // D^-1 u', K = D^-1 F^T
    Chol<T, N> ch;
    ch.factor(D);
#pragma unroll
    for (int a = 0; a < N; a++) S.pivot(ch.inv[a] < T(1e30) ? ch.inv[a] : T(0));  // (1 / sqrt(pivot): Inf for 0, NaN below it)
    ch.solve(u);
    T K[6 * N];
#pragma unroll
    for (int r = 0; r < 6; r++) {
        T col[N];
#pragma unroll
        for (int a = 0; a < N; a++) col[a] = F[r][a];
        ch.solve(col);
#pragma unroll
        for (int a = 0; a < N; a++) K[a * 6 + r] = col[a];
    }
    S.stK(c.slot_K, K);
    S.stK(c.slot_y0, u);

    // corrections on the parent body: IA_p -= F D^-1 F^T, pA_p += F D^-1 u'
    if (c.parent_body >= 0) {
        T dI[21], dp[6];
#pragma unroll
        for (int r = 0; r < 6; r++) {
            T s = 0;
#pragma unroll
            for (int a = 0; a < N; a++) s += F[r][a] * u[a];
            dp[r] = s;
#pragma unroll
            for (int cc = r; cc < 6; cc++) {
                T m = 0;
#pragma unroll
                for (int a = 0; a < N; a++) m -= F[r][a] * K[a * 6 + cc];
                dI[sidx(r, cc)] = m;
            }
        }
        if (c.carry_out) {
#pragma unroll
            for (int j = 0; j < 6; j++) carry.psi[j] = ppsi[j] + dp[j];
#pragma unroll
            for (int j = 0; j < 21; j++) carry.IA[j] = pIA[j] + dI[j];
        } else {
            S.accK(c.parent_slot_psi, dp, 0);
            S.accK(c.parent_slot_IA, dI, c.corr_first_IA);
        }
    }
    PROF_SYNC();
    PROF_ADD(11);  // solve, K/y0 stores, parent correction
}

// Free root: S = 1, D = IA, c = 0 (FreeJoint.cpp:10-36)
template <class T, class SL>
__device__ __forceinline__ void aba_bwd_free(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                             const Lane<T> &L, const Carry<T> &carry)
{
    const BodyRec b = load_rec(P.bodies + (c.first_body));
    cptr<T> Ic = P.consts + b.cofs + 12;
    T v[6], psi[6], Iv[6], IA[21];
#pragma unroll
    for (int j = 0; j < 6; j++) v[j] = L.qd(c.v_index + j);
    symv_c(Ic, v, Iv);
    crf(v, Iv, psi);
    if (L.fext) {
        T Xa[12];
        free_absolute(P, c, L, Xa);
        subtract_external_force(L, c.first_body, Xa, psi);
    }
    // own inertia, plus the constant X0^T I X0 of axisymmetric leaf children (rotors) when there are any
    cptr<T> Ib = b.xofs >= 0 ? P.consts + b.xofs : Ic;
    if (b.carry_in) {
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j] + carry.IA[j];
#pragma unroll
        for (int j = 0; j < 6; j++) psi[j] += carry.psi[j];
    } else if (b.has_child) {
        T acc[21], pacc[6];
        S.ldK(b.slot_IA, acc);
        S.ldK(b.slot_psi, pacc);
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j] + acc[j];
#pragma unroll
        for (int j = 0; j < 6; j++) psi[j] += pacc[j];
    } else {
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j];
    }
    T D[6][6], u[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        u[i] = L.x(c.v_index + i) - psi[i];
#pragma unroll
        for (int j = 0; j < 6; j++) D[i][j] = IA[sidx(i, j)];
    }
    Chol<T, 6> ch;
    ch.factor(D);
#pragma unroll
    for (int i = 0; i < 6; i++) S.pivot(ch.inv[i] < T(1e30) ? ch.inv[i] : T(0));
    ch.solve(u);
    S.stK(c.slot_y0, u);
}

// ---------------------------------------------------------------------------------------------
// Straight-line handlers of the two commonest cluster shapes (plan.h, ClusterRec::shape): a single
// revolute body, and the reference's RevoluteWithRotor (ClusterJoints/RevoluteWithRotorJoint.h) -- a
// link and an axisymmetric rotor leaf, both hanging off the parent body.  Same arithmetic as the generic
// handlers with n = 1, but no body loop, no accumulate-into-zero arrays and one combined hand-over to
// the parent, which removes most of the register shuffling the generic code needs.  Fast kernels only
// (canonical axes: every joint turns about z).
// ---------------------------------------------------------------------------------------------
template <class T, bool ROTOR, class SL>
__device__ __forceinline__ void aba_bwd_rev(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                            const Lane<T> &L, Carry<T> &carry)
{
    // (the plan compiler gives these shapes only to clusters with a parent body)
    const BodyRec b = load_rec(P.bodies + c.link_body);
    cptr<T> C = P.consts + b.cofs;
    const T g0 = C[kBodyConstFixed];
    const T yd = L.cyd(c, 0);
    const T qdi = g0 * yd;
    const bool has_child = b.has_child, carry_in = b.carry_in, carry_out = c.carry_out;

    // ---- link kinematics (TreeModel.cpp:6-32) ----
    T E[9], v[6], vp[6];
    if (has_child) {
        T sc[2];
        S.ld(b.slot_sc, sc);
        S.ld(b.slot_v, v);
        rotate_z(sc[0], sc[1], C, E);
        if constexpr (ROTOR) S.ld(b.parent_slot_v, vp);
    } else {
        T sn, cs;
        sincos_t(g0 * L.cy(c, 0), &sn, &cs);
        rotate_z(sn, cs, C, E);
        S.ld(b.parent_slot_v, vp);
        xmotion(E, C + 9, vp, v);
        v[2] += qdi;
    }
    // c = v x (z qd): (v1, -v0, 0, v4, -v3, 0) qd
    T chat[6] = {v[1] * qdi, -v[0] * qdi, 0, v[4] * qdi, -v[3] * qdi, 0};

    // ---- articulated inertia and bias of the link (ClusterTreeDynamics.cpp:93-129) ----
    cptr<T> Ic = C + 12;
    T IA[21], psi[6];
    {
        T Iv[6];
        symv_c(Ic, v, Iv);
        crf(v, Iv, psi);
    }
    // own inertia, plus the constant X0^T I X0 of axisymmetric leaf children (rotors) when there are any
    cptr<T> Ib = b.xofs >= 0 ? P.consts + b.xofs : Ic;
    if (carry_in) {
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j] + carry.IA[j];
#pragma unroll
        for (int j = 0; j < 6; j++) psi[j] += carry.psi[j];
    } else if (has_child) {
        T acc[21], pacc[6];
        S.ldK(b.slot_IA, acc);
        S.ldK(b.slot_psi, pacc);
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j] + acc[j];
#pragma unroll
        for (int j = 0; j < 6; j++) psi[j] += pacc[j];
    } else {
#pragma unroll
        for (int j = 0; j < 21; j++) IA[j] = Ib[j];
    }
    T h[6];
#pragma unroll
    for (int i = 0; i < 6; i++) h[i] = IA[sidx(i, 2)];
    const T bj = psi[2] + h[0] * chat[0] + h[1] * chat[1] + h[3] * chat[3] + h[4] * chat[4];
    T u = L.cx(c, 0) - g0 * bj;
    T D = h[2] * g0 * g0;
    T F[6];
    xforce_inv(E, C + 9, h, F);
#pragma unroll
    for (int r = 0; r < 6; r++) F[r] *= g0;

    T out_psi[6], out_IA[21];
    {
        T t[6];
        symv_z(IA, chat, t);
#pragma unroll
        for (int j = 0; j < 6; j++) t[j] += psi[j];
        xforce_inv(E, C + 9, t, out_psi);
        congruence(E, C + 9, IA, out_IA);
    }

    // ---- rotor: evaluated at q = 0 (plan.cpp, axisymmetric leaves) ----
    int first_psi = b.acc_first;
    if constexpr (ROTOR) {
        const BodyRec rb = load_rec(P.bodies + c.rotor_body);
        cptr<T> Cr = P.consts + rb.cofs;
        cptr<T> Ir = Cr + 12;
        const T gr = Cr[kBodyConstFixed];
        const T qdr = gr * yd;
        first_psi |= rb.acc_first;
        T E0[9], vr[6];
#pragma unroll
        for (int j = 0; j < 9; j++) E0[j] = Cr[j];
        xmotion(E0, Cr + 9, vp, vr);
        vr[2] += qdr;
        T cr[6] = {vr[1] * qdr, -vr[0] * qdr, 0, vr[4] * qdr, -vr[3] * qdr, 0};
        T pr[6], hr[6];
        {
            T Iv[6];
            symv_c(Ir, vr, Iv);
            crf(vr, Iv, pr);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) hr[i] = Ir[sidx(i, 2)];
        const T bjr = pr[2] + hr[0] * cr[0] + hr[1] * cr[1] + hr[3] * cr[3] + hr[4] * cr[4];
        u -= gr * bjr;
        D += hr[2] * gr * gr;
        T fr[6];
        xforce_inv(E0, Cr + 9, hr, fr);
#pragma unroll
        for (int r = 0; r < 6; r++) F[r] += fr[r] * gr;
        T t[6], tp[6];
        symv_z(Ir, cr, t);
#pragma unroll
        for (int j = 0; j < 6; j++) t[j] += pr[j];
        xforce_inv(E0, Cr + 9, t, tp);
#pragma unroll
        for (int j = 0; j < 6; j++) out_psi[j] += tp[j];
    }

    // ---- D^-1 u', K = D^-1 F^T (n = 1) ----
    const T Dinv = rcp_t(D);
    S.pivot(D);
    const T y0 = u * Dinv;
    T K[6];
#pragma unroll
    for (int r = 0; r < 6; r++) K[r] = F[r] * Dinv;
    S.stK(c.slot_K, K);
    S.stK1(c.slot_y0, y0);

    // ---- one combined hand-over to the parent body: X^T IA X - F D^-1 F^T,  X^T (pA + IA c) + F D^-1 u' ----
#pragma unroll
    for (int r = 0; r < 6; r++) {
        out_psi[r] += F[r] * y0;
#pragma unroll
        for (int cc = r; cc < 6; cc++) out_IA[sidx(r, cc)] -= F[r] * K[cc];
    }
    if (carry_out) {
#pragma unroll
        for (int j = 0; j < 6; j++) carry.psi[j] = out_psi[j];
#pragma unroll
        for (int j = 0; j < 21; j++) carry.IA[j] = out_IA[j];
    } else {
        S.accK(c.parent_slot_psi, out_psi, first_psi);
        S.accK(c.parent_slot_IA, out_IA, b.acc_first_IA | c.corr_first_IA);
    }
}

// acceleration sweep of the same shapes (ClusterTreeDynamics.cpp:131-152); a rotor has no children
template <class T, class SL>
__device__ __forceinline__ void aba_acc_rev(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                            const Lane<T> &L, const T (&kblk)[7], bool have_kblk PROF_ARGS)
{
    T K[6], ap[6], ydd;
    PROF_SYNCV();
    PROF_ADD(13);  // drain of everything outstanding at step entry
    if (have_kblk) {  // [K 6][y0 1] fetched while the previous step ran
#pragma unroll
        for (int r = 0; r < 6; r++) K[r] = kblk[r];
        ydd = kblk[6];
    } else {
        S.ldK(c.slot_K, K);
        ydd = S.ldK1(c.slot_y0);
    }
    S.ld(c.parent_slot_a3, ap);
    PROF_SYNCV();
    PROF_ADD(14);  // K, y0, parent acceleration loads
#pragma unroll
    for (int r = 0; r < 6; r++) ydd -= K[r] * ap[r];
    L.put(c.v_index, ydd);
    PROF_SYNCV();
    PROF_ADD(15);  // ydd + output store (acknowledged)
    if (!c.child_mask) return;
    const BodyRec b = load_rec(P.bodies + c.link_body);
    PROF_SYNC();
    PROF_ADD(16);  // body record
    cptr<T> C = P.consts + b.cofs;
    const T g0 = C[kBodyConstFixed];
    const T qdi = g0 * L.cyd(c, 0);
    T sn, cs, E[9], v[6], a[6];
    sincos_t(g0 * L.cy(c, 0), &sn, &cs);
    rotate_z(sn, cs, C, E);
    PROF_SYNC();
    PROF_ADD(17);  // constants, inputs, sincos, E
    {
        T vp[6];
        S.ld(b.parent_slot_v3, vp);
        xmotion(E, C + 9, vp, v);
    }
    xmotion(E, C + 9, ap, a);
    v[2] += qdi;
    T chat[6];
    vxaxis(2, v, qdi, chat);
#pragma unroll
    for (int j = 0; j < 6; j++) a[j] += chat[j];
    a[2] += g0 * ydd;
    S.st(b.slot_v3, v);
    S.st(b.slot_a3, a);
}

// forward sweep of the same shapes: only a link with children leaves anything behind
template <class T, class SL>
__device__ __forceinline__ void aba_fwd_rev(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                            const Lane<T> &L)
{
    if (!c.child_mask) return;
    const BodyRec b = load_rec(P.bodies + c.link_body);
    cptr<T> C = P.consts + b.cofs;
    const T g0 = C[kBodyConstFixed];
    T sc[2], E[9], v[6];
    sincos_t(g0 * L.cy(c, 0), &sc[0], &sc[1]);
    S.st(b.slot_sc, sc);
    rotate_z(sc[0], sc[1], C, E);
    {
        T vp[6];
        S.ld(b.parent_slot_v, vp);
        xmotion(E, C + 9, vp, v);
    }
    v[2] += g0 * L.cyd(c, 0);
    S.st(b.slot_v, v);
}

// ---------------------------------------------------------------------------------------------
// ABA sweep 3: joint accelerations (ClusterTreeDynamics.cpp:131-152).  Velocities of bodies with
// children are recomputed on the way down (cheaper than keeping them live across the sweeps).
// ---------------------------------------------------------------------------------------------
template <class T, int N, bool LOOP, bool GEN, class SL>
__device__ __forceinline__ void aba_acc_static(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                               const Lane<T> &L)
{
    T K[6 * N], ydd[N], ap[6];
#ifdef GRBDA_EXP_NOK
    for (int j = 0; j < 6 * N; j++) K[j] = T(0.01) * j;
    for (int j = 0; j < N; j++) ydd[j] = T(0.5);
#else
    S.ldK(c.slot_K, K);
    S.ldK(c.slot_y0, ydd);
#endif
    if (c.parent_slot_a3 >= 0) {
        S.ld(c.parent_slot_a3, ap);
    } else {
#pragma unroll
        for (int j = 0; j < 6; j++) ap[j] = P.a_root[j];
    }
#pragma unroll
    for (int a = 0; a < N; a++) {
        T s = ydd[a];
#pragma unroll
        for (int r = 0; r < 6; r++) s -= K[a * 6 + r] * ap[r];
        ydd[a] = s;
        L.put(c.v_index + a, s);
    }
    T y[N], yd[N];
#pragma unroll
    for (int a = 0; a < N; a++) {
        y[a] = L.cy(c, a);
        yd[a] = L.cyd(c, a);
    }
    const int imp = c.slot_imp_fwd;  // evaluated by the forward step
    for (int i = 0; i < c.k; i++) {
        if (!((c.child_mask >> i) & 1)) continue;  // nothing downstream needs this body's acceleration
        const BodyRec b = load_body<GEN>(P.bodies + (c.first_body + i));
        cptr<T> C = P.consts + b.cofs;
        T qi, gi;
        typename RowSel<T, N, LOOP>::type G;
        body_coupling<T, N, LOOP>(P, S, c, imp, i, C, y, qi, G, gi);
        T sc[2], E[9], v[6], a[6];
        sincos_t(qi, &sc[0], &sc[1]);
        build_E(b.axis, sc[0], sc[1], C, E);
        if (b.parent >= 0) {
            T vp[6], api[6];
            S.ld(b.parent_slot_v3, vp);
            S.ld(b.parent_slot_a3, api);
            xmotion(E, C + 9, vp, v);
            xmotion(E, C + 9, api, a);
        } else {
#pragma unroll
            for (int j = 0; j < 6; j++) v[j] = 0;
            xmotion(E, C + 9, ap, a);
        }
        const T qdi = rdot<T, N>(G, yd);
        add_axis(v, b.axis, qdi);
        T chat[6];
        vxaxis(b.axis, v, qdi, chat);
#pragma unroll
        for (int j = 0; j < 6; j++) a[j] += chat[j];
        add_axis(a, b.axis, rdot<T, N>(G, ydd) + gi);
        S.st(b.slot_v3, v);
        S.st(b.slot_a3, a);
    }
}

template <class T, class SL>
__device__ __forceinline__ void aba_acc_free(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                             const Lane<T> &L)
{
    const BodyRec b = load_rec(P.bodies + (c.first_body));
    T y0[6], ag[6];
    S.ldK(c.slot_y0, y0);
    free_base_accel(P, c, L, ag);
    // ydd = D^-1 u - D^-1 U^T a' with U = IA, D = IA  =>  ydd = y0 - a' ;  a = a' + ydd = y0
#pragma unroll
    for (int j = 0; j < 6; j++)
        L.put(c.v_index + j, y0[j] - ag[j]);
    if (b.has_child) {
        T v[6];
#pragma unroll
        for (int j = 0; j < 6; j++) v[j] = L.qd(c.v_index + j);
        S.st(b.slot_v3, v);
        S.st(b.slot_a3, y0);
    }
}

// free root, forward sweep: children only need its velocity
template <class T, class SL>
__device__ __forceinline__ void aba_fwd_free(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                             const Lane<T> &L)
{
    const BodyRec b = load_rec(P.bodies + (c.first_body));
    if (!b.has_child) return;
    T v[6];
#pragma unroll
    for (int j = 0; j < 6; j++) v[j] = L.qd(c.v_index + j);
    S.st(b.slot_v, v);
    if (L.fext && b.slot_Xa >= 0) {
        T Xa[12];
        free_absolute(P, c, L, Xa);
        S.st(b.slot_Xa, Xa);
    }
}

// ---------------------------------------------------------------------------------------------
// RNEA (TreeModel.cpp:34-57,173-212)
// ---------------------------------------------------------------------------------------------
template <class T, int N, bool LOOP, bool GEN, class SL>
__device__ __forceinline__ void rnea_fwd_static(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                                const Lane<T> &L)
{
    T y[N], yd[N], ydd[N];
#pragma unroll
    for (int a = 0; a < N; a++) {
        y[a] = L.cy(c, a);
        yd[a] = L.cyd(c, a);
        ydd[a] = L.cx(c, a);
    }
    const int imp = c.slot_imp_fwd;
    if constexpr (LOOP) eval_loop_constraint<T, N>(P, S, c, L, yd, true);  // kept for the backward step
    for (int i = 0; i < c.k; i++) {
        const BodyRec b = load_body<GEN>(P.bodies + (c.first_body + i));
        cptr<T> C = P.consts + b.cofs;
        T qi, gi;
        typename RowSel<T, N, LOOP>::type G;
        body_coupling<T, N, LOOP>(P, S, c, imp, i, C, y, qi, G, gi);
        const T qdi = rdot<T, N>(G, yd), qddi = rdot<T, N>(G, ydd) + gi;
        T sc[2], E[9], v[6], a[6];
        sincos_t(qi, &sc[0], &sc[1]);
        if (b.parent >= 0) S.st(b.slot_sc, sc);
        build_E(b.axis, sc[0], sc[1], C, E);
        if (b.parent >= 0) {
            T vp[6], apar[6];
            S.ld(b.parent_slot_v, vp);
            S.ld(b.parent_slot_a3, apar);
            xmotion(E, C + 9, vp, v);
            xmotion(E, C + 9, apar, a);
        } else {
            T g[6];
#pragma unroll
            for (int j = 0; j < 6; j++) { v[j] = 0; g[j] = P.a_root[j]; }
            xmotion(E, C + 9, g, a);
        }
        add_axis(v, b.axis, qdi);
        T chat[6];
        vxaxis(b.axis, v, qdi, chat);
#pragma unroll
        for (int j = 0; j < 6; j++) a[j] += chat[j];
        add_axis(a, b.axis, qddi);
        if (b.has_child) {
            S.st(b.slot_v, v);
            S.st(b.slot_a3, a);
        }
        T Ia[6], Iv[6], f[6];
        symv_c(C + 12, a, Ia);
        symv_c(C + 12, v, Iv);
        crf(v, Iv, f);
#pragma unroll
        for (int j = 0; j < 6; j++) f[j] += Ia[j];
        if constexpr (GEN) {
            if (L.fext) {
                T Xa[12];
                compose_absolute(S, b.parent_slot_Xa, E, C + 9, Xa);
                if (b.has_child) S.st(b.slot_Xa, Xa);
                subtract_external_force(L, c.first_body + i, Xa, f);
            }
        }
        S.st(b.slot_f, f);
    }
}

template <class T, class SL>
__device__ __forceinline__ void rnea_fwd_free(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                              const Lane<T> &L)
{
    const BodyRec b = load_rec(P.bodies + (c.first_body));
    T v[6], a[6];
    free_base_accel(P, c, L, a);
#pragma unroll
    for (int j = 0; j < 6; j++) {
        v[j] = L.qd(c.v_index + j);
        a[j] += L.x(c.v_index + j);
    }
    if (b.has_child) {
        S.st(b.slot_v, v);
        S.st(b.slot_a3, a);
    }
    T Ia[6], Iv[6], f[6];
    cptr<T> Ic = P.consts + b.cofs + 12;
    symv_c(Ic, a, Ia);
    symv_c(Ic, v, Iv);
    crf(v, Iv, f);
#pragma unroll
    for (int j = 0; j < 6; j++) f[j] += Ia[j];
    if (L.fext) {
        T Xa[12];
        free_absolute(P, c, L, Xa);
        if (b.has_child && b.slot_Xa >= 0) S.st(b.slot_Xa, Xa);
        subtract_external_force(L, c.first_body, Xa, f);
    }
    S.st(b.slot_f, f);
}

template <class T, int N, bool LOOP, bool GEN, class SL>
__device__ __forceinline__ void rnea_bwd_static(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                                const Lane<T> &L)
{
    T tau[N], y[N], yd[N];
#pragma unroll
    for (int a = 0; a < N; a++) {
        tau[a] = 0;
        y[a] = L.cy(c, a);
        yd[a] = L.cyd(c, a);
    }
    const int imp = c.slot_imp_fwd;  // evaluated by the forward step
    for (int i = c.k - 1; i >= 0; i--) {
        const BodyRec b = load_body<GEN>(P.bodies + (c.first_body + i));
        cptr<T> C = P.consts + b.cofs;
        T qi, gi;
        typename RowSel<T, N, LOOP>::type G;
        body_coupling<T, N, LOOP>(P, S, c, imp, i, C, y, qi, G, gi);
        T f[6];
        S.ld(b.slot_f, f);
        const T t = pick(f, b.axis);
#pragma unroll
        for (int a = 0; a < N; a++) tau[a] += G[a] * t;
        if (b.parent >= 0) {
            T sc[2], E[9], fp[6];
            S.ld(b.slot_sc, sc);
            build_E(b.axis, sc[0], sc[1], C, E);
            xforce_inv(E, C + 9, f, fp);
            S.acc(b.parent_slot_f, fp, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < N; a++)
        L.put(c.v_index + a, tau[a]);
}

template <class T, class SL>
__device__ __forceinline__ void rnea_bwd_free(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                              const Lane<T> &L)
{
    const BodyRec b = load_rec(P.bodies + (c.first_body));
    T f[6];
    S.ld(b.slot_f, f);
#pragma unroll
    for (int j = 0; j < 6; j++)
        L.put(c.v_index + j, f[j]);
}

// ---------------------------------------------------------------------------------------------
// RNEA of the revolute / revolute-with-rotor shapes (see aba_bwd_rev).  Leaves finish inside the forward
// step: a rotor (evaluated at q = 0) and a childless link hand X^T f to the parent body and emit their
// torque at once, so only links with children keep a force slot and get a backward step.
// ---------------------------------------------------------------------------------------------
template <class T, class I21>
__device__ __forceinline__ void body_force(const I21 &Ic, const T (&v)[6], const T (&a)[6], T (&f)[6])
{
    T Ia[6], Iv[6];
    symv_c(Ic, a, Ia);
    symv_c(Ic, v, Iv);
    crf(v, Iv, f);  // f = I a + v x* (I v), TreeModel.cpp:185-189
#pragma unroll
    for (int j = 0; j < 6; j++) f[j] += Ia[j];
}

template <class T, bool ROTOR, class SL>
__device__ __forceinline__ void rnea_fwd_rev(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                             const Lane<T> &L)
{
    const BodyRec b = load_rec(P.bodies + c.link_body);
    cptr<T> C = P.consts + b.cofs;
    const T g0 = C[kBodyConstFixed];
    const T yd = L.cyd(c, 0), ydd = L.cx(c, 0);
    const T qdi = g0 * yd;
    T sc[2], E[9], vp[6], ap[6], v[6], a[6], f[6];
    sincos_t(g0 * L.cy(c, 0), &sc[0], &sc[1]);
    rotate_z(sc[0], sc[1], C, E);
    S.ld(b.parent_slot_v, vp);
    S.ld(b.parent_slot_a3, ap);
    xmotion(E, C + 9, vp, v);
    xmotion(E, C + 9, ap, a);
    v[2] += qdi;
    a[0] += v[1] * qdi;
    a[1] -= v[0] * qdi;
    a[3] += v[4] * qdi;
    a[4] -= v[3] * qdi;
    a[2] += g0 * ydd;
    body_force(C + 12, v, a, f);

    T tau_r = 0, fpr[6];
    if constexpr (ROTOR) {
        const BodyRec rb = load_rec(P.bodies + c.rotor_body);
        cptr<T> Cr = P.consts + rb.cofs;
        const T gr = Cr[kBodyConstFixed];
        const T qdr = gr * yd;
        T E0[9], vr[6], ar[6], fr[6];
#pragma unroll
        for (int j = 0; j < 9; j++) E0[j] = Cr[j];
        xmotion(E0, Cr + 9, vp, vr);
        xmotion(E0, Cr + 9, ap, ar);
        vr[2] += qdr;
        ar[0] += vr[1] * qdr;
        ar[1] -= vr[0] * qdr;
        ar[3] += vr[4] * qdr;
        ar[4] -= vr[3] * qdr;
        ar[2] += gr * ydd;
        body_force(Cr + 12, vr, ar, fr);
        tau_r = gr * fr[2];
        xforce_inv(E0, Cr + 9, fr, fpr);
    }
    if (b.has_child) {
        S.st(b.slot_sc, sc);
        S.st(b.slot_v, v);
        S.st(b.slot_a3, a);
        S.st(b.slot_f, f);
        if constexpr (ROTOR) {
            S.acc(b.parent_slot_f, fpr, 0);
            S.st1(c.slot_y0, tau_r);
        }
    } else {
        T fp[6];
        xforce_inv(E, C + 9, f, fp);
        if constexpr (ROTOR) {
#pragma unroll
            for (int j = 0; j < 6; j++) fp[j] += fpr[j];
        }
        S.acc(b.parent_slot_f, fp, 0);
        L.put(c.v_index, g0 * f[2] + tau_r);
    }
}

// backward step of a link with children (the plan drops the step for childless links)
template <class T, bool ROTOR, class SL>
__device__ __forceinline__ void rnea_bwd_rev(const Tables<T> &P, const SL &S, const ClusterRec &c,
                                             const Lane<T> &L)
{
    const BodyRec b = load_rec(P.bodies + c.link_body);
    cptr<T> C = P.consts + b.cofs;
    T f[6], sc[2], E[9], fp[6];
    S.ld(b.slot_f, f);
    S.ld(b.slot_sc, sc);
    T tau = C[kBodyConstFixed] * f[2];
    if constexpr (ROTOR) tau += S.ld1(c.slot_y0);
    rotate_z(sc[0], sc[1], C, E);
    xforce_inv(E, C + 9, f, fp);
    S.acc(b.parent_slot_f, fp, 0);
    L.put(c.v_index, tau);
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// HAS_LOOP is a kernel template parameter: models without implicit-loop clusters run a kernel that
// does not contain the loop-constraint code at all (code size and register pressure matter: the
// interpreter loop must stay resident in the instruction cache)
#ifdef GRBDA_EXP_NMAX2
#define GRBDA_EXP_BIG_N(FN, ...)
#else
#define GRBDA_EXP_BIG_N(FN, ...)                           \
    case 3: FN<T, 3, false, HAS_LOOP>(__VA_ARGS__); break; \
    default: FN<T, 4, false, HAS_LOOP>(__VA_ARGS__); break;
#endif
#ifdef GRBDA_EXP_SHAPES_ONLY
#define GRBDA_DISPATCH_N(c, FN, ...) {}
#else
#define GRBDA_DISPATCH_N(c, FN, ...)                                                        \
    if (HAS_LOOP && (c).kind == CK_LOOP) {                                                     \
        if constexpr (HAS_LOOP) {                                                              \
            switch ((c).n) {                                                                   \
                case 1: FN<T, 1, true, true>(__VA_ARGS__); break;                                    \
                case 2: FN<T, 2, true, true>(__VA_ARGS__); break;                                    \
                default: FN<T, 3, true, true>(__VA_ARGS__); break;                                   \
            }                                                                                  \
        }                                                                                      \
    } else {                                                                                   \
        switch ((c).n) {                                                                       \
            case 1: FN<T, 1, false, HAS_LOOP>(__VA_ARGS__); break;                                       \
            case 2: FN<T, 2, false, HAS_LOOP>(__VA_ARGS__); break;                                       \
            GRBDA_EXP_BIG_N(FN, __VA_ARGS__)                                                             \
        }                                                                                      \
    }
#endif

// WPS: wavefronts per SIMD the kernel is register-allocated for.  f32: 2.  f64: the fast kernel exists for 1
// (no spills; small batches that cannot fill two anyway) and for 2 (~280 spilled registers, still 4-12 %
// faster once the batch fills the chip); the launcher picks.
template <class T, bool HAS_LOOP, int WPS, bool SPLIT = false>
__global__ __launch_bounds__(kWave, WPS) void aba_kernel(DevPlan<T> DP, const T *__restrict__ q,
                                                     const T *__restrict__ qd, const T *__restrict__ tau,
                                                     T *__restrict__ ydd, size_t B, T *__restrict__ scratch)
{
    const Tables<T> P = make_tables(DP);
    const int lane = threadIdx.x;
    // the straight-line shape handlers carry no external-force code: layouts with absolute transforms
    // (DP.fext set) run every cluster through the generic handlers
    const bool use_shapes = !HAS_LOOP || DP.fext == nullptr;
    Slots<T, SPLIT> S;
    S.lane = lane;
    // wave slab: [nq + 2 nv input rows][n_glb_slots state rows], 64 scalars per row
    T *slab = scratch + (size_t)blockIdx.x * (size_t)(DP.n_glb_slots + P.nq + 2 * P.nv) * kWave;
    S.glb = slab + (size_t)(P.nq + 2 * P.nv) * kWave;

#ifdef GRBDA_PROFILE
    unsigned long long prof_acc[24] = {0};
#endif
    PROF_T0();
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + lane;
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
        PROF_SYNCV();
        PROF_ADD(19);  // drain of the previous tile's result stores
        stage_inputs(q, qd, tau, tile, rows_valid, P.nq, P.nv, slab, lane, DP.lds_bytes);
        PROF_SYNCV();
        PROF_ADD(0);
        Lane<T> L;
        L.active = r < B;
        const size_t rr = L.active ? r : B - 1;
        L.in_q = slab + lane;
        L.in_qd = slab + (size_t)P.nq * kWave + lane;
        L.in_x = slab + (size_t)(P.nq + P.nv) * kWave + lane;
        L.out_rows = slab + (size_t)(P.nq + P.nv) * kWave + lane;
        L.fext = DP.fext ? DP.fext + rr * (size_t)DP.n_bodies * 6 : nullptr;
        Carry<T> carry;
#pragma unroll
        for (int j = 0; j < 21; j++) carry.IA[j] = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) carry.psi[j] = 0;
        L.lane = lane;
        T kpre[7] = {0, 0, 0, 0, 0, 0, 0};  // [K][y0] block of the next acceleration step (Layout::acc_k)
        bool kpre_valid = false;
        for (int s = 0; s < P.n_steps; s++) {
            const Step st = load_rec(P.steps + s);
            const ClusterRec c = load_rec(P.clusters + st.cluster);
            PROF_SYNC();
            PROF_ADD(12);  // step + cluster record round trips
            if (use_shapes && (st.op & kOpSkipFast)) continue;
            const int op = st.op & kOpMask;
            if (op == OP_ABA_FWD) {
                if (c.kind == CK_FREE) {
                    aba_fwd_free(P, S, c, L);
                } else if (use_shapes && c.shape) {
                    aba_fwd_rev<T>(P, S, c, L);
                } else {
                    GRBDA_DISPATCH_N(c, aba_fwd_static, P, S, c, L)
                }
                PROF_ADD(2);
            } else if (op == OP_ABA_BWD) {
                if (c.kind == CK_FREE) {
                    aba_bwd_free(P, S, c, L, carry);
                } else if (use_shapes && c.shape == SHAPE_REV) {
                    aba_bwd_rev<T, false>(P, S, c, L, carry);
                } else if (use_shapes && c.shape == SHAPE_REV_ROTOR) {
                    aba_bwd_rev<T, true>(P, S, c, L, carry);
                } else {
                    GRBDA_DISPATCH_N(c, aba_bwd_static, P, S, c, L, carry PROF_PASS)
                }
                PROF_ADD(3);
            } else {
                if (use_shapes && !c.shape) {  // generic / free step: start the prefetch chain for a following shape step
                    const int knext = P.acc_k[s + 1];
                    kpre_valid = knext != -1;
                    if (kpre_valid) S.ldK(knext, kpre);
                }
                if (c.kind == CK_FREE) {
                    aba_acc_free(P, S, c, L);
                } else if (use_shapes && c.shape) {
                    T kblk[7];
#pragma unroll
                    for (int j = 0; j < 7; j++) kblk[j] = kpre[j];
                    const bool have = kpre_valid;
                    // fetch the next step's [K][y0] block now: it has this whole step to arrive
                    const int knext = P.acc_k[s + 1];
                    kpre_valid = knext != -1;
                    if (kpre_valid) S.ldK(knext, kpre);
                    aba_acc_rev<T>(P, S, c, L, kblk, have PROF_PASS);
                } else {
                    GRBDA_DISPATCH_N(c, aba_acc_static, P, S, c, L)
                }
                PROF_ADD(4);
            }
        }
        write_outputs(slab + (size_t)(P.nq + P.nv) * kWave, ydd, tile, rows_valid, P.nv, lane);
        S.flush_bad(DP.bad_count, rows_valid);
        PROF_ADD(18);  // tile epilogue
    }
#ifdef GRBDA_PROFILE
    if (lane == 0)
        for (int i = 0; i < 24; i++) atomicAdd(&grbda_prof[i], prof_acc[i]);
#endif
}

template <class T, bool HAS_LOOP, bool SPLIT = false>
__global__ __launch_bounds__(kWave, 2) void rnea_kernel(DevPlan<T> DP, const T *__restrict__ q,
                                                      const T *__restrict__ qd, const T *__restrict__ ydd,
                                                      T *__restrict__ tau, size_t B, T *__restrict__ scratch)
{
    const Tables<T> P = make_tables(DP);
    const int lane = threadIdx.x;
    // the straight-line shape handlers carry no external-force code: layouts with absolute transforms
    // (DP.fext set) run every cluster through the generic handlers
    const bool use_shapes = !HAS_LOOP || DP.fext == nullptr;
    Slots<T, SPLIT> S;
    S.lane = lane;
    // wave slab: [nq + 2 nv input rows][n_glb_slots state rows], 64 scalars per row
    T *slab = scratch + (size_t)blockIdx.x * (size_t)(DP.n_glb_slots + P.nq + 2 * P.nv) * kWave;
    S.glb = slab + (size_t)(P.nq + 2 * P.nv) * kWave;

    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + lane;
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
        stage_inputs(q, qd, ydd, tile, rows_valid, P.nq, P.nv, slab, lane, DP.lds_bytes);
        Lane<T> L;
        L.active = r < B;
        const size_t rr = L.active ? r : B - 1;
        L.in_q = slab + lane;
        L.in_qd = slab + (size_t)P.nq * kWave + lane;
        L.in_x = slab + (size_t)(P.nq + P.nv) * kWave + lane;
        L.out_rows = slab + (size_t)(P.nq + P.nv) * kWave + lane;
        L.fext = DP.fext ? DP.fext + rr * (size_t)DP.n_bodies * 6 : nullptr;
        L.lane = lane;
        for (int s = 0; s < P.n_steps; s++) {
            const Step st = load_rec(P.steps + s);
            const ClusterRec c = load_rec(P.clusters + st.cluster);
            if (use_shapes && (st.op & kOpSkipFast)) continue;
            const int op = st.op & kOpMask;
            if (op == OP_RNEA_FWD) {
                if (c.kind == CK_FREE) {
                    rnea_fwd_free(P, S, c, L);
                } else if (use_shapes && c.shape == SHAPE_REV) {
                    rnea_fwd_rev<T, false>(P, S, c, L);
                } else if (use_shapes && c.shape == SHAPE_REV_ROTOR) {
                    rnea_fwd_rev<T, true>(P, S, c, L);
                } else {
                    GRBDA_DISPATCH_N(c, rnea_fwd_static, P, S, c, L)
                }
            } else {
                if (c.kind == CK_FREE) {
                    rnea_bwd_free(P, S, c, L);
                } else if (use_shapes && c.shape == SHAPE_REV) {
                    rnea_bwd_rev<T, false>(P, S, c, L);
                } else if (use_shapes && c.shape == SHAPE_REV_ROTOR) {
                    rnea_bwd_rev<T, true>(P, S, c, L);
                } else {
                    GRBDA_DISPATCH_N(c, rnea_bwd_static, P, S, c, L)
                }
            }
        }
        write_outputs(slab + (size_t)(P.nq + P.nv) * kWave, tau, tile, rows_valid, P.nv, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// Steps either side of the path: Newton projection of the dependent coordinates of implicit
// clusters onto phi(q) = 0 (GenericJoint.cpp:289-385 -- the reference does it per state with CasADi root
// finding when it draws random states), and recovery of the spanning velocities / accelerations
// qd_span = G yd, qdd_span = G ydd + g (ClusterJoint.cpp:55-58; the benchmarks' pinocchioBenchmark.cpp:168-176).
// Not hot: one state per lane, inputs read straight from the batch arrays.
// ---------------------------------------------------------------------------------------------
template <class T, int N, class SL>
__device__ __forceinline__ void project_cluster(const Tables<T> &P, const SL &S, const ClusterRec &c, T *qrow,
                                                int max_iter, T tol, bool &good)
{
    const ImpLayout<N> lay(c.slot_imp_fwd, c.slot_imp_bwd, c.k, c.rows);
    const int k = c.k, rows = c.rows;
    cptr<int32_t> ip = P.cints + c.iofs;
    const int hdr0 = ip[0], n_ind = ip[1];
    cptr<int32_t> dep = ip + 3 + n_ind;
    cptr<int32_t> payload = dep + rows;
    for (int i = 0; i < k; i++) S.st1(lay.qs + i, qrow[c.q_index + i]);
    T nrm = 0;
    bool active = true;  // this lane still iterates
    for (int it = 0; it <= max_iter; it++) {
        for (int i = 0; i < rows * k; i++) S.st1(lay.K + i, T(0));
        T phi[3] = {0, 0, 0}, kdq[3];
        if (c.cons_type == 0) loop_position_K<T, N>(P, S, c, lay, payload, hdr0, phi);
        else trig_poly_eval<T, N>(P, S, c, lay, payload, true, kdq, phi);
        const T n2 = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
        if (active) nrm = n2;
        active = active && !(n2 < T(1e-12)) && it < max_iter;  // NaN keeps iterating and fails at the end
        if (!__any(active)) break;
        T Kd[3][3], Kdi[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int j = 0; j < 3; j++) Kd[r][j] = (r < rows && j < rows) ? S.ld1(lay.K + r * k + dep[j]) : T(r == j);
        inv_small(rows, Kd, Kdi);
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (j < rows) {
                const T dq = -(Kdi[j][0] * phi[0] + Kdi[j][1] * phi[1] + Kdi[j][2] * phi[2]);
                const T cur = S.ld1(lay.qs + dep[j]);
                S.st1(lay.qs + dep[j], active ? cur + dq : cur);
            }
    }
    if (!(nrm < tol)) good = false;
    for (int i = 0; i < k; i++) qrow[c.q_index + i] = S.ld1(lay.qs + i);
}

template <class T>
__global__ __launch_bounds__(kWave, 1) void project_kernel(DevPlan<T> DP, int n_clusters, T *__restrict__ q,
                                                           int32_t *__restrict__ ok, size_t B, int max_iter, T tol,
                                                           T *__restrict__ scratch)
{
    const Tables<T> P = make_tables(DP);
    const int lane = threadIdx.x;
    Slots<T> S;
    S.lane = lane;
    S.glb = scratch + (size_t)blockIdx.x * (size_t)(DP.n_glb_slots + P.nq + 2 * P.nv) * kWave +
            (size_t)(P.nq + 2 * P.nv) * kWave;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + lane;
        const bool in_range = r < B;
        T *qrow = q + (in_range ? r : B - 1) * (size_t)P.nq;
        // lanes past the end of the batch redo the last state (same values, same result)
        bool good = true;
        for (int ci = 0; ci < n_clusters; ci++) {
            const ClusterRec c = load_rec(P.clusters + ci);
            if (c.kind != CK_LOOP) continue;
            switch (c.n) {
                case 1: project_cluster<T, 1>(P, S, c, qrow, max_iter, tol, good); break;
                case 2: project_cluster<T, 2>(P, S, c, qrow, max_iter, tol, good); break;
                default: project_cluster<T, 3>(P, S, c, qrow, max_iter, tol, good); break;
            }
        }
        if (ok && in_range) ok[r] = good ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// State input in the reference's conventions (ClusterJoints::Base::toSpanningTreeState, ClusterJoint.cpp:22-71;
// ClusterTreeModel::setState(ModelState), ClusterTreeModel.cpp:256-276): per cluster the caller's positions and
// velocities are either independent or SPANNING coordinates (JointCoordinate::isSpanning()).  The engine's own
// coordinates are independent ones for explicit clusters and spanning positions for implicit ones, so:
//   explicit, spanning positions   y = G+ q_span           (the reference keeps q_span as it is: same state)
//   implicit, spanning positions   valid iff |phi(q)| < tol (LoopConstraint::Base::isValidSpanningPosition,
//                                  LoopConstraint.cpp:15-19, nearZero: 2-norm, Utilities.h:124-130)
//   spanning velocities            valid iff |K qd_span| < tol (isValidSpanningVelocity, LoopConstraint.cpp:22-26);
//                                  yd = the independent entries (G+ qd_span for explicit clusters)
// status[b] = 0, or (1 = "Spanning position is not valid", 2 = "Spanning velocity is not valid") + 256 * cluster for
// the first failure in cluster order, as the reference throws.  cond[b][2] (optional), both maxima over the implicit
// clusters: [0] = max |Kd^-1 Ki|, the entries of G that depend on the state, i.e. how strongly the constraint amplifies
// rates; [1] = |Kd|_F |Kd^-1|_F, the condition number of the block that is inverted (large next to change points of a
// linkage, where [0] stays finite) -- the measures behind the fp32 accuracy gate of generalized_rbda_amd/states.py.
// Not hot: one state per lane, rows read straight from the batch arrays.
// ---------------------------------------------------------------------------------------------
template <class T, int N, class SL>
__device__ __forceinline__ void state_loop_cluster(const Tables<T> &P, const SL &S, const ClusterRec &c, const T *qi,
                                                   const T *vi, bool vel_span, T *qo, T *vo, bool wq, bool wv, T tol,
                                                   int ci, int &st, T &gm, T &kc)
{
    const ImpLayout<N> lay(c.slot_imp_fwd, c.slot_imp_bwd, c.k, c.rows);
    const int k = c.k, rows = c.rows;
    cptr<int32_t> ip = P.cints + c.iofs;
    const int hdr0 = ip[0], n_ind = ip[1];
    cptr<int32_t> ind = ip + 2;
    cptr<int32_t> dep = ip + 3 + n_ind;
    cptr<int32_t> payload = dep + rows;
    for (int i = 0; i < k; i++) {
        const T v = qi[i];
        S.st1(lay.qs + i, v);
        if (wq) qo[i] = v;
    }
    for (int i = 0; i < rows * k; i++) S.st1(lay.K + i, T(0));
    T phi[3] = {0, 0, 0}, kdq[3];
    if (c.cons_type == 0) loop_position_K<T, N>(P, S, c, lay, payload, hdr0, phi);
    else trig_poly_eval<T, N>(P, S, c, lay, payload, true, kdq, phi);
    const T n2 = sqrt(phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2]);
    if (!(n2 < tol) && st == 0) st = 1 + 256 * ci;
    if (vi) {
        if (vel_span) {
            T s2 = 0;
            for (int r = 0; r < rows; r++) {
                T sacc = 0;
                for (int i = 0; i < k; i++) sacc += S.ld1(lay.K + r * k + i) * vi[i];
                s2 += sacc * sacc;
            }
            if (!(sqrt(s2) < tol) && st == 0) st = 2 + 256 * ci;
        }
        for (int a = 0; a < N; a++) {
            const T v = vi[vel_span ? ind[a] : a];
            if (wv) vo[a] = v;
        }
    }
    T Kd[3][3], Kdi[3][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++) Kd[r][j] = (r < rows && j < rows) ? S.ld1(lay.K + r * k + dep[j]) : T(r == j);
    inv_small(rows, Kd, Kdi);
    T f1 = 0, f2 = 0;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (r < rows && j < rows) {
                f1 += Kd[r][j] * Kd[r][j];
                f2 += Kdi[r][j] * Kdi[r][j];
            }
    const T cn = sqrt(f1 * f2);
    kc = (cn > kc || cn != cn) ? cn : kc;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int a = 0; a < N; a++) {
            T x = 0;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (r < rows && j < rows) x += Kdi[r][j] * S.ld1(lay.K + j * k + ind[a]);
            x = fabs(x);
            gm = (x > gm || x != x) ? x : gm;  // NaN (singular Kd) sticks
        }
}

template <class T>
__global__ __launch_bounds__(kWave, 1) void state_kernel(DevPlan<T> DP, int n_clusters, StateFlags F, const T *__restrict__ q_in,
                                                         const T *__restrict__ qd_in, int in_nq, int in_nv, T *__restrict__ q_out,
                                                         T *__restrict__ qd_out, int32_t *__restrict__ status,
                                                         T *__restrict__ gmax, size_t B, T tol, T *__restrict__ scratch)
{
    const Tables<T> P = make_tables(DP);
    const int lane = threadIdx.x;
    Slots<T> S;
    S.lane = lane;
    S.glb = scratch + (size_t)blockIdx.x * (size_t)(DP.n_glb_slots + P.nq + 2 * P.nv) * kWave +
            (size_t)(P.nq + 2 * P.nv) * kWave;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + lane;
        const bool write = r < B;
        const size_t row = write ? r : B - 1;  // lanes past the end redo the last state and write nothing
        const T *qi = q_in + row * (size_t)in_nq;
        const T *vi = qd_in ? qd_in + row * (size_t)in_nv : nullptr;
        T *qo = q_out ? q_out + row * (size_t)P.nq : nullptr;
        T *vo = qd_out ? qd_out + row * (size_t)P.nv : nullptr;
        int st = 0;
        T gm = 0, kc = 0;
        for (int ci = 0; ci < n_clusters; ci++) {
            const ClusterRec c = load_rec(P.clusters + ci);
            const bool ps = (F.pos[ci >> 6] >> (ci & 63)) & 1, vs = (F.vel[ci >> 6] >> (ci & 63)) & 1;
            T *qoc = qo ? qo + c.q_index : nullptr, *voc = vo ? vo + c.v_index : nullptr;
            if (c.kind == CK_FREE) {
                const int npos = P.ori_repr == 0 ? 7 : 6;
                if (qoc && write)
                    for (int j = 0; j < npos; j++) qoc[j] = qi[j];
                if (vi && voc && write)
                    for (int j = 0; j < 6; j++) voc[j] = vi[j];
                qi += npos;
                if (vi) vi += 6;
            } else if (c.kind == CK_LOOP) {
                const bool wq = write && qoc, wv = write && voc;
                switch (c.n) {
                    case 1: state_loop_cluster<T, 1>(P, S, c, qi, vi, vs, qoc, voc, wq, wv, tol, ci, st, gm, kc); break;
                    case 2: state_loop_cluster<T, 2>(P, S, c, qi, vi, vs, qoc, voc, wq, wv, tol, ci, st, gm, kc); break;
                    default: state_loop_cluster<T, 3>(P, S, c, qi, vi, vs, qoc, voc, wq, wv, tol, ci, st, gm, kc); break;
                }
                qi += c.k;
                if (vi) vi += vs ? c.k : c.n;
            } else {  // explicit: consts[dofs] = K (rows x k), then G+ (n x k)  (plan.cpp)
                cptr<T> Kc = P.consts + c.dofs;
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
                        if (!(sqrt(s2) < tol) && st == 0) st = 2 + 256 * ci;
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
        if (status && write) status[r] = st;
        if (gmax && write) {
            gmax[2 * r] = gm;
            gmax[2 * r + 1] = kc;
        }
    }
}

// Absolute transforms world -> body of every body, (E 9 row-major, r 3) per body: TreeNode::Xa_
// (TreeModel.cpp:20-27), the input of the reference's contact-point kinematics (TreeModel.cpp:59-113).
// The poses are composed in the plan's canonical body frames and converted to the reference's frames at the
// end (BodyRec::canon_axis).
template <class T>
__global__ __launch_bounds__(kWave, 1) void poses_kernel(DevPlan<T> DP, int n_clusters, const T *__restrict__ q,
                                                         T *__restrict__ Xa, size_t B)
{
    const Tables<T> P = make_tables(DP);
    const int nb = DP.n_bodies;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + threadIdx.x;
        if (r >= B) continue;
        const T *qr = q + r * (size_t)P.nq;
        T *out = Xa + r * (size_t)nb * 12;
        for (int ci = 0; ci < n_clusters; ci++) {
            const ClusterRec c = load_rec(P.clusters + ci);
            for (int i = 0; i < c.k; i++) {
                const int gb = c.first_body + i;
                const BodyRec b = load_rec(P.bodies + gb);
                T E[9], rr[3];
                if (c.kind == CK_FREE) {
                    T o[4];
                    const int nori = P.ori_repr == 0 ? 4 : 3;
                    for (int j = 0; j < 4; j++) o[j] = j < nori ? qr[c.q_index + 3 + j] : T(0);
                    free_rotation(P.ori_repr, o, E);
                    for (int j = 0; j < 3; j++) rr[j] = qr[c.q_index + j];
                } else {
                    cptr<T> C = P.consts + b.cofs;
                    T qi = 0;
                    if (c.kind == CK_LOOP) {
                        qi = qr[c.q_index + i];
                    } else {
                        for (int a = 0; a < c.n; a++) qi += C[kBodyConstFixed + a] * qr[c.q_index + a];
                    }
                    T sn, cs, El[9];
                    sincos_t(qi, &sn, &cs);
                    build_E(b.axis, sn, cs, C, El);
                    if (b.parent >= 0) {  // Xa = X * Xa_parent: E = E_l E_p, r = r_p + E_p^T r_l
                        const T *Xp = out + (size_t)b.parent * 12;
                        for (int u = 0; u < 3; u++)
                            for (int w = 0; w < 3; w++)
                                E[3 * u + w] = El[3 * u] * Xp[w] + El[3 * u + 1] * Xp[3 + w] + El[3 * u + 2] * Xp[6 + w];
                        for (int u = 0; u < 3; u++) rr[u] = Xp[9 + u] + Xp[u] * C[9] + Xp[3 + u] * C[10] + Xp[6 + u] * C[11];
                    } else {
                        for (int u = 0; u < 9; u++) E[u] = El[u];
                        for (int u = 0; u < 3; u++) rr[u] = C[9 + u];
                    }
                }
                for (int u = 0; u < 9; u++) out[(size_t)gb * 12 + u] = E[u];
                for (int u = 0; u < 3; u++) out[(size_t)gb * 12 + 9 + u] = rr[u];
            }
        }
        // back to the reference's body frames: E_ref = Rc^T E, Rc = the cyclic permutation taking the axis to z
        for (int gb = 0; gb < nb; gb++) {
            const BodyRec b = load_rec(P.bodies + gb);
            if (b.canon_axis == 2) continue;
            T *e = out + (size_t)gb * 12;
            T E[9];
            for (int u = 0; u < 9; u++) E[u] = e[u];
            // x -> z: Rc rows (y, z, x): row 1 <- E row 0, row 2 <- E row 1, row 0 <- E row 2;  y -> z: Rc rows (z, x, y)
            const int to0 = b.canon_axis == 0 ? 1 : 2, to1 = b.canon_axis == 0 ? 2 : 0, to2 = b.canon_axis == 0 ? 0 : 1;
            for (int w = 0; w < 3; w++) {
                e[3 * to0 + w] = E[w];
                e[3 * to1 + w] = E[3 + w];
                e[3 * to2 + w] = E[6 + w];
            }
        }
    }
}

// Spatial velocity and acceleration of every body in its own coordinates, V[B][n_bodies][12] = [v 6 | a 6] ([angular 3; linear 3]
// each): TreeNode::v_ / a_ after TreeModel::forwardAccelerationKinematics (TreeModel.cpp:6-57), the input of the reference's
// contact-point velocities and accelerations (TreeModel.cpp:59-99) and of ClusterTreeModel::getLinearAcceleration
// (ClusterTreeModel.cpp:376-404).  As there, the acceleration is the one the recursions carry: the base starts from -gravity.
// Spanning-tree form: v_i = X_i v_parent + S_i qd_i, a_i = X_i a_parent + S_i qdd_i + v_i x S_i qd_i with the spanning rates
// qd_span = G yd, qdd_span = G ydd + g of spanning_kernel (identical to the cluster form: cJ = Xdot S qd + X S g).
// One state per lane, parents read back from the output row; canonical body frames until the end (see poses_kernel).
template <class T>
__global__ __launch_bounds__(kWave, 1) void twists_kernel(DevPlan<T> DP, int n_clusters, int n_span, const T *__restrict__ q,
                                                          const T *__restrict__ qd_span, const T *__restrict__ qdd_span,
                                                          T *__restrict__ V, size_t B)
{
    const Tables<T> P = make_tables(DP);
    const int nb = DP.n_bodies;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + threadIdx.x;
        if (r >= B) continue;
        const T *qr = q + r * (size_t)P.nq;
        const T *vs = qd_span + r * (size_t)n_span, *as = qdd_span + r * (size_t)n_span;
        T *out = V + r * (size_t)nb * 12;
        int at = 0;
        for (int ci = 0; ci < n_clusters; ci++) {
            const ClusterRec c = load_rec(P.clusters + ci);
            for (int i = 0; i < c.k; i++) {
                const int gb = c.first_body + i;
                const BodyRec b = load_rec(P.bodies + gb);
                T v[6], a[6];
                if (c.kind == CK_FREE) {  // S = 1, the rates are the base twist; a = X (-gravity) + ydd
                    T o[4], E[9];
                    const int nori = P.ori_repr == 0 ? 4 : 3;
                    for (int j = 0; j < 4; j++) o[j] = j < nori ? qr[c.q_index + 3 + j] : T(0);
                    free_rotation(P.ori_repr, o, E);
                    for (int j = 0; j < 6; j++) {
                        v[j] = vs[at + j];
                        a[j] = as[at + j];
                    }
                    for (int u = 0; u < 3; u++) {
                        a[u] += E[3 * u] * P.a_root[0] + E[3 * u + 1] * P.a_root[1] + E[3 * u + 2] * P.a_root[2];
                        a[3 + u] += E[3 * u] * P.a_root[3] + E[3 * u + 1] * P.a_root[4] + E[3 * u + 2] * P.a_root[5];
                    }
                } else {
                    cptr<T> C = P.consts + b.cofs;
                    T qi = 0;
                    if (c.kind == CK_LOOP) {
                        qi = qr[c.q_index + i];
                    } else {
                        for (int k2 = 0; k2 < c.n; k2++) qi += C[kBodyConstFixed + k2] * qr[c.q_index + k2];
                    }
                    T sn, cs, El[9], vp[6], ap[6];
                    sincos_t(qi, &sn, &cs);
                    build_E(b.axis, sn, cs, C, El);
                    if (b.parent >= 0) {
                        const T *Vp = out + (size_t)b.parent * 12;
                        for (int j = 0; j < 6; j++) {
                            vp[j] = Vp[j];
                            ap[j] = Vp[6 + j];
                        }
                    } else {
                        for (int j = 0; j < 6; j++) {
                            vp[j] = 0;
                            ap[j] = P.a_root[j];
                        }
                    }
                    xmotion(El, C + 9, vp, v);
                    xmotion(El, C + 9, ap, a);
                    const T qdi = vs[at + i], qddi = as[at + i];
                    add_axis(v, b.axis, qdi);
                    // v x (S qdi), S the unit angular axis e: [w x e; v_lin x e] qdi
                    const int e0 = b.axis, e1 = (b.axis + 1) % 3, e2 = (b.axis + 2) % 3;
                    a[e1] += v[e2] * qdi;
                    a[e2] -= v[e1] * qdi;
                    a[3 + e1] += v[3 + e2] * qdi;
                    a[3 + e2] -= v[3 + e1] * qdi;
                    add_axis(a, e0, qddi);
                }
                for (int j = 0; j < 6; j++) {
                    out[(size_t)gb * 12 + j] = v[j];
                    out[(size_t)gb * 12 + 6 + j] = a[j];
                }
            }
            at += c.kind == CK_FREE ? 6 : c.k;
        }
        // back to the reference's body frames (x_ref = Rc^T x_canon on each 3-vector, the row permutation of poses_kernel)
        for (int gb = 0; gb < nb; gb++) {
            const BodyRec b = load_rec(P.bodies + gb);
            if (b.canon_axis == 2) continue;
            T *e = out + (size_t)gb * 12;
            const int to0 = b.canon_axis == 0 ? 1 : 2, to1 = b.canon_axis == 0 ? 2 : 0, to2 = b.canon_axis == 0 ? 0 : 1;
            for (int h = 0; h < 4; h++) {
                const T x0 = e[3 * h], x1 = e[3 * h + 1], x2 = e[3 * h + 2];
                e[3 * h + to0] = x0;
                e[3 * h + to1] = x1;
                e[3 * h + to2] = x2;
            }
        }
    }
}

template <class T>
__global__ __launch_bounds__(kWave, 1) void spanning_kernel(DevPlan<T> DP, int n_clusters, int n_span,
                                                            const T *__restrict__ q, const T *__restrict__ qd,
                                                            const T *__restrict__ ydd, T *__restrict__ qd_span,
                                                            T *__restrict__ qdd_span, size_t B, T *__restrict__ scratch)
{
    const Tables<T> P = make_tables(DP);
    const int lane = threadIdx.x;
    Slots<T> S;
    S.lane = lane;
    T *slab = scratch + (size_t)blockIdx.x * (size_t)(DP.n_glb_slots + P.nq + 2 * P.nv) * kWave;
    S.glb = slab + (size_t)(P.nq + 2 * P.nv) * kWave;
    const size_t n_tiles = (B + kWave - 1) / kWave;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const size_t r = tile * kWave + lane;
        const size_t left = B - tile * kWave;
        const int rows_valid = left < (size_t)kWave ? (int)left : kWave;
        stage_inputs(q, qd, ydd, tile, rows_valid, P.nq, P.nv, slab, lane, DP.lds_bytes);
        Lane<T> L;
        L.active = r < B;
        L.in_q = slab + lane;
        L.in_qd = slab + (size_t)P.nq * kWave + lane;
        L.in_x = slab + (size_t)(P.nq + P.nv) * kWave + lane;
        L.out_rows = nullptr;
        L.fext = nullptr;
        L.lane = lane;
        T *ov = qd_span ? qd_span + (L.active ? r : B - 1) * (size_t)n_span : nullptr;
        T *oa = qdd_span + (L.active ? r : B - 1) * (size_t)n_span;
        int at = 0;
        for (int ci = 0; ci < n_clusters; ci++) {
            const ClusterRec c = load_rec(P.clusters + ci);
            if (c.kind == CK_FREE) {
                for (int j = 0; j < 6; j++) {
                    if (ov && L.active) ov[at + j] = L.qd(c.v_index + j);
                    if (L.active) oa[at + j] = L.x(c.v_index + j);
                }
                at += 6;
                continue;
            }
            if (c.kind == CK_LOOP) {
                T yd3[3];
                for (int a = 0; a < 3; a++) yd3[a] = a < c.n ? L.qd(c.v_index + a) : T(0);
                if (c.n == 1) { const T yd1[1] = {yd3[0]}; eval_loop_constraint<T, 1>(P, S, c, L, yd1, true); }
                else if (c.n == 2) { const T yd2[2] = {yd3[0], yd3[1]}; eval_loop_constraint<T, 2>(P, S, c, L, yd2, true); }
                else eval_loop_constraint<T, 3>(P, S, c, L, yd3, true);
            }
            for (int i = 0; i < c.k; i++) {
                T vs = 0, as = 0;
                if (c.kind == CK_LOOP) {
                    const int w = c.n + 1;  // row of G with g in the last column (ImpLayout)
                    for (int a = 0; a < c.n; a++) {
                        const T g = S.ld1(c.slot_imp_fwd + i * w + a);
                        vs += g * L.qd(c.v_index + a);
                        as += g * L.x(c.v_index + a);
                    }
                    as += S.ld1(c.slot_imp_fwd + i * w + c.n);
                } else {
                    const BodyRec b = load_rec(P.bodies + (c.first_body + i));
                    cptr<T> G = P.consts + b.cofs + kBodyConstFixed;
                    for (int a = 0; a < c.n; a++) {
                        vs += G[a] * L.qd(c.v_index + a);
                        as += G[a] * L.x(c.v_index + a);
                    }
                }
                if (ov && L.active) ov[at + i] = vs;
                if (L.active) oa[at + i] = as;
            }
            at += c.k;
        }
    }
}

#ifdef GRBDA_EXP_ONLY_ABA32
// compile-time experiments (register budgets): only the fast f32 ABA kernel is instantiated
#ifdef GRBDA_EXP_SPLIT
template __global__ void aba_kernel<float, false, GRBDA_ABA32_WAVES, true>(DevPlan<float>, const float *, const float *, const float *, float *, size_t, float *);
#else
template __global__ void aba_kernel<float, false, GRBDA_ABA32_WAVES>(DevPlan<float>, const float *, const float *, const float *, float *, size_t, float *);
#endif
#else
// ---------------------------------------------------------------------------------------------
// host launchers (called by capi.cpp)
// ---------------------------------------------------------------------------------------------
template <class T>
hipError_t launch_aba(const DevPlan<T> &P, const T *q, const T *qd, const T *tau, T *ydd, size_t B, T *scratch,
                      int grid, size_t lds_bytes, hipStream_t stream, bool two_waves_per_simd)
{
    constexpr int W32 = GRBDA_ABA32_WAVES;
    if constexpr (sizeof(T) == 4) {
        if (P.general)
            hipLaunchKernelGGL((aba_kernel<T, true, W32>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
        else if (P.split)
            hipLaunchKernelGGL((aba_kernel<T, false, W32, true>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
        else
            hipLaunchKernelGGL((aba_kernel<T, false, W32>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    } else {
        if (P.general)
            hipLaunchKernelGGL((aba_kernel<T, true, 1>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
        else if (two_waves_per_simd)
            hipLaunchKernelGGL((aba_kernel<T, false, 2>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
        else
            hipLaunchKernelGGL((aba_kernel<T, false, 1>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, tau, ydd, B, scratch);
    }
    return hipGetLastError();
}
template <class T>
hipError_t launch_rnea(const DevPlan<T> &P, const T *q, const T *qd, const T *ydd, T *tau, size_t B, T *scratch,
                       int grid, size_t lds_bytes, hipStream_t stream)
{
    if (P.general)
        hipLaunchKernelGGL((rnea_kernel<T, true>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    else if (sizeof(T) == 4 && P.split)
        hipLaunchKernelGGL((rnea_kernel<T, false, sizeof(T) == 4>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    else
        hipLaunchKernelGGL((rnea_kernel<T, false>), dim3(grid), dim3(kWave), lds_bytes, stream, P, q, qd, ydd, tau, B, scratch);
    return hipGetLastError();
}

template <class T>
hipError_t launch_project(const DevPlan<T> &P, int n_clusters, T *q, int32_t *ok, size_t B, int max_iter, T tol,
                          T *scratch, int grid, size_t lds_bytes, hipStream_t stream)
{
    hipLaunchKernelGGL((project_kernel<T>), dim3(grid), dim3(kWave), lds_bytes, stream, P, n_clusters, q, ok, B,
                       max_iter, tol, scratch);
    return hipGetLastError();
}
template <class T>
hipError_t launch_state(const DevPlan<T> &P, int n_clusters, const StateFlags &F, const T *q_in, const T *qd_in, int in_nq, int in_nv,
                        T *q_out, T *qd_out, int32_t *status, T *gmax, size_t B, T tol, T *scratch, int grid, size_t lds_bytes,
                        hipStream_t stream)
{
    hipLaunchKernelGGL((state_kernel<T>), dim3(grid), dim3(kWave), lds_bytes, stream, P, n_clusters, F, q_in, qd_in, in_nq, in_nv,
                       q_out, qd_out, status, gmax, B, tol, scratch);
    return hipGetLastError();
}
template hipError_t launch_state<float>(const DevPlan<float> &, int, const StateFlags &, const float *, const float *, int, int, float *,
                                        float *, int32_t *, float *, size_t, float, float *, int, size_t, hipStream_t);
template hipError_t launch_state<double>(const DevPlan<double> &, int, const StateFlags &, const double *, const double *, int, int,
                                         double *, double *, int32_t *, double *, size_t, double, double *, int, size_t, hipStream_t);
template <class T>
hipError_t launch_spanning(const DevPlan<T> &P, int n_clusters, int n_span, const T *q, const T *qd, const T *ydd,
                           T *qd_span, T *qdd_span, size_t B, T *scratch, int grid, size_t lds_bytes,
                           hipStream_t stream)
{
    hipLaunchKernelGGL((spanning_kernel<T>), dim3(grid), dim3(kWave), lds_bytes, stream, P, n_clusters, n_span, q, qd,
                       ydd, qd_span, qdd_span, B, scratch);
    return hipGetLastError();
}
template <class T>
hipError_t launch_poses(const DevPlan<T> &P, int n_clusters, const T *q, T *Xa, size_t B, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL((poses_kernel<T>), dim3(grid), dim3(kWave), 0, stream, P, n_clusters, q, Xa, B);
    return hipGetLastError();
}
template <class T>
hipError_t launch_twists(const DevPlan<T> &P, int n_clusters, int n_span, const T *q, const T *qd_span, const T *qdd_span, T *V, size_t B,
                         int grid, hipStream_t stream)
{
    hipLaunchKernelGGL((twists_kernel<T>), dim3(grid), dim3(kWave), 0, stream, P, n_clusters, n_span, q, qd_span, qdd_span, V, B);
    return hipGetLastError();
}
template hipError_t launch_twists<float>(const DevPlan<float> &, int, int, const float *, const float *, const float *, float *, size_t, int,
                                         hipStream_t);
template hipError_t launch_twists<double>(const DevPlan<double> &, int, int, const double *, const double *, const double *, double *, size_t,
                                          int, hipStream_t);
template hipError_t launch_poses<float>(const DevPlan<float> &, int, const float *, float *, size_t, int, hipStream_t);
template hipError_t launch_poses<double>(const DevPlan<double> &, int, const double *, double *, size_t, int, hipStream_t);
template hipError_t launch_project<float>(const DevPlan<float> &, int, float *, int32_t *, size_t, int, float, float *,
                                          int, size_t, hipStream_t);
template hipError_t launch_project<double>(const DevPlan<double> &, int, double *, int32_t *, size_t, int, double,
                                           double *, int, size_t, hipStream_t);
template hipError_t launch_spanning<float>(const DevPlan<float> &, int, int, const float *, const float *,
                                           const float *, float *, float *, size_t, float *, int, size_t, hipStream_t);
template hipError_t launch_spanning<double>(const DevPlan<double> &, int, int, const double *, const double *,
                                            const double *, double *, double *, size_t, double *, int, size_t,
                                            hipStream_t);

template hipError_t launch_aba<float>(const DevPlan<float> &, const float *, const float *, const float *, float *,
                                      size_t, float *, int, size_t, hipStream_t, bool);
template hipError_t launch_aba<double>(const DevPlan<double> &, const double *, const double *, const double *,
                                       double *, size_t, double *, int, size_t, hipStream_t, bool);
template hipError_t launch_rnea<float>(const DevPlan<float> &, const float *, const float *, const float *, float *,
                                       size_t, float *, int, size_t, hipStream_t);
template hipError_t launch_rnea<double>(const DevPlan<double> &, const double *, const double *, const double *,
                                        double *, size_t, double *, int, size_t, hipStream_t);

#endif  // GRBDA_EXP_ONLY_ABA32

#ifdef GRBDA_PROFILE
extern "C" int grbda_debug_profile(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(grbda_prof), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(grbda_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifndef GRBDA_EXP_ONLY_ABA32
hipError_t set_max_dynamic_lds()
{
    const int maxb = 160 * 1024;
    const void *fns[] = {reinterpret_cast<const void *>(&project_kernel<float>),
                         reinterpret_cast<const void *>(&project_kernel<double>),
                         reinterpret_cast<const void *>(&state_kernel<float>),
                         reinterpret_cast<const void *>(&state_kernel<double>),
                         reinterpret_cast<const void *>(&spanning_kernel<float>),
                         reinterpret_cast<const void *>(&spanning_kernel<double>),
                         reinterpret_cast<const void *>(&aba_kernel<float, false, GRBDA_ABA32_WAVES>),
                         reinterpret_cast<const void *>(&aba_kernel<float, false, GRBDA_ABA32_WAVES, true>),
                         reinterpret_cast<const void *>(&rnea_kernel<float, false, true>),
                         reinterpret_cast<const void *>(&aba_kernel<float, true, GRBDA_ABA32_WAVES>),
                         reinterpret_cast<const void *>(&aba_kernel<double, false, 1>),
                         reinterpret_cast<const void *>(&aba_kernel<double, false, 2>),
                         reinterpret_cast<const void *>(&aba_kernel<double, true, 1>),
                         reinterpret_cast<const void *>(&rnea_kernel<float, false>),
                         reinterpret_cast<const void *>(&rnea_kernel<float, true>),
                         reinterpret_cast<const void *>(&rnea_kernel<double, false>),
                         reinterpret_cast<const void *>(&rnea_kernel<double, true>)};
    for (const void *f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, maxb);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

#endif

}  // namespace grbda_hip
